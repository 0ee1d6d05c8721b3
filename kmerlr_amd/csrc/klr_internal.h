// klr_internal.h -- shared declarations of libkmerlr_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/kmerlr.h"

typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t  i64;

// ---------------------------------------------------------------------------------------
// device buffer that grows on demand (stream-ordered use only; never shrinks)
// ---------------------------------------------------------------------------------------
struct DevBuf {
  void*  p = nullptr;
  size_t cap = 0;
};

// coefficients of a reduced problem that ride in the row pass's kernel arguments instead of an upload (4 KiB; the runtime
// takes kernel arguments of 8 KiB and more)
#define KLR_THETA_INLINE_MAX 512

enum {
  WS_S = 0,        // per-row weights s_i / w_i (f64[n])
  WS_Z,            // IRLS working response (f64[n])
  WS_PART,         // per-chunk partial sums of the column sweep
  WS_G,            // [g(dim) | scalars] all-reduce buffer
  WS_THETA,        // dense theta on device
  WS_BLK,          // per-block partials of row passes
  WS_SEL_KEYS,     // selection: keys
  WS_SEL_MISC,     // selection: histograms, state, block counts
  WS_SEL_CAND,     // selection: candidates
  WS_MASK,         // u8 mask / colmap
  WS_CD_G,         // Gram matrix
  WS_CD_VEC,       // xy, norm, theta, q, hist, deltas...
  WS_CD_LOSS,      // K losses (+ block partials)
  WS_CD_CTL,       // {ticket, max bits} of k_irls_finish: zeroed when allocated, the kernel leaves it zero
  WS_TMP0,
  WS_TMP1,
  WS_TMP2,
  WS_COUNT
};

struct klr_ctx {
  int          device = 0;
  hipStream_t  stream = nullptr;
  bool         own_stream = false;
  std::string  err;
  // all-reduce hook
  klr_allreduce_fn ar_fn = nullptr;
  void*        ar_user = nullptr;
  int          rank = 0, nranks = 1;
  klr_reduce_scatter_fn rs_fn = nullptr;   // optional: reduce-scatter + all-gather form of the sweep's collective
  klr_allgather_fn      ag_fn = nullptr;
  void*        coll_user = nullptr;
  void*        rccl = nullptr;       // struct RcclState*
  void*        shm = nullptr;        // struct ShmState* (klr_shm.hip)
  hipStream_t  stream2 = nullptr;    // second stream: finish + collective of strip range g while range g + 1 streams
  std::vector<hipEvent_t> range_ev;  // "range g's sweep kernel is done", recorded on `stream`, awaited by `stream2`
  DevBuf       ws[WS_COUNT];
  // pinned host staging
  void*        pin = nullptr;
  bool         sel_cursor_dirty = false;   // a hinted selection left its candidate count in the control block (k_fast_rank_out
                                           // does not clear it: the next sweep's row pass does, or a memset before any other use)
  u64          sel_hint = 0;      // hinted compaction threshold (key) from the last selection, 0 = none
  i64          sel_hint_K = 0, sel_hint_m = 0;
  size_t       pin_cap = 0;
  // profiling of the column sweep
  bool         prof_on = false;
  int          prof_stride = 1, prof_count = 0;   // time every prof_stride-th launch of the sweep kernel (klr_prof_enable(ctx, stride))
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
  // ... and of the reduced-problem CD sweep kernels: (start, stop, sweeps executed) per batch of klr_cd_solve
  struct CdProf { hipEvent_t e0, e1; int sweeps; };
  std::vector<CdProf> prof_cd;
  std::vector<hipEvent_t> prof_free;         // timing events handed back by klr_prof_read(reset): a timed step creates none
  void*        evict_buf = nullptr;          // scratch of klr_prof_evict
  size_t       evict_cap = 0;
  int          n_cu = 256;
  void*        sel_hist_zeroed = nullptr;   // selection histogram known to be all zero
  // device blocks of released reduced matrices (one block per matrix), reused in stream order by the next
  // column sub-select: a leapfrog epoch allocates nothing
  std::vector<std::pair<void*, size_t>> pool;
  // pinned staging of the small host tables of a column sub-select, one async copy; stage_ev = last copy out of it
  void*        stage = nullptr;
  size_t       stage_cap = 0;
  hipEvent_t   stage_ev = nullptr;
  // KLR_* switches: the process environment as it was at klr_ctx_create, changed afterwards only
  // through klr_ctx_set_option -- nothing in the library reads the environment per call.  sw = parsed copies of the
  // switches the full-width step looks at on every launch.
  std::map<std::string, std::string> opts;
  struct Switches {
    int  q64_lds_kib = 96;       // KLR_Q64_LDS_KIB
    int  q64_slds_a = 3;         // KLR_Q64_SLDS_A
    int  stream_nt = 1;          // KLR_STREAM_NT
    int  finish_iter = 0;        // KLR_FINISH_ITER (0: by width)
    bool finish_no_flat = false; // KLR_FINISH_NO_FLAT
    i64  rs_min_dim = 32768;     // KLR_RS_MIN_DIM: from here on the two forms of the collective are TIMED and the faster one used
    bool rs_force = false;       // KLR_RS_FORCE
    int  rs = -1;                // KLR_RS: 1 reduce-scatter form on, 0 off, -1 default (on once the communicator passed its self-check)
    int  overlap = -1;           // KLR_SWEEP_RANGES: strip ranges of the overlapped sweep (0/1: off; -1: default)
  } sw;
  // kernels whose dynamic-LDS limit was raised for THIS context's device (hipFuncSetAttribute is per device: a
  // process-wide latch would leave a second GPU's kernels without it)
  std::set<const void*> lds_attr;
  // sweep collectives that fell back from the reduce-scatter form to the all-reduce (klr_ctx_comm_stats)
  i64 n_rs_sweeps = 0, n_rs_fallbacks = 0, n_ar_sweeps = 0, n_ranged_sweeps = 0;
  // klr_ctx_fork: the contexts forked from this one that are not in use (klr_ctx_release puts them back: warm workspaces
  // for the next cross-validation); they end with this context
  // klr_cd_solve: accepted sweeps of re-linearisation r in the previous solve of this run (klr_solve_run_begin clears it
  // and switches the prediction on): sizes the first batch of re-linearisation r of the next solve
  std::vector<int> cd_pred;
  bool cd_pred_on = false;
  klr_ctx* fork_parent = nullptr;
  std::vector<klr_ctx*> fork_idle;
  std::mutex fork_mu;
  bool comm_checked = false;     // the installed collectives passed klr_ctx_comm_selfcheck
  // strip ranges of the overlapped sweep, chosen once per vector length from the MEASURED reduce-scatter time of the
  // installed hook (sweep_ranges_auto, klr_select.hip): probe_dim = that length, probe_G = the choice, probe_ms = the time
  i64 probe_dim = 0; int probe_G = 1; double probe_ms = 0.0;
  double probe_ag_ms = 0.0, probe_ar_ms = 0.0;   // ... of the all-gather of the candidate blocks and of ONE all-reduce of the vector
  bool probe_scatter = true;                     // reduce-scatter + all-gather beat the all-reduce on this communicator
};

bool prof_take(klr_ctx* ctx);                                   // profiling on and this launch of the sweep is one to be timed
int  prof_event_get(klr_ctx* ctx, hipEvent_t* ev);              // a timing event from the context's pool
void shm_release(klr_ctx* ctx);                                  // klr_shm.hip
// value of switch `name` ("KLR_...") on this context, nullptr when unset
const char* ctx_opt(const klr_ctx* ctx, const char* name);
void ctx_parse_switches(klr_ctx* ctx);
// raise a kernel's dynamic shared memory limit once per context (device)
int  ctx_func_lds(klr_ctx* ctx, const void* fn, int bytes);

bool ctx_alive(const klr_ctx* ctx);                              // false once klr_ctx_destroy ran
int  pool_get(klr_ctx* ctx, size_t bytes, void** p, size_t* cap);
void pool_put(klr_ctx* ctx, void* p, size_t cap);
int  stage_get(klr_ctx* ctx, size_t bytes, void** out);          // waits for the previous copy out of the buffer

// what a launch of k_sweep_q64 needs beside the stream itself: the whole matrix (the tables the ingest made) or one
// strip range of the overlapped form (q64_build_ranges)
struct Q64View { u32 nwg = 0; const u32* t_piece = nullptr; const u32* t_wgpiece = nullptr; u32 slds_rows = 0; };
// Strip ranges of a q64 stream (klr_ingest.hip q64_build_ranges): the strips are dealt into G contiguous ranges of about
// equal stream length, each cut for ALL workgroups like the whole stream is -- one launch per range; while range g + 1
// streams, range g's column sums are final and can be summed, reduce-scattered and searched on a second stream
// (klr_select.hip).  Piece ids run through all ranges (one `part` buffer), ctptr / ctlist group them by strip.
struct Q64Range { u32 strip0 = 0, strip1 = 0; i64 col0 = 0, col1 = 0; u32 piece0 = 0, npiece = 0; u32* t_wgpiece = nullptr; u32 nwg = 0; u32 slds_rows = 0; };
struct Q64Ranges {
  int G = 0;
  std::vector<Q64Range> r;
  u32* t_piece = nullptr; u32* t_ctptr = nullptr; u32* t_ctlist = nullptr;
  u32 npiece = 0, max_ct_pieces = 0;
};

struct klr_mat {
  klr_ctx* ctx = nullptr;
  int  device = 0;
  i64  n = 0, dim = 0, nnz = 0, n_total = 0;
  bool val_f64 = false;          // value arrays are f64 (else f32)
  bool has_explicit_zero = false;
  // ---- CSR (device)
  i64* rowptr = nullptr;         // n+1
  u32* col = nullptr;            // nnz
  void* val = nullptr;           // nnz  (f32|f64)
  // ---- CSC (device), every column segment padded to a multiple of 4 entries (row 0, val 0)
  i64* colptr = nullptr;         // dim+1 (padded offsets)
  u32* crow = nullptr;           // nnz_pad
  void* cval = nullptr;          // nnz_pad
  i64  nnz_pad = 0;
  // chunks of the column sweep: one lane group per chunk
  i64  nchunks = 0;
  i64* chunk_begin = nullptr;    // nchunks (offset into crow/cval, multiple of 4)
  u32* chunk_len = nullptr;      // nchunks (multiple of 4)
  i64* colchunk = nullptr;       // dim+1: chunks of column j are [colchunk[j], colchunk[j+1])
  int  group = 64;               // lanes per chunk
  uint8_t* labels = nullptr;     // n
  uint8_t* active = nullptr;     // n row mask (nullable): rows that take part in sums over samples
  bool view = false;             // klr_mat_view: every device array but `active` belongs to another matrix
  // reduced matrices: every device array above is a slice of ONE pooled block (arena); active_slot = the n bytes
  // reserved for a row mask
  void*    arena = nullptr;
  size_t   arena_cap = 0;
  uint8_t* active_slot = nullptr;
  u32*     d_collen = nullptr;   // reduced matrices: true column lengths on the device (dim)
  i64  n_total_all = 0;          // n_total without a mask
  i64  n_active_local = 0;       // active local rows (valid when active != nullptr)
  std::vector<uint8_t> h_labels; // host copy (class counts under a mask)
  // ---- 2-D tiled packed COO stream for the full-width sweep (klr_sweep.hip, DESIGN.md "data layout"):
  //      tile (rt, ct) = rows [rt*4096, +4096) x columns [ct*4096, +4096); entries row-major inside a
  //      tile; entry = (count:8 << 24) | (col-in-tile:12 << 12) | row-in-tile:12   (4 B per stored nonzero;
  //      counts > 255 are split, the bias column is not stored)
  bool tiled = false, rt_major = false;
  u32  nrt = 0, nct = 0, nwg = 0, npiece = 0;
  i64  total4 = 0;               // quads (4 entries) incl. per-tile padding
  i64  stored_entries = 0;       // entries before padding
  double col_abs_max = 0.0;      // max over non-bias columns of sum_i |x_ij| (local rows)
  double col_abs_max_global = -1.0;   // bound over all ranks (sweep_protocol)
  int    rs_verified = 0;        // the reduce-scatter form of the sweep's collective was checked against the all-reduce form on this matrix
  int    sweep_int = -1;         // protocol of the full-width sweep agreed by all ranks: 1 integer stream + int64 all-reduce,
                                 // 0 plain CSC + f64 all-reduce, -1 not negotiated yet (sweep_protocol, klr_sweep.hip)
  u32* t_ent = nullptr;          // total4*4 packed entries
  u32* t_piece = nullptr;        // npiece * 4: {column tile, quad begin, quad end, first tile index}
  u32* t_tileq = nullptr;        // ntile+1 quad offsets of the tiles (column-tile major: tile = ct*nrt + rt)
  u32* t_wgpiece = nullptr;      // 8 u32 per workgroup: its pieces and everything its first tile needs
  int  fmt = 0;                  // FMT_U32: 4-byte entries (count:8|col:12|row:12); FMT_U24: 3-byte unit entries (col:12|row:12);
                                 // FMT_R64: 8-byte quads (row:12 | 4 x col:12), a row's run padded to 4 columns: 2 bytes per unit entry
  u32 tile_r = 4096, tile_c = 4096;   // rows / columns per tile (<= TILE_DIM), chosen at ingest (tile_geometry)
  u32* t_blkrow = nullptr;       // q64: first row of every 64-unit block of the stream (row fields are relative to it)
  u32  max_ct_pieces = 0;        // most pieces of one column tile / strip (0: not recorded): k_finish_coo's FLAT form up to 8
  u32  q64_slds_rows = 0;        // q64: > 0 = every workgroup's rows fit LDS beside its accumulators (k_sweep_q64 SLDS): the most rows of one
  std::vector<u64> h_q64_sb;     // q64: block offset of every strip (nct + 1) and ...
  std::vector<u32> h_q64_blkrow; // ... the first row of every block, host copies for cuts made later (q64_build_ranges)
  Q64Ranges* qr = nullptr;       // strip-range layout of the overlapped sweep, built on first use
  u32* t_ctptr = nullptr;        // nct+1: pieces per column tile (for the finish kernel)
  u32* t_ctlist = nullptr;       // npiece piece ids grouped by column tile
  // ---- host metadata
  std::vector<i64> h_colptr;     // dim+1 padded offsets
  std::vector<u32> h_collen;     // dim   true lengths
  std::vector<u32> h_colnz;      // dim   entries with value != 0
  i64  n1_local = 0;             // labels == 1 among local rows
  double avg_row = 0.0;
};

// ---------------------------------------------------------------------------------------
// error helpers
// ---------------------------------------------------------------------------------------
extern thread_local std::string g_klr_err;

int klr_fail(klr_ctx* ctx, int code, const char* fmt, ...);

#define KLR_HIP(ctx, expr)                                                             \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess)                                                             \
      return klr_fail((ctx), KLR_EHIP, "%s failed: %s (%s:%d)", #expr,                 \
                      hipGetErrorString(e__), __FILE__, __LINE__);                     \
  } while (0)

#define KLR_TRY(expr)                 \
  do {                                \
    int rc__ = (expr);                \
    if (rc__ != KLR_OK) return rc__;  \
  } while (0)

#define KLR_ARG(ctx, cond)                                                                   \
  do {                                                                                       \
    if (!(cond)) return klr_fail((ctx), KLR_EINVAL, "invalid argument: %s (%s:%d)", #cond,   \
                                 __FILE__, __LINE__);                                        \
  } while (0)

// ---------------------------------------------------------------------------------------
// internal (stream-ordered, no host sync unless stated)
// ---------------------------------------------------------------------------------------
int   ws_get(klr_ctx* ctx, int which, size_t bytes, void** out);
int   pin_get(klr_ctx* ctx, size_t bytes, void** out);
int   klr_allreduce(klr_ctx* ctx, void* dev, i64 count, int dtype);   // dtype KLR_F64|KLR_I64; no-op without hook
int   dev_alloc(klr_ctx* ctx, void** p, size_t bytes);
int   h2d(klr_ctx* ctx, void* dst, const void* src, size_t bytes);     // async on ctx stream
int   d2h_sync(klr_ctx* ctx, void* dst, const void* src, size_t bytes); // copies and waits

// row pass modes
enum { ROW_SWEEP = 0, ROW_LOSS = 1, ROW_IRLS = 2, ROW_LOGPDF = 3, ROW_SWEEP_FIX = 4 };
// Row pass over mat's CSR with dense device theta (length mat->dim).
//  ROW_SWEEP : out0[i] = s_i = cw[y]/n_total * (sigma - y); out1[i] (nullable) = loss term of row i
//  ROW_LOSS  : *loss_acc_dev += sum_i cw[y] * (-log sigma(+-r))
//  ROW_IRLS  : out0[i] = w_i*cw, out1[i] = z_i; loss as ROW_LOSS
//  ROW_LOGPDF: out0[i] = log sigma(r_i)
// loss_acc_dev (nullable) is OVERWRITTEN with the local sum (deterministic two-stage reduction).
int   launch_row_pass(klr_ctx* ctx, const klr_mat* mat, const double* theta_dev, int mode,
                      const double cw[2], double* out0, double* out1, double* loss_acc_dev,
                      const double* theta_host = nullptr, i64 theta_len = 0);   // theta_host (<= KLR_THETA_INLINE_MAX): in the kernel arguments
// out0 for rows when theta has only a bias (no matrix access)
int   launch_row_const(klr_ctx* ctx, const klr_mat* mat, double theta0, int mode, const double cw[2],
                       double* out0, double* out1, double* loss_acc_dev);

// column sweep: g_raw[j] = sum_i s[i] * x_ij (sq: x_ij^2) for every column of mat (length dim)
int   launch_xt_s(klr_ctx* ctx, const klr_mat* mat, const double* s_dev, double* g_raw_dev, bool square);
int   launch_irls_cols(klr_ctx* ctx, const klr_mat* mat, const double* w, const double* z, double** part_out);

// full-width sweep incl. the cross-rank reduction: gbuf[0..dim) = X's, gbuf[dim] = sum of loss terms.
// Count data (mat->tiled): lterm_dev holds per-row loss terms; otherwise gbuf[dim] must already hold the
// local loss partial (row pass loss_acc_dev).
int   sweep_all(klr_ctx* ctx, const klr_mat* mat, const double cw[2], const double* s_dev, const double* lterm_dev,
                double* gbuf);
// count data, fused form used by klr_sweep_select: the row pass writes the fixed-point weights itself
// (sweep_fix_begin gives it scale/S/acc), the conversion back to f64 also emits the selection keys and
// their coarse histogram (keys/hist from sel_prepare; may be NULL).
int   sweep_protocol(klr_ctx* ctx, const klr_mat* mat, bool* use_int);
int   sweep_fix_begin(klr_ctx* ctx, const klr_mat* mat, const double cw[2], double* scale_out, i64** S_out,
                      unsigned long long** acc_out);
// selection candidates and the control block of the fast path (klr_select.hip)
struct Cand { u64 key; u32 idx; u32 pad; };
#define GCAND_CAP 32768
struct FastCtl { u32 cursor, n, above, pad0; u64 low; u32 pad[10]; };
// hinted compaction: while the keys are produced, everything at or above `tau` (a little below the K-th key of the
// previous selection on this context) goes straight into the candidate list -- the selection then needs no pass
// over the keys at all.  tau = 0: off.
// lean: nobody reads g or the keys when the hint holds (no g_out, no transform): the conversion then writes the candidates
// and the scalars only -- 16 bytes per column less, 16 MB at a million columns -- and sweep_fix_emit repeats it in full
// when the hint fails.
struct SelHint { u64 tau; Cand* cand; FastCtl* ctl; bool lean = false; };
// the last stage of sweep_fix_finish / sweep_fix_tail alone, in full (g, keys, coarse histogram; no hint): from the piece
// vectors (one rank) or the all-reduced sums (several) the sweep left behind
int   sweep_fix_emit(klr_ctx* ctx, const klr_mat* mat, double scale, double* gbuf, u64* keys, u32* hist,
                     const unsigned long long* rowblk, i64 nrowblk);
int   sweep_fix_finish(klr_ctx* ctx, const klr_mat* mat, double scale, double* gbuf, u64* keys, u32* hist,
                       const unsigned long long* rowblk, i64 nrowblk, const SelHint* hint = nullptr, i64** ibuf_out = nullptr,
                       i64 ibuf_elems = 0);
// the second half of sweep_fix_finish when it was stopped before its collective (ibuf_out): all-reduce + conversion
int   sweep_fix_tail(klr_ctx* ctx, const klr_mat* mat, double scale, i64* ibuf, double* gbuf, u64* keys, u32* hist,
                     const SelHint* hint);
#define IB_EXTRA 5
// strip-range form of the integer sweep (klr_sweep.hip): layout + shared piece buffer; one launch per range on the ctx
// stream; piece sums of a column range into a vector of its own on any stream
int   sweep_ranges_prepare(klr_ctx* ctx, const klr_mat* mat, int G_want, const Q64Ranges** qr_out, i64** part_out);
int   sweep_range_launch(klr_ctx* ctx, const klr_mat* mat, int g, i64* part);
int   sweep_range_finish(klr_ctx* ctx, const klr_mat* mat, i64 col0, i64 col1, const i64* part, i64* vec, const unsigned long long* rowblk,
                         i64 nrowblk, hipStream_t st);
int   launch_row_pass_fix(klr_ctx* ctx, const klr_mat* mat, const double* theta_dev, double theta0, bool const_mode,
                          const double cw[2], double scale, i64* S, unsigned long long** rowblk, i64* nrowblk,
                          const double* theta_host = nullptr, i64 theta_len = 0, u32* zero_u32 = nullptr);
int   sel_prepare(klr_ctx* ctx, i64 m, u64** keys, u32** hist);
int   apply_transform_dev(klr_ctx* ctx, double* gbuf, i64 dim, const double* scale, const double* offset);

// selection keys: bit pattern of |g| (NaN -> 0); coarse histogram over the top 13 bits
#define FAST_BINS 8192
#define FAST_SHIFT 50
__device__ __forceinline__ u64 abs_key(double g) {
  if (g != g) return 0ull;
  return (u64)__double_as_longlong(fabs(g));
}

// device exclusive scan of u32 counts into i64 offsets (out has n+1 entries)
int   launch_exscan_u32(klr_ctx* ctx, const u32* in, i64 n, i64* out);

// selection on a device gradient (g_dev[0..dim), feature j at g_dev[j+1]); finishes Select on host
int   select_from_device(klr_ctx* ctx, const double* g_dev, i64 m, uint8_t* b, i64 c_prev, i64 N,
                         double lambda_prev, double eps_lambda, double* lambda_out, i64* c_out,
                         int* ok_out, i64* picked_out, bool keys_hist_ready = false,
                         const double* extra_src = nullptr, double* extra_out = nullptr,
                         u64 hint_tau = 0, const std::function<int()>* refill = nullptr);   // refill: g / keys / histogram were not
                                                                                             // written (SelHint.lean): call it before they are read

// matrix construction helpers
int   mat_build_chunks_and_upload(klr_ctx* ctx, klr_mat* mat);   // uses h_colptr/h_collen
void  mat_free(klr_mat* mat);

int   build_reduced(klr_ctx* ctx, const klr_mat* full, const i64* cols, i64 c1, const double* scale,
                    const double* offset, bool need_csc, klr_mat** out);

static inline i64 ceil_div(i64 a, i64 b) { return (a + b - 1) / b; }

// 2-D tiled packed COO (klr_mat.hip build_tiled, klr_sweep.hip k_sweep_coo)
enum { FMT_U32 = 0, FMT_U24 = 1, FMT_R64 = 2,      // tiled formats (k_sweep_coo)
       FMT_Q64C12 = 3, FMT_Q64C13 = 4 };          // row-streamed quads, 4032 / 8128 columns per strip (k_sweep_q64)
static const u32 TILE_BITS = 12;
// an r64 quad with all four columns padding: the quad's position spreads them over the 64 spare slots
static inline __host__ __device__ unsigned long long r64_null_quad(unsigned row, unsigned pos) {
  const unsigned long long d = 4032u + (pos & 63u);
  return ((unsigned long long)row << 48) | (d << 36) | (d << 24) | (d << 12) | d;
}
static const u32 R64_DUMP = 4032;          // accumulator slots 4032..4095 take the padding columns of r64 quads (tiles are at most 4032 wide)
static const u32 TILE_DIM = 1u << TILE_BITS;   // at most 4096 rows x 4096 columns per tile: 32 KiB of s + 32 KiB of g in LDS
// rows / columns per tile and the tile grid for an n x dim shard swept by nwg workgroups
void tile_geometry(i64 n, i64 dim, u32 nwg, u32* tile_r, u32* tile_c, i64* nrt, i64* nct);
// klr_ingest.hip: CSC + tiles + host metadata from a narrow count CSR already on the device
int device_finish_csr(klr_ctx* ctx, klr_mat* m);
// strip-range layout with G ranges (G >= 2) of a q64 matrix; *qr_out = nullptr when the matrix has too few strips
int q64_build_ranges(klr_ctx* ctx, klr_mat* m, int G);
void q64_free_ranges(klr_mat* m);
