// klr_mat.hip -- the sparse k-mer / score matrix in HBM: ingest of the reference row layout,
// CSR + padded CSC + sweep chunks, and the column sub-select (featureSelection.Data).
//
// HBM layout (DESIGN.md "data layout"):
//   CSR   rowptr i64[n+1], col u32[nnz], val f32|f64[nnz]            rows = samples
//   CSC   colptr i64[dim+1], crow u32[nnz_pad], cval f32|f64[nnz_pad] columns = k-mers;
//         every column segment starts 16-byte aligned and is padded to a multiple of 4 entries
//         with (row 0, value 0) so a lane can fetch 4 entries with one dwordx4 load per array
//   chunks  a column longer than CHUNK_MAX entries is cut into chunks; the sweep gives one lane
//         group to each chunk and a second tiny kernel adds a column's chunk sums in order
// Values are stored as f32 when every value is exactly representable (k-mer counts are small
// integers), else f64; all accumulation is f64.
#include "klr_internal.h"

#include <memory>

#include <algorithm>
#include <cstring>
#include <cmath>
#include <thread>

static const i64 CHUNK_MAX = 4096;   // entries per sweep chunk (multiple of 256)

void mat_free(klr_mat* m) {
  if (!m) return;
  // the owning ctx may already be gone: use the recorded device, wait for everything on it
  (void)hipSetDevice(m->device);
  if (m->arena) {
    // a reduced matrix: one pooled block; the context hands it to the next column sub-select in stream order
    if (ctx_alive(m->ctx) && m->ctx->device == m->device) pool_put(m->ctx, m->arena, m->arena_cap);
    else { (void)hipDeviceSynchronize(); (void)hipFree(m->arena); }
    delete m;
    return;
  }
  if (m->view) {
    // a view owns its row mask and nothing else
    if (m->active) {
      if (ctx_alive(m->ctx) && m->ctx->device == m->device) (void)hipStreamSynchronize(m->ctx->stream);   // its own context's work only
      else (void)hipDeviceSynchronize();
      (void)hipFree(m->active);
    }
    delete m;
    return;
  }
  (void)hipDeviceSynchronize();
  q64_free_ranges(m);
  void* ps[] = {m->rowptr, m->col, m->val, m->colptr, m->crow, m->cval,
                m->chunk_begin, m->chunk_len, m->colchunk, m->labels, m->active, m->t_ent, m->t_piece, m->t_wgpiece,
                m->t_ctptr, m->t_ctlist, m->t_tileq, m->t_blkrow};
  for (void* p : ps) if (p) (void)hipFree(p);
  delete m;
}

extern "C" void klr_mat_destroy(klr_mat* mat) { mat_free(mat); }

// A matrix for ANOTHER context of the same device that shares `mat`'s device storage: its own row mask, its own
// protocol / range state, everything else read-only and common.  `mat` must outlive the view and must not be destroyed
// or re-masked while views of it are in use by other threads (row masks are per view: klr_mat_set_active_rows).
extern "C" int klr_mat_view(klr_ctx* ctx, const klr_mat* mat, klr_mat** out) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, mat != nullptr && out != nullptr);
  *out = nullptr;
  if (mat->device != ctx->device) return klr_fail(ctx, KLR_EINVAL, "klr_mat_view: the matrix lives on device %d, the context on %d", mat->device, ctx->device);
  if (mat->arena) return klr_fail(ctx, KLR_EINVAL, "klr_mat_view: reduced matrices are not shared");
  klr_mat* v = new klr_mat(*mat);
  v->ctx = ctx;
  v->view = true;
  v->active = nullptr;
  v->qr = nullptr;
  v->rs_verified = 0;
  v->n_total = mat->n_total_all;
  v->n_active_local = mat->n;
  v->n1_local = 0;
  for (uint8_t l : v->h_labels) if (l) v->n1_local++;
  *out = v;
  return KLR_OK;
}

extern "C" int klr_mat_shape(const klr_mat* mat, int64_t* n, int64_t* dim, int64_t* nnz, int64_t* n_total) {
  if (!mat) return klr_fail(nullptr, KLR_EINVAL, "mat is NULL");
  if (n) *n = mat->n;
  if (dim) *dim = mat->dim;
  if (nnz) *nnz = mat->nnz;
  if (n_total) *n_total = mat->n_total;
  return KLR_OK;
}

extern "C" int klr_mat_layout(const klr_mat* mat, int* tiled, int64_t* tile_rows, int64_t* nseg,
                               int64_t* padded_entries, int* lanes_per_segment, int* value_bytes) {
  if (!mat) return klr_fail(nullptr, KLR_EINVAL, "mat is NULL");
  if (tiled) *tiled = mat->tiled ? 1 : 0;
  if (tile_rows) *tile_rows = mat->tiled ? (int64_t)mat->tile_r : mat->n;
  if (nseg) *nseg = mat->tiled ? (int64_t)mat->npiece : mat->nchunks;
  const bool quads = mat->fmt == FMT_R64 || mat->fmt == FMT_Q64C12 || mat->fmt == FMT_Q64C13;     // 8 unit entries per 16-byte unit
  if (padded_entries) *padded_entries = mat->tiled ? mat->total4 * (quads ? 8 : 4) : mat->nnz_pad;
  if (lanes_per_segment) *lanes_per_segment = mat->tiled ? 0 : mat->group;
  // tiled: bytes per padded entry = 3 + value_bytes (u32: 1, u24: 0, r64 / q64: -1 -- four columns share one row field)
  if (value_bytes) *value_bytes = mat->tiled ? (mat->fmt == FMT_U32 ? 1 : (mat->fmt == FMT_U24 ? 0 : -1)) : (mat->val_f64 ? 8 : 4);
  return KLR_OK;
}

extern "C" int klr_mat_stream_info(const klr_mat* mat, int* format, int64_t* stream_bytes, int64_t* piece_bytes, int64_t* table_bytes) {
  if (!mat) return klr_fail(nullptr, KLR_EINVAL, "mat is NULL");
  const bool t = mat->tiled;
  if (format) *format = t ? mat->fmt : -1;
  if (stream_bytes) *stream_bytes = t ? mat->total4 * (mat->fmt == FMT_U24 ? 12 : 16) : 0;
  const bool q64 = mat->fmt == FMT_Q64C12 || mat->fmt == FMT_Q64C13;
  if (piece_bytes) *piece_bytes = t ? (mat->fmt == FMT_Q64C13 ? 8192 : (int64_t)TILE_DIM) * 8 : 0;
  // read by the sweep: piece records (16 B) and -- tiled formats: one 32-byte record per workgroup, tile offsets (4 B per
  // tile); q64: a 32-byte record per workgroup, the first row of every 64-unit block (4 B per KiB of stream)
  if (table_bytes) *table_bytes = !t ? 0 : q64 ? (int64_t)mat->npiece * 16 + (int64_t)mat->nwg * 32 + mat->total4 / 64 * 4
                                               : (int64_t)mat->npiece * 16 + (int64_t)mat->nwg * 32 + ((int64_t)mat->nrt * mat->nct + 2) * 4;
  return KLR_OK;
}

static int pick_group(double avg_len) {
  int g = 4;
  while (g < 64 && 4.0 * g < avg_len) g *= 2;
  return g;
}

// Builds the chunk list from h_colptr/h_collen and uploads colptr + chunks.
int mat_build_chunks_and_upload(klr_ctx* ctx, klr_mat* mat) {
  const i64 dim = mat->dim;
  std::vector<i64> cb, cc(dim + 1);
  std::vector<u32> cl;
  cb.reserve(dim + 16); cl.reserve(dim + 16);
  for (i64 j = 0; j < dim; j++) {
    cc[j] = (i64)cb.size();
    i64 b = mat->h_colptr[j], e = mat->h_colptr[j + 1];
    while (b < e) {
      i64 len = std::min(CHUNK_MAX, e - b);
      cb.push_back(b); cl.push_back((u32)len);
      b += len;
    }
  }
  cc[dim] = (i64)cb.size();
  mat->nchunks = (i64)cb.size();
  mat->group = pick_group(mat->nchunks ? (double)mat->nnz_pad / (double)mat->nchunks : 4.0);
  KLR_TRY(dev_alloc(ctx, (void**)&mat->colptr, (size_t)(dim + 1) * sizeof(i64)));
  KLR_TRY(dev_alloc(ctx, (void**)&mat->chunk_begin, (size_t)std::max<i64>(mat->nchunks, 1) * sizeof(i64)));
  KLR_TRY(dev_alloc(ctx, (void**)&mat->chunk_len, (size_t)std::max<i64>(mat->nchunks, 1) * sizeof(u32)));
  KLR_TRY(dev_alloc(ctx, (void**)&mat->colchunk, (size_t)(dim + 1) * sizeof(i64)));
  KLR_HIP(ctx, hipMemcpy(mat->colptr, mat->h_colptr.data(), (size_t)(dim + 1) * sizeof(i64), hipMemcpyHostToDevice));
  if (mat->nchunks) {
    KLR_HIP(ctx, hipMemcpy(mat->chunk_begin, cb.data(), (size_t)mat->nchunks * sizeof(i64), hipMemcpyHostToDevice));
    KLR_HIP(ctx, hipMemcpy(mat->chunk_len, cl.data(), (size_t)mat->nchunks * sizeof(u32), hipMemcpyHostToDevice));
  }
  KLR_HIP(ctx, hipMemcpy(mat->colchunk, cc.data(), (size_t)(dim + 1) * sizeof(i64), hipMemcpyHostToDevice));
  return KLR_OK;
}

// pieces / per-workgroup ranges / per-column-tile piece lists / tile offsets of a tiled stream whose tile
// quad offsets are toff4[ntile+1]; uploads them and marks the matrix tiled (shared by the host and the device ingest)
int build_pieces_and_upload(klr_ctx* ctx, klr_mat* m, const std::vector<u64>& toff4, i64 nrt, i64 nct, bool rt_major) {
  const i64 ntile = nrt * nct;
  const i64 total4 = (i64)toff4[(size_t)ntile];
  // pieces: equal parts of the quad stream, cut where the column tile changes
  const u32 nwg = (u32)std::max(2, 2 * ctx->n_cu);
  std::vector<u32> piece; piece.reserve((size_t)(nwg + nct) * 4);
  std::vector<u32> wgp((size_t)nwg + 1);
  // cut points: every workgroup gets an equal share of what is left of the stream, snapped to a tile boundary
  // where one lies within a few percent of a share (a workgroup that starts and ends on tile boundaries has no
  // boundary work in mid-stream; with tile_geometry's one-tile-per-workgroup grid every cut snaps)
  std::vector<u64> cut((size_t)nwg + 1);
  {
    const char* sn = ctx_opt(ctx, "KLR_CUT_SNAP");
    const double tolf = sn ? atof(sn) : 0.08;
    i64 t = 0;
    cut[0] = 0;
    for (u32 w = 1; w <= nwg; w++) {
      const u64 prev = cut[w - 1];
      const u64 share = ((u64)total4 - prev) / (u64)(nwg - w + 1);
      const u64 ideal = prev + share;
      const double tol = tolf * (double)share;
      while (t < ntile && toff4[(size_t)t + 1] <= ideal) t++;
      u64 c = ideal;
      const u64 lo = toff4[(size_t)t], hi = t < ntile ? toff4[(size_t)t + 1] : (u64)total4;
      if (lo > prev && (double)(ideal - lo) <= tol && (ideal - lo) <= (hi - ideal)) c = lo;
      else if ((double)(hi - ideal) <= tol) c = hi;
      cut[w] = std::max(c, prev);
    }
    cut[nwg] = (u64)total4;
  }
  {
    i64 tile = 0;
    for (u32 w = 0; w < nwg; w++) {
      wgp[w] = (u32)(piece.size() / 4);
      const u64 q0 = cut[w], q1 = cut[w + 1];
      u64 q = q0;
      while (q < q1) {
        while (toff4[(size_t)tile + 1] <= q) tile++;
        const i64 ct = rt_major ? tile % nct : tile / nrt;
        // a piece ends where the column tile changes: the next tile (row-tile major) or the end of the
        // column strip (column-tile major)
        const u64 ct_end = rt_major ? toff4[(size_t)tile + 1] : toff4[(size_t)((ct + 1) * nrt)];
        const u64 e = std::min<u64>(q1, ct_end);
        piece.push_back((u32)ct); piece.push_back((u32)q); piece.push_back((u32)e); piece.push_back((u32)tile);
        q = e;
      }
    }
    wgp[nwg] = (u32)(piece.size() / 4);
  }
  const u32 npiece = (u32)(piece.size() / 4);
  std::vector<u32> ctptr((size_t)nct + 1, 0u), ctlist(npiece);
  for (u32 p = 0; p < npiece; p++) ctptr[(size_t)piece[(size_t)p * 4] + 1]++;
  for (i64 c = 0; c < nct; c++) ctptr[(size_t)c + 1] += ctptr[(size_t)c];
  {
    std::vector<u32> cur(ctptr.begin(), ctptr.end() - 1);
    for (u32 p = 0; p < npiece; p++) ctlist[cur[(size_t)piece[(size_t)p * 4]]++] = p;
  }
  KLR_TRY(dev_alloc(ctx, (void**)&m->t_piece, (size_t)npiece * 16));
  KLR_TRY(dev_alloc(ctx, (void**)&m->t_wgpiece, (size_t)nwg * 32));
  KLR_TRY(dev_alloc(ctx, (void**)&m->t_ctptr, ((size_t)nct + 1) * 4));
  KLR_TRY(dev_alloc(ctx, (void**)&m->t_ctlist, (size_t)std::max<u32>(npiece, 1) * 4));
  KLR_TRY(dev_alloc(ctx, (void**)&m->t_tileq, ((size_t)ntile + 2) * 4));
  {
    std::vector<u32> tq((size_t)ntile + 2);
    for (i64 t = 0; t <= ntile; t++) tq[(size_t)t] = (u32)toff4[(size_t)t];
    tq[(size_t)ntile + 1] = (u32)total4;                 // sentinel: the kernel reads tileq[t+1] before it stops
    KLR_HIP(ctx, hipMemcpy(m->t_tileq, tq.data(), ((size_t)ntile + 2) * 4, hipMemcpyHostToDevice));
  }
  KLR_HIP(ctx, hipMemcpy(m->t_piece, piece.data(), (size_t)npiece * 16, hipMemcpyHostToDevice));
  {
    // per-workgroup record: {first piece, end piece, first tile, its row tile} {-, quad begin, quad end of the first
    // piece, quad end inside the first tile}
    std::vector<u32> rec((size_t)nwg * 8, 0u);
    for (u32 w = 0; w < nwg; w++) {
      u32* r = &rec[(size_t)w * 8];
      r[0] = wgp[w]; r[1] = wgp[w + 1];
      if (wgp[w] == wgp[w + 1]) continue;
      const u32* d = &piece[(size_t)wgp[w] * 4];
      const i64 t0 = d[3];
      r[2] = (u32)t0; r[3] = (u32)(rt_major ? t0 / nct : t0 % nrt);
      r[5] = d[1]; r[6] = d[2]; r[7] = (u32)std::min<u64>(d[2], toff4[(size_t)t0 + 1]);
    }
    KLR_HIP(ctx, hipMemcpy(m->t_wgpiece, rec.data(), (size_t)nwg * 32, hipMemcpyHostToDevice));
  }
  KLR_HIP(ctx, hipMemcpy(m->t_ctptr, ctptr.data(), ((size_t)nct + 1) * 4, hipMemcpyHostToDevice));
  KLR_HIP(ctx, hipMemcpy(m->t_ctlist, ctlist.data(), (size_t)npiece * 4, hipMemcpyHostToDevice));
  m->rt_major = rt_major; m->tiled = true; m->nrt = (u32)nrt; m->nct = (u32)nct; m->nwg = nwg; m->npiece = npiece;
  m->total4 = total4;
  return KLR_OK;
}

// =================================================================================================
// 2-D tiled packed COO stream for the full-width sweep.  Built when every value is a non-negative
// integer (k-mer counts) and the matrix is large enough for the sweep to be bandwidth bound;
// otherwise the plain CSC kernel is used.
//   tile (rt, ct): rows [rt*RT, +RT) x columns [ct*CT, +CT), RT, CT <= 4096 (tile_geometry); tiles are stored COLUMN-TILE major
//            (tile id = ct*nrt + rt) so a workgroup walks down one column tile, reloading only the s tile,
//            and writes its 4096 column sums once per column tile it touches
//   entry  = 4 bytes (count:8 << 24) | (col - ct*CT) << 12 | (row - rt*RT), counts > 255 split into several
//            entries (the sweep is additive) -- or 3 bytes (col - ct*CT) << 12 | (row - rt*RT), a count c stored
//            as c entries, whichever is fewer bytes; column 0 (bias, all ones) is not stored
//   order  = row-major inside a tile (so consecutive entries share the row -> LDS broadcast reads of s
//            and distinct columns -> few same-address LDS atomics); every tile padded to 4 entries
//   pieces = the entry stream cut into nwg equal parts (one per workgroup), each part cut where the
//            column tile changes; a piece produces one 4096-wide int64 partial column-sum vector
// =================================================================================================
// Tile geometry.  A tile is at most 4094 x 4032 (12-bit offsets in an entry, 32 KiB + 32 KiB of LDS; row slot 4095 is the
// null entry's zero, column slots 4032..4095 take the padding columns of r64 quads).  Columns:
// the fewest strips that cover dim, equally wide.  Rows: if the matrix is small enough for one tile per
// workgroup (nrt_min * nct <= nwg), as many equally tall row tiles as give one tile per workgroup -- then a
// workgroup starts and ends on tile boundaries and streams its whole part without boundary work in between
// (configs[2]: 30 x 17 tiles of 3334 x 3856 for 512 workgroups); otherwise the tallest tiles.
void tile_geometry(i64 n, i64 dim, u32 nwg, u32* tile_r, u32* tile_c, i64* nrt_out, i64* nct_out) {
  i64 nct = std::max<i64>(1, ceil_div(dim, (i64)R64_DUMP));       // at most 4032 columns: 64 spare accumulator slots
  const i64 CT = std::max<i64>(1, ceil_div(dim, nct));
  nct = ceil_div(dim, CT);
  const i64 RT_MAX = TILE_DIM - 2;                           // slot 4095 of the s tile is the null entry's zero
  const i64 nrt_min = std::max<i64>(1, ceil_div(n, RT_MAX));
  i64 nrt = nrt_min;
  if (nrt_min * nct <= (i64)nwg) nrt = std::max<i64>(nrt_min, (i64)nwg / nct);
  i64 RT = ceil_div(std::max<i64>(n, 1), nrt);
  RT = std::min<i64>((RT + 1) & ~(i64)1, RT_MAX);              // even: the s tile is read with 16-byte loads
  nrt = std::max<i64>(1, ceil_div(n, RT));
  *tile_r = (u32)RT; *tile_c = (u32)CT; *nrt_out = nrt; *nct_out = nct;
}

static int build_tiled(klr_ctx* ctx, klr_mat* m, const i64* rowptr, const i64* col, const double* val) {
  const i64 n = m->n, dim = m->dim;
  if (n < 2048 || m->nnz < (1 << 20)) return KLR_OK;          // small: plain kernel
  i64 nrt, nct;
  tile_geometry(n, dim, (u32)std::max(2, 2 * ctx->n_cu), &m->tile_r, &m->tile_c, &nrt, &nct);
  const i64 RT = m->tile_r, CT = m->tile_c;
  const i64 ntile = nrt * nct;
  if (ntile >= (i64)1 << 31) return KLR_OK;
  // Tile order.  Row-tile major: a workgroup crossing a tile boundary keeps its s tile and starts a new
  // 4096-wide output vector (one 32 KiB write per tile).  Column-tile major: it keeps the output vector and
  // reloads the s tile (32 KiB from L2, no HBM traffic).  Dense tiles (>= 32k entries, e.g. 100k x 65k k-mer
  // counts) amortise the writes and run fastest row-tile major; sparse tiles (1M columns) are dominated by the
  // per-tile writes and run column-tile major (measured: profiles/README.md).  With one tile per workgroup the
  // two orders do the same work; column-tile major measured 3-5 % faster (neighbouring workgroups then read
  // different s tiles instead of 17 of them the same one).  KLR_TILE_ORDER=rt|ct overrides.
  const char* ord = ctx_opt(ctx, "KLR_TILE_ORDER");
  bool rt_major = (double)m->nnz / (double)ntile >= 32768.0 && ntile > 2 * (i64)ctx->n_cu;   // one tile per workgroup: column-tile major
  if (ord && ord[0] == 'r') rt_major = true;
  if (ord && ord[0] == 'c') rt_major = false;
  auto tix = [&](i64 rt, i64 ct) -> i64 { return rt_major ? rt * nct + ct : ct * nrt + rt; };
  int T = (int)std::min<i64>(std::max<unsigned>(1u, std::min(8u, std::thread::hardware_concurrency())), nrt);
  std::vector<int> bad(T, 0);
  std::vector<u64> tlen((size_t)ntile, 0);           // 4-byte entries per tile (counts > 255 split)
  std::vector<u64> tlen1((size_t)ntile, 0);          // unit entries per tile (sum of the counts)
  std::vector<u64> tlenr((size_t)ntile, 0);          // r64: quads per tile (every (row, tile) run padded to 4 columns)
  std::vector<std::vector<double>> colsum(T, std::vector<double>((size_t)dim, 0.0));
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
      for (i64 rt = t; rt < nrt; rt += T) {
        u64* l = tlen.data();
        u64* l1 = tlen1.data();
        u64* lr = tlenr.data();
        const i64 r0 = rt * RT, r1 = std::min<i64>(n, r0 + RT);
        for (i64 i = r0; i < r1; i++) {
          i64 cur_ct = -1; u64 run = 0;
          for (i64 a = rowptr[i]; a < rowptr[i + 1]; a++) {
            const double v = val[a];
            if (!(v >= 0.0 && v <= 16777216.0 && v == (double)(u32)v)) { bad[t] = 1; return; }
            const i64 c = col[a];
            if (c == 0) continue;
            const u32 iv = (u32)v;
            const i64 ct = c / CT;
            if (ct != cur_ct) { if (cur_ct >= 0) lr[(size_t)tix(rt, cur_ct)] += (run + 3) / 4; cur_ct = ct; run = 0; }
            run += iv;
            l[(size_t)tix(rt, ct)] += (iv + 254u) / 255u;               // 0 -> no entry
            l1[(size_t)tix(rt, ct)] += iv;
            colsum[t][(size_t)c] += v;
          }
          if (cur_ct >= 0) lr[(size_t)tix(rt, cur_ct)] += (run + 3) / 4;
        }
      }
    });
    for (auto& x : th) x.join();
  }
  for (int t = 0; t < T; t++) if (bad[t]) return KLR_OK;      // not count data: keep the plain path
  double amax = 0.0;
  for (i64 j = 1; j < dim; j++) { double sj = 0.0; for (int t = 0; t < T; t++) sj += colsum[t][(size_t)j]; amax = std::max(amax, sj); }
  colsum.clear();
  // entry format, whichever is fewer bytes (KLR_ENTRY_FORMAT=u32|u24|r64 overrides), all row-major inside a tile:
  //   u32  4-byte entries with an 8-bit count (counts > 255 split)
  //   u24  3-byte unit entries (a k-mer count row sums to its number of windows, so unit entries cost at most
  //        windows / distinct more entries)
  //   r64  unit entries with every (row, tile) run padded to 4 columns: an 8-byte quad = row:12 << 48 | 4 x col:12, i.e.
  //        2 bytes per unit entry, one s read per quad; padding columns point at the spare accumulator slots 4032..4095
  u64 tot4 = 0, tot1 = 0, totr = 0, units1 = 0;
  for (i64 t = 0; t < ntile; t++) {
    tot4 += (tlen[(size_t)t] + 3) / 4; tot1 += (tlen1[(size_t)t] + 3) / 4; totr += (tlenr[(size_t)t] + 1) / 2; units1 += tlen1[(size_t)t];
  }
  int fmt = FMT_U32;
  u64 best = 16 * tot4;
  if (units1 < ((u64)1 << 31)) {                     // unit entries only while the stream stays small
    if (12 * tot1 <= best) { fmt = FMT_U24; best = 12 * tot1; }
    if (16 * totr <= best) { fmt = FMT_R64; best = 16 * totr; }
    if (const char* ef = ctx_opt(ctx, "KLR_ENTRY_FORMAT")) {
      if (!strcmp(ef, "u24")) fmt = FMT_U24; else if (!strcmp(ef, "u32")) fmt = FMT_U32; else if (!strcmp(ef, "r64")) fmt = FMT_R64;
    }
  }
  u64 stored = 0;
  for (i64 t = 0; t < ntile; t++) stored += fmt == FMT_U32 ? tlen[(size_t)t] : tlen1[(size_t)t];
  std::vector<u64> toff4((size_t)ntile + 1);
  u64 run = 0;
  for (i64 t = 0; t < ntile; t++) {
    toff4[(size_t)t] = run;
    run += fmt == FMT_R64 ? (tlenr[(size_t)t] + 1) / 2 : ((fmt == FMT_U24 ? tlen1[(size_t)t] : tlen[(size_t)t]) + 3) / 4;
  }
  toff4[(size_t)ntile] = run;
  if (run >= 0xFFFFFFF0ull || run == 0) return KLR_OK;
  const i64 total4 = (i64)run;
  const size_t qbytes = fmt == FMT_U24 ? 12 : 16;     // bytes per load unit: 4 entries (u32, u24) or 2 quads = 8 entries (r64)
  std::vector<unsigned char> ent((size_t)total4 * qbytes, 0);
  if (fmt == FMT_U24) for (size_t e = 0; e < (size_t)total4 * 4; e++) { ent[e * 3] = 0xFF; ent[e * 3 + 1] = 0x0F; }   // null entries: row 4095, column 0
  if (fmt == FMT_R64) {                               // null quads: row 4095 (s = 0), spread padding columns
    u64* w = reinterpret_cast<u64*>(ent.data());
    for (size_t e = 0; e < (size_t)total4 * 2; e++) w[e] = r64_null_quad(4095u, (u32)e);
  }
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
      std::vector<u64> cur((size_t)nct);
      u32* ent32 = reinterpret_cast<u32*>(ent.data());
      u64* ent64 = reinterpret_cast<u64*>(ent.data());
      for (i64 rt = t; rt < nrt; rt += T) {
        const i64 r0 = rt * RT, r1 = std::min<i64>(n, r0 + RT);
        if (fmt == FMT_R64) {
          for (i64 ct = 0; ct < nct; ct++) cur[(size_t)ct] = toff4[(size_t)tix(rt, ct)] * 2;      // in quads
          for (i64 i = r0; i < r1; i++) {
            i64 cur_ct = -1; u32 p = 0; u64 w = 0;
            auto flush = [&]() { if (cur_ct >= 0 && p % 4 != 0) ent64[cur[(size_t)cur_ct]++] = w; };
            for (i64 a = rowptr[i]; a < rowptr[i + 1]; a++) {
              const i64 c = col[a];
              if (c == 0) continue;
              const i64 ct = c / CT;
              if (ct != cur_ct) { flush(); cur_ct = ct; p = 0; }
              for (u32 iv = (u32)val[a]; iv > 0; iv--) {
                if (p % 4 == 0) w = r64_null_quad((u32)(i - r0), (u32)cur[(size_t)ct]);
                const int sh = 12 * (int)(p % 4);
                w = (w & ~(0xFFFull << sh)) | ((u64)(c - ct * CT) << sh);
                if (++p % 4 == 0) ent64[cur[(size_t)ct]++] = w;
              }
            }
            flush();
          }
          continue;
        }
        for (i64 ct = 0; ct < nct; ct++) cur[(size_t)ct] = toff4[(size_t)tix(rt, ct)] * 4;
        for (i64 i = r0; i < r1; i++)
          for (i64 a = rowptr[i]; a < rowptr[i + 1]; a++) {
            const i64 c = col[a];
            if (c == 0) continue;
            u32 iv = (u32)val[a];
            const i64 ct = c / CT;
            const u32 base = ((u32)(c - ct * CT) << TILE_BITS) | (u32)(i - r0);
            u64& cu = cur[(size_t)ct];
            if (fmt == FMT_U24) {
              for (; iv > 0; iv--, cu++) { unsigned char* b = &ent[cu * 3]; b[0] = (unsigned char)base; b[1] = (unsigned char)(base >> 8); b[2] = (unsigned char)(base >> 16); }
            } else {
              while (iv > 0) { const u32 q = std::min(iv, 255u); ent32[cu++] = (q << 24) | base; iv -= q; }
            }
          }
      }
    });
    for (auto& x : th) x.join();
  }
  KLR_TRY(dev_alloc(ctx, (void**)&m->t_ent, (size_t)total4 * qbytes));
  KLR_HIP(ctx, hipMemcpy(m->t_ent, ent.data(), (size_t)total4 * qbytes, hipMemcpyHostToDevice));
  m->fmt = fmt;
  m->stored_entries = (i64)stored; m->col_abs_max = amax;
  KLR_TRY(build_pieces_and_upload(ctx, m, toff4, nrt, nct, rt_major));
  return KLR_OK;
}

extern "C" int klr_mat_from_csr(klr_ctx* ctx, int64_t n, int64_t dim, const int64_t* rowptr,
                                const int64_t* col, const double* val, const uint8_t* labels,
                                int64_t n_total, klr_mat** out) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, out != nullptr);
  *out = nullptr;
  KLR_ARG(ctx, n >= 0 && dim >= 1 && rowptr != nullptr && labels != nullptr);
  KLR_ARG(ctx, n < (i64)0xFFFFFFFFll && dim < (i64)0xFFFFFFFFll);
  KLR_ARG(ctx, rowptr[0] == 0);
  if (n_total == 0) n_total = n;
  KLR_ARG(ctx, n_total >= n);
  const i64 nnz = rowptr[n];
  KLR_ARG(ctx, nnz >= 0 && (nnz == 0 || (col != nullptr && val != nullptr)));
  // the row pointers are checked before anything is read or written through them
  for (i64 i = 0; i < n; i++)
    if (rowptr[i] < 0 || rowptr[i] > rowptr[i + 1] || rowptr[i + 1] > nnz)
      return klr_fail(ctx, KLR_EINVAL, "rowptr must be non-decreasing within [0, rowptr[n]] (row %lld: %lld, %lld; nnz %lld)",
                      (long long)i, (long long)rowptr[i], (long long)rowptr[i + 1], (long long)nnz);
  KLR_HIP(ctx, hipSetDevice(ctx->device));

  // ---- validate + per-column counts (threads own contiguous row blocks) ----------------------
  int T = (int)std::min<i64>(std::max<unsigned>(1u, std::min(8u, std::thread::hardware_concurrency())),
                             std::max<i64>(1, n / 4096));
  std::vector<std::vector<u32>> cnt(T, std::vector<u32>((size_t)dim, 0u));
  std::vector<std::vector<u32>> cnz(T, std::vector<u32>((size_t)dim, 0u));
  std::vector<int> bad(T, 0), notf32(T, 0), notcount(T, 0);
  std::vector<i64> n1(T, 0);
  // large inputs also get their narrow copy (u32 column, f32 value) in this pass: if they turn out to be count data,
  // the device builds CSC and tiles from it (KLR_HOST_INGEST=1 keeps everything on the host)
  const bool maybe_device = n >= 2048 && nnz >= (1 << 20) && !ctx_opt(ctx, "KLR_HOST_INGEST");
  // (not value-initialised: a vector would touch 8 bytes per entry on ONE thread before the worker threads fill them)
  std::unique_ptr<u32[]> n_col(maybe_device ? new u32[(size_t)std::max<i64>(nnz, 1)] : nullptr);
  std::unique_ptr<float[]> n_val(maybe_device ? new float[(size_t)std::max<i64>(nnz, 1)] : nullptr);
  auto rows_of = [&](int t, i64& r0, i64& r1) { r0 = n * t / T; r1 = n * (t + 1) / T; };
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
      i64 r0, r1; rows_of(t, r0, r1);
      for (i64 i = r0; i < r1; i++) {
        const i64 b = rowptr[i], e = rowptr[i + 1];
        if (labels[i]) n1[t]++;
        // LinearPdf asserts i[0] == 0 (kmerLr_logistic_regression.go:53); bias value is 1.0
        if (e <= b || col[b] != 0 || val[b] != 1.0) { bad[t] = 1; continue; }
        i64 prev = -1;
        for (i64 a = b; a < e; a++) {
          const i64 c = col[a];
          if (c <= prev || c >= dim) { bad[t] = 2; break; }
          prev = c;
          cnt[t][c]++;
          const double v = val[a];
          if (v != 0.0) cnz[t][c]++;
          if ((double)(float)v != v) notf32[t] = 1;
          if (!(v >= 1.0 && v <= 16777216.0 && v == (double)(u32)v)) notcount[t] = 1;
          if (maybe_device) { n_col[(size_t)a] = (u32)c; n_val[(size_t)a] = (float)v; }
        }
      }
    });
    for (auto& x : th) x.join();
  }
  for (int t = 0; t < T; t++) {
    if (bad[t] == 1) return klr_fail(ctx, KLR_EINVAL, "every row must start with the bias entry (column 0, value 1.0)");
    if (bad[t] == 2) return klr_fail(ctx, KLR_EINVAL, "column ids must be strictly ascending within a row and < dim");
  }
  klr_mat* m = new klr_mat();
  m->ctx = ctx; m->device = ctx->device; m->n = n; m->dim = dim; m->nnz = nnz; m->n_total = n_total;
  m->n_total_all = n_total; m->h_labels.assign(labels, labels + n);
  m->val_f64 = false;
  bool counts = true;
  for (int t = 0; t < T; t++) { if (notf32[t]) m->val_f64 = true; if (notcount[t]) counts = false; m->n1_local += n1[t]; }
  m->avg_row = n ? (double)nnz / (double)n : 0.0;
  if (maybe_device && counts) {
    // ---- device route: upload the narrow CSR, everything else is built there (klr_ingest.hip) -------------------
    int rc = KLR_OK;
    do {
      if ((rc = dev_alloc(ctx, (void**)&m->rowptr, (size_t)(n + 1) * sizeof(i64)))) break;
      if ((rc = dev_alloc(ctx, (void**)&m->col, (size_t)nnz * sizeof(u32)))) break;
      if ((rc = dev_alloc(ctx, &m->val, (size_t)nnz * 4))) break;
      if ((rc = dev_alloc(ctx, (void**)&m->labels, (size_t)n))) break;
      hipError_t e = hipMemcpy(m->rowptr, rowptr, (size_t)(n + 1) * sizeof(i64), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(m->col, n_col.get(), (size_t)nnz * sizeof(u32), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(m->val, n_val.get(), (size_t)nnz * 4, hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(m->labels, labels, (size_t)n, hipMemcpyHostToDevice);
      if (e != hipSuccess) { rc = klr_fail(ctx, KLR_EHIP, "matrix upload failed: %s", hipGetErrorString(e)); break; }
      rc = device_finish_csr(ctx, m);
    } while (0);
    if (rc != KLR_OK) { std::string keep = ctx->err; mat_free(m); ctx->err = keep; return rc; }
    *out = m;
    return KLR_OK;
  }
  m->h_colptr.assign((size_t)dim + 1, 0);
  m->h_collen.assign((size_t)dim, 0);
  m->h_colnz.assign((size_t)dim, 0);
  for (i64 j = 0; j < dim; j++) {
    u32 c = 0, z = 0;
    for (int t = 0; t < T; t++) { c += cnt[t][j]; z += cnz[t][j]; }
    m->h_collen[j] = c; m->h_colnz[j] = z;
    if (z != c) m->has_explicit_zero = true;
    m->h_colptr[j + 1] = m->h_colptr[j] + (i64)((c + 3u) & ~3u);
  }
  m->nnz_pad = m->h_colptr[dim];

  // ---- host CSC (stable: rows ascending inside a column) + narrow CSR ------------------------
  const size_t vsz = m->val_f64 ? 8 : 4;
  std::vector<u32> h_col((size_t)std::max<i64>(nnz, 1));
  std::vector<char> h_val((size_t)std::max<i64>(nnz, 1) * vsz);
  std::vector<u32> h_crow((size_t)std::max<i64>(m->nnz_pad, 1), 0u);
  std::vector<char> h_cval((size_t)std::max<i64>(m->nnz_pad, 1) * vsz, 0);
  {
    // cursor[t][j] = first slot of thread t in column j
    std::vector<std::vector<i64>> cur(T, std::vector<i64>((size_t)dim));
    for (i64 j = 0; j < dim; j++) {
      i64 o = m->h_colptr[j];
      for (int t = 0; t < T; t++) { cur[t][j] = o; o += cnt[t][j]; }
    }
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
      i64 r0, r1; rows_of(t, r0, r1);
      std::vector<i64>& cu = cur[t];
      for (i64 i = r0; i < r1; i++) {
        for (i64 a = rowptr[i]; a < rowptr[i + 1]; a++) {
          const i64 c = col[a];
          const i64 s = cu[c]++;
          h_col[a] = (u32)c;
          h_crow[s] = (u32)i;
          if (m->val_f64) { ((double*)h_val.data())[a] = val[a]; ((double*)h_cval.data())[s] = val[a]; }
          else            { ((float*)h_val.data())[a] = (float)val[a]; ((float*)h_cval.data())[s] = (float)val[a]; }
        }
      }
    });
    for (auto& x : th) x.join();
  }

  // ---- upload ----------------------------------------------------------------------------------
  int rc = KLR_OK;
  do {
    if ((rc = dev_alloc(ctx, (void**)&m->rowptr, (size_t)(n + 1) * sizeof(i64)))) break;
    if ((rc = dev_alloc(ctx, (void**)&m->col, (size_t)nnz * sizeof(u32)))) break;
    if ((rc = dev_alloc(ctx, &m->val, (size_t)nnz * vsz))) break;
    if ((rc = dev_alloc(ctx, (void**)&m->crow, (size_t)m->nnz_pad * sizeof(u32)))) break;
    if ((rc = dev_alloc(ctx, &m->cval, (size_t)m->nnz_pad * vsz))) break;
    if ((rc = dev_alloc(ctx, (void**)&m->labels, (size_t)n))) break;
    hipError_t e = hipSuccess;
    e = hipMemcpy(m->rowptr, rowptr, (size_t)(n + 1) * sizeof(i64), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz) e = hipMemcpy(m->col, h_col.data(), (size_t)nnz * sizeof(u32), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz) e = hipMemcpy(m->val, h_val.data(), (size_t)nnz * vsz, hipMemcpyHostToDevice);
    if (e == hipSuccess && m->nnz_pad) e = hipMemcpy(m->crow, h_crow.data(), (size_t)m->nnz_pad * sizeof(u32), hipMemcpyHostToDevice);
    if (e == hipSuccess && m->nnz_pad) e = hipMemcpy(m->cval, h_cval.data(), (size_t)m->nnz_pad * vsz, hipMemcpyHostToDevice);
    if (e == hipSuccess && n) e = hipMemcpy(m->labels, labels, (size_t)n, hipMemcpyHostToDevice);
    if (e != hipSuccess) { rc = klr_fail(ctx, KLR_EHIP, "matrix upload failed: %s", hipGetErrorString(e)); break; }
    rc = mat_build_chunks_and_upload(ctx, m);
    if (rc == KLR_OK && !m->val_f64) rc = build_tiled(ctx, m, rowptr, col, val);
  } while (0);
  if (rc != KLR_OK) { std::string keep = ctx->err; mat_free(m); ctx->err = keep; return rc; }
  *out = m;
  return KLR_OK;
}

extern "C" int klr_mat_to_csr(klr_ctx* ctx, const klr_mat* mat, int64_t* rowptr, int64_t* col, double* val) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, mat != nullptr);
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  KLR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (rowptr) KLR_HIP(ctx, hipMemcpy(rowptr, mat->rowptr, (size_t)(mat->n + 1) * sizeof(i64), hipMemcpyDeviceToHost));
  if (col && mat->nnz) {
    std::vector<u32> t((size_t)mat->nnz);
    KLR_HIP(ctx, hipMemcpy(t.data(), mat->col, (size_t)mat->nnz * sizeof(u32), hipMemcpyDeviceToHost));
    for (i64 a = 0; a < mat->nnz; a++) col[a] = (i64)t[a];
  }
  if (val && mat->nnz) {
    if (mat->val_f64) KLR_HIP(ctx, hipMemcpy(val, mat->val, (size_t)mat->nnz * 8, hipMemcpyDeviceToHost));
    else {
      std::vector<float> t((size_t)mat->nnz);
      KLR_HIP(ctx, hipMemcpy(t.data(), mat->val, (size_t)mat->nnz * 4, hipMemcpyDeviceToHost));
      for (i64 a = 0; a < mat->nnz; a++) val[a] = (double)t[a];
    }
  }
  return KLR_OK;
}

extern "C" int klr_mat_set_active_rows(klr_ctx* ctx, klr_mat* mat, const uint8_t* mask, int64_t n_active_total) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, mat != nullptr);
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  KLR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (!mask) {
    if (mat->active && !mat->arena) (void)hipFree(mat->active);
    mat->active = nullptr;
    mat->n_total = mat->n_total_all;
    mat->n1_local = 0;
    for (uint8_t l : mat->h_labels) if (l) mat->n1_local++;
    mat->n_active_local = mat->n;
    return KLR_OK;
  }
  KLR_ARG(ctx, n_active_total > 0);
  if (!mat->active) {
    if (mat->arena) mat->active = mat->active_slot;
    else KLR_TRY(dev_alloc(ctx, (void**)&mat->active, (size_t)std::max<i64>(mat->n, 1)));
  }
  KLR_HIP(ctx, hipMemcpy(mat->active, mask, (size_t)mat->n, hipMemcpyHostToDevice));
  i64 na = 0, n1 = 0;
  for (i64 i = 0; i < mat->n; i++) if (mask[i]) { na++; if (mat->h_labels[(size_t)i]) n1++; }
  mat->n_active_local = na; mat->n1_local = n1;
  mat->n_total = n_active_total;
  return KLR_OK;
}

// compute_class_weights, kmerLr_data.go:178-193, over all shards
extern "C" int klr_class_weights(klr_ctx* ctx, const klr_mat* mat, double cw[2]) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, mat != nullptr && cw != nullptr);
  const i64 nloc = mat->active ? mat->n_active_local : mat->n;
  double cnts[2] = {(double)(nloc - mat->n1_local), (double)mat->n1_local};
  if (ctx->ar_fn) {
    double* d;
    KLR_TRY(ws_get(ctx, WS_G, 2 * sizeof(double), (void**)&d));
    KLR_TRY(h2d(ctx, d, cnts, sizeof(cnts)));
    KLR_TRY(klr_allreduce(ctx, d, 2, KLR_F64));
    KLR_TRY(d2h_sync(ctx, cnts, d, sizeof(cnts)));
  }
  const double n0 = cnts[0], n1 = cnts[1];
  cw[0] = (n0 + n1) / (2.0 * n0);
  cw[1] = (n0 + n1) / (2.0 * n1);
  return KLR_OK;
}

// =================================================================================================
// device exclusive scan (u32 counts -> i64 offsets), three tiny kernels
// =================================================================================================
#define SCAN_BLOCK 1024
#define SCAN_ITEMS 4

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_block_sums(const u32* __restrict__ in, i64 n, i64* __restrict__ bsum) {
  __shared__ i64 sh[SCAN_BLOCK / 64];
  const i64 base = (i64)blockIdx.x * SCAN_BLOCK * SCAN_ITEMS;
  i64 s = 0;
  for (int k = 0; k < SCAN_ITEMS; k++) {
    const i64 i = base + (i64)threadIdx.x * SCAN_ITEMS + k;
    if (i < n) s += in[i];
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    i64 t = 0;
    for (int w = 0; w < SCAN_BLOCK / 64; w++) t += sh[w];
    bsum[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(1024) void k_scan_bsums(i64* bsum, i64 nb) {
  // single block, serial over tiles of 1024 (nb is small: n / 4096)
  __shared__ i64 sh[1024];
  __shared__ i64 carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (i64 base = 0; base < nb; base += 1024) {
    const i64 i = base + threadIdx.x;
    const i64 v = i < nb ? bsum[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      i64 t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    const i64 incl = sh[threadIdx.x];
    if (i < nb) bsum[i] = carry + incl - v;   // exclusive
    __syncthreads();
    if (threadIdx.x == 1023) carry += incl;
    __syncthreads();
  }
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_apply(const u32* __restrict__ in, i64 n, const i64* __restrict__ bsum,
                                                          i64* __restrict__ out) {
  __shared__ i64 sh[SCAN_BLOCK];
  const i64 base = (i64)blockIdx.x * SCAN_BLOCK * SCAN_ITEMS;
  u32 v[SCAN_ITEMS];
  i64 s = 0;
  for (int k = 0; k < SCAN_ITEMS; k++) {
    const i64 i = base + (i64)threadIdx.x * SCAN_ITEMS + k;
    v[k] = i < n ? in[i] : 0u;
    s += v[k];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < SCAN_BLOCK; o <<= 1) {
    i64 t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  i64 run = bsum[blockIdx.x] + sh[threadIdx.x] - s;
  for (int k = 0; k < SCAN_ITEMS; k++) {
    const i64 i = base + (i64)threadIdx.x * SCAN_ITEMS + k;
    if (i < n) out[i] = run;
    run += v[k];
    if (i == n - 1) out[n] = run;
  }
}

int launch_exscan_u32(klr_ctx* ctx, const u32* in, i64 n, i64* out) {
  if (n == 0) {
    KLR_HIP(ctx, hipMemsetAsync(out, 0, sizeof(i64), ctx->stream));
    return KLR_OK;
  }
  const i64 nb = ceil_div(n, (i64)SCAN_BLOCK * SCAN_ITEMS);
  i64* bsum;
  KLR_TRY(ws_get(ctx, WS_TMP2, (size_t)nb * sizeof(i64), (void**)&bsum));
  hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, ctx->stream, in, n, bsum);
  hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(1024), 0, ctx->stream, bsum, nb);
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, ctx->stream, in, n, bsum, out);
  KLR_HIP(ctx, hipGetLastError());
  return KLR_OK;
}

// =================================================================================================
// column sub-select: featureSelection.Data (kmerLr_feature_selection.go:309-343) + Transform.Apply
// (kmerLr_transform.go:584-629), built from the CSC side so only the selected columns are read.
// =================================================================================================
// Flat over the padded entries of the reduced CSC (one thread per entry, its column by binary search in the
// LDS copy of the column offsets): a block per column left the bias column (n entries) to one block -- 0.9 ms of
// a 1 ms sub-select at configs[2].  RED_MASK_WORDS_MAX*64 selected columns or fewer: every row keeps a bit mask of
// the selected columns it holds (atomicOr: order free), from which both the row lengths and the position of an
// entry inside its row (= the set bits below its column) follow -- rows come out sorted by column without a sort
// and without a cursor.
#define RED_MASK_WORDS_MAX 16
#define RED_LDS_COLS 4096

__device__ __forceinline__ int red_find_col(const i64* __restrict__ cp, int c1, i64 e) {
  int lo = 0, hi = c1;                                  // largest p with cp[p] <= e
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cp[mid] <= e) lo = mid; else hi = mid; }
  return lo;
}

// GATHER: copy (and scale) the source column entries; else the reduced CSC is already filled (explicit-zero path)
// and only the row marks are made.  mask != null: bit p of row r; else cnt[r]++.
template <typename VT, bool GATHER>
__global__ __launch_bounds__(256) void k_red_gather(const i64* __restrict__ src_colptr, const u32* __restrict__ crow,
                                                    const VT* __restrict__ cval, const i64* __restrict__ sel,
                                                    const i64* __restrict__ dst_colptr, const u32* __restrict__ dst_len, int c1,
                                                    const double* __restrict__ scale, u32* __restrict__ rcrow,
                                                    double* __restrict__ rcval, unsigned long long* __restrict__ mask, int W,
                                                    u32* __restrict__ cnt) {
  extern __shared__ i64 red_cp[];
  const bool in_lds = c1 + 1 <= RED_LDS_COLS;
  if (in_lds) { for (int t = threadIdx.x; t <= c1; t += 256) red_cp[t] = dst_colptr[t]; __syncthreads(); }
  const i64* cp = in_lds ? red_cp : dst_colptr;
  const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
  if (e >= cp[c1]) return;
  const int p = red_find_col(cp, c1, e);
  const i64 off = e - cp[p];
  u32 r = 0;
  if (off < (i64)dst_len[p]) {
    if (GATHER) {
      const i64 sb = src_colptr[sel[p]];
      r = crow[sb + off];
      rcrow[e] = r; rcval[e] = (double)cval[sb + off] * (scale ? scale[p] : 1.0);
    } else r = rcrow[e];
    if (mask) atomicOr(&mask[(i64)r * W + (p >> 6)], 1ull << (p & 63));
    else atomicAdd(&cnt[r], 1u);
  } else if (GATHER) { rcrow[e] = 0u; rcval[e] = 0.0; }
}

__global__ __launch_bounds__(256) void k_red_count(const unsigned long long* __restrict__ mask, int W, i64 n, u32* __restrict__ cnt) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  u32 c = 0;
  for (int w = 0; w < W; w++) c += (u32)__popcll(mask[i * W + w]);
  cnt[i] = c;
}

// CSR side: entry (r, p) goes to rowptr[r] + (selected columns below p in row r) -- or, without masks, to the next
// free slot of the row (k_sort_rows then orders the row)
__global__ __launch_bounds__(256) void k_red_fill(const i64* __restrict__ dst_colptr, const u32* __restrict__ dst_len, int c1,
                                                  const u32* __restrict__ rcrow, const double* __restrict__ rcval,
                                                  const i64* __restrict__ rrowptr, const unsigned long long* __restrict__ mask,
                                                  int W, u32* __restrict__ cursor, u32* __restrict__ rcol, double* __restrict__ rval) {
  extern __shared__ i64 red_cp[];
  const bool in_lds = c1 + 1 <= RED_LDS_COLS;
  if (in_lds) { for (int t = threadIdx.x; t <= c1; t += 256) red_cp[t] = dst_colptr[t]; __syncthreads(); }
  const i64* cp = in_lds ? red_cp : dst_colptr;
  const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
  if (e >= cp[c1]) return;
  const int p = red_find_col(cp, c1, e);
  if (e - cp[p] >= (i64)dst_len[p]) return;
  const u32 r = rcrow[e];
  i64 slot = rrowptr[r];
  if (mask) {
    const unsigned long long* mr = mask + (i64)r * W;
    int below = 0;
    for (int w = 0; w < (p >> 6); w++) below += __popcll(mr[w]);
    below += __popcll(mr[p >> 6] & ((1ull << (p & 63)) - 1ull));
    slot += below;
  } else slot += (i64)atomicAdd(&cursor[r], 1u);
  rcol[slot] = (u32)p;
  rval[slot] = rcval[e];
}

template <typename VT>
__global__ __launch_bounds__(64) void k_gather_cols_nz(const i64* __restrict__ src_colptr, const u32* __restrict__ src_len,
                                                       const u32* __restrict__ crow, const VT* __restrict__ cval,
                                                       const i64* __restrict__ sel, const i64* __restrict__ dst_colptr,
                                                       const double* __restrict__ scale, u32* __restrict__ rcrow,
                                                       double* __restrict__ rcval) {
  // one wave per selected column; keeps entries with value != 0 in order (:332 `value != 0.0`)
  const int p = blockIdx.x;
  const int lane = threadIdx.x;
  const i64 j = sel[p];
  const i64 sb = src_colptr[j];
  const i64 len = src_len[p];     // TRUE source length here
  const i64 db = dst_colptr[p], de = dst_colptr[p + 1];
  const double sc = scale ? scale[p] : 1.0;
  i64 o = db;
  for (i64 e0 = 0; e0 < len; e0 += 64) {
    const i64 e = e0 + lane;
    double v = 0.0; u32 r = 0;
    if (e < len) { v = (double)cval[sb + e]; r = crow[sb + e]; }
    const bool keep = (e < len) && (v != 0.0);
    const unsigned long long mask = __ballot(keep);
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    if (keep) { rcrow[o + rank] = r; rcval[o + rank] = v * sc; }
    o += __popcll(mask);
  }
  for (i64 e = o + lane; e < de; e += 64) { rcrow[e] = 0u; rcval[e] = 0.0; }
}

// ---- row-block binned variant (<= 1024 selected columns): no global atomics, no scattered stores, no scan launches --
// The flat kernels above spend their time on random 8-byte writes (1M atomicOr in 27 us, 2M scattered stores in 29 us
// at configs[2]).  Here a workgroup owns RB_ROWS consecutive rows: the part of every selected column that falls in
// its rows is found by binary search (the rows of a column ascend), the row masks are built with LDS atomics, and the
// rows' entries are placed in an LDS image of the block's CSR segment that is written out in one coalesced pass.
// (Needs the rows of a CSC column in ascending order: both ingest routes give that -- the host build walks the rows
// in order, the device build sorts (column, entry) pairs with a stable radix sort.)
//   k_red_bin_a: searches, reduced CSC copy, masks -> global masks, per-block entry totals, segment table
//   k_red_bin_b: block base = sum of the totals before it, row offsets by a block scan, CSR segment via LDS
#define RB_ROWS 256
#define RB_WMAX 16                      // mask words per row
#define RB_CMAX 1024                    // selected columns (one search thread each)
#define RB_TCAP 3584                    // entries of a block staged in LDS (more: scattered stores for that block)

__device__ __forceinline__ u32 rb_lower_bound(const u32* __restrict__ a, u32 len, u32 key) {   // first index with a[i] >= key
  u32 lo = 0, hi = len;
  while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}

// exclusive prefix of a[0 .. c) in place, a[c] = total; one wavefront, RB_CMAX / 64 consecutive values per lane
__device__ __forceinline__ void rb_prefix(u32* a, int c, int lane) {
  constexpr int PL = RB_CMAX / 64;
  u32 v[PL], sum = 0;
#pragma unroll
  for (int u = 0; u < PL; u++) { const int i = lane * PL + u; v[u] = i < c ? a[i] : 0u; sum += v[u]; }
  u32 incl = sum;
  for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
  u32 acc = incl - sum;
#pragma unroll
  for (int u = 0; u < PL; u++) { const int i = lane * PL + u; if (i < c) a[i] = acc; acc += v[u]; }
  if (lane == 63) a[c] = incl;
}

template <typename VT>
__global__ __launch_bounds__(1024) void k_red_bin_a(const i64* __restrict__ src_colptr, const u32* __restrict__ crow,
                                                    const VT* __restrict__ cval, const i64* __restrict__ sel,
                                                    const i64* __restrict__ dst_colptr, const u32* __restrict__ dst_len, int c1,
                                                    const double* __restrict__ scale, i64 n, int W,
                                                    u32* __restrict__ rcrow, double* __restrict__ rcval,
                                                    unsigned long long* __restrict__ gmask, u32* __restrict__ blocktot,
                                                    u32* __restrict__ seg /* nblk x c1 x 2 */) {
  __shared__ unsigned long long smask[RB_ROWS * RB_WMAX];
  __shared__ u32 s_lo[RB_CMAX], s_pre[RB_CMAX + 1];
  __shared__ i64 s_src[RB_CMAX], s_dst[RB_CMAX];       // source / destination offset of the segment's first entry
  __shared__ double s_sc[RB_CMAX];
  __shared__ u32 s_wsum[16];
  const int tid = threadIdx.x;
  const i64 r0 = (i64)blockIdx.x * RB_ROWS;
  const int nr = (int)(n - r0 < RB_ROWS ? n - r0 : RB_ROWS);
  for (int t = tid; t < RB_ROWS * W; t += 1024) smask[t] = 0ull;
  if (tid < c1) {
    const i64 sb = src_colptr[sel[tid]];
    const u32 len = dst_len[tid];
    const u32 lo = rb_lower_bound(crow + sb, len, (u32)r0);
    const u32 hi = rb_lower_bound(crow + sb, len, (u32)(r0 + nr));
    s_lo[tid] = lo; s_pre[tid] = hi - lo;
    s_src[tid] = sb + lo; s_dst[tid] = dst_colptr[tid] + lo; s_sc[tid] = scale ? scale[tid] : 1.0;
    seg[((size_t)blockIdx.x * c1 + tid) * 2] = lo; seg[((size_t)blockIdx.x * c1 + tid) * 2 + 1] = hi - lo;
  }
  __syncthreads();
  if (tid < 64) rb_prefix(s_pre, c1, tid);             // exclusive prefix of the segment lengths (c1 <= 256)
  __syncthreads();
  const u32 T = s_pre[c1];
  for (u32 e = tid; e < T; e += 1024) {
    int lo = 0, hi = c1;                               // column: largest p with s_pre[p] <= e
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_pre[mid] <= e) lo = mid; else hi = mid; }
    const int p = lo;
    const u32 k = e - s_pre[p];                        // every per-column quantity from LDS: one round trip per entry
    const i64 se = s_src[p] + k, de = s_dst[p] + k;
    const u32 r = crow[se];
    rcrow[de] = r; rcval[de] = (double)cval[se] * s_sc[p];
    atomicOr(&smask[(size_t)(r - (u32)r0) * W + (p >> 6)], 1ull << (p & 63));
  }
  // padding of the reduced CSC columns (to multiples of 4 entries): block 0
  if (blockIdx.x == 0 && tid < c1) {
    const i64 db = dst_colptr[tid], de = dst_colptr[tid + 1];
    for (i64 e = db + dst_len[tid]; e < de; e++) { rcrow[e] = 0u; rcval[e] = 0.0; }
  }
  __syncthreads();
  u32 cntr = 0;
  if (tid < nr) for (int w = 0; w < W; w++) cntr += (u32)__popcll(smask[(size_t)tid * W + w]);
  for (int t = tid; t < nr * W; t += 1024) gmask[(size_t)r0 * W + t] = smask[t];
  for (int o = 32; o > 0; o >>= 1) cntr += __shfl_xor(cntr, o, 64);
  if ((tid & 63) == 0) s_wsum[tid >> 6] = cntr;
  __syncthreads();
  if (tid == 0) { u32 acc = 0; for (int w = 0; w < 16; w++) acc += s_wsum[w]; blocktot[blockIdx.x] = acc; }
}

__global__ __launch_bounds__(1024) void k_red_bin_b(const i64* __restrict__ dst_colptr, int c1, i64 n, int W,
                                                    const u32* __restrict__ rcrow, const double* __restrict__ rcval,
                                                    const unsigned long long* __restrict__ gmask, const u32* __restrict__ blocktot,
                                                    const u32* __restrict__ seg, i64* __restrict__ rrowptr,
                                                    u32* __restrict__ rcol, double* __restrict__ rval) {
  extern __shared__ unsigned char rb_lds[];
  unsigned long long* smask = reinterpret_cast<unsigned long long*>(rb_lds);              // RB_ROWS * W
  double* sval = reinterpret_cast<double*>(smask + (size_t)RB_ROWS * W);                  // RB_TCAP
  u32* scol = reinterpret_cast<u32*>(sval + RB_TCAP);                                     // RB_TCAP
  u32* s_off = scol + RB_TCAP;                                                            // RB_ROWS + 1: row offsets inside the block
  __shared__ u32 s_pre[RB_CMAX + 1];
  __shared__ i64 s_dst[RB_CMAX];
  __shared__ u32 s_wsum[17];
  __shared__ i64 s_base;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const i64 r0 = (i64)blockIdx.x * RB_ROWS;
  const int nr = (int)(n - r0 < RB_ROWS ? n - r0 : RB_ROWS);
  // block base: entries of all blocks before this one
  if (w == 0) {
    i64 acc = 0;
    for (int b = lane; b < (int)blockIdx.x; b += 64) acc += blocktot[b];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) s_base = acc;
  }
  for (int t = tid; t < nr * W; t += 1024) smask[t] = gmask[(size_t)r0 * W + t];
  if (tid < c1) { s_dst[tid] = dst_colptr[tid] + seg[((size_t)blockIdx.x * c1 + tid) * 2]; s_pre[tid] = seg[((size_t)blockIdx.x * c1 + tid) * 2 + 1]; }
  __syncthreads();
  if (tid < 64) rb_prefix(s_pre, c1, tid);             // exclusive prefix of the segment lengths (c1 <= 256)
  // exclusive scan of the row counts over the block (one row per thread)
  u32 cntr = 0;
  if (tid < nr) for (int ww = 0; ww < W; ww++) cntr += (u32)__popcll(smask[(size_t)tid * W + ww]);
  u32 incl = cntr;
  for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
  if (lane == 63) s_wsum[w] = incl;
  __syncthreads();
  if (tid == 0) { u32 acc = 0; for (int t = 0; t < 16; t++) { const u32 v = s_wsum[t]; s_wsum[t] = acc; acc += v; } s_wsum[16] = acc; }
  __syncthreads();
  const u32 excl = s_wsum[w] + incl - cntr;
  const u32 T = s_wsum[16];
  const i64 base = s_base;
  if (tid < RB_ROWS) s_off[tid] = excl;
  if (tid < nr) rrowptr[r0 + tid] = base + excl;
  if (r0 + nr == n && tid == 0) rrowptr[n] = base + T;
  __syncthreads();
  const bool staged = T <= RB_TCAP;
  const u32 TE = s_pre[c1];                            // == T
  for (u32 e = tid; e < TE; e += 1024) {
    int lo = 0, hi = c1;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_pre[mid] <= e) lo = mid; else hi = mid; }
    const int p = lo;
    const i64 de = s_dst[p] + (e - s_pre[p]);
    const double v = rcval[de];
    const u32 rl = rcrow[de] - (u32)r0;
    const unsigned long long* mr = smask + (size_t)rl * W;
    int below = 0;
    for (int ww = 0; ww < (p >> 6); ww++) below += __popcll(mr[ww]);
    below += __popcll(mr[p >> 6] & ((1ull << (p & 63)) - 1ull));
    const u32 slot = s_off[rl] + (u32)below;
    if (staged) { scol[slot] = (u32)p; sval[slot] = v; }
    else { rcol[base + slot] = (u32)p; rval[base + slot] = v; }
  }
  if (staged) {
    __syncthreads();
    for (u32 t = tid; t < T; t += 1024) { rcol[base + t] = scol[t]; rval[base + t] = sval[t]; }
  }
}

// in-row insertion sort by (distinct) reduced column id -> deterministic, ascending rows
__global__ __launch_bounds__(256) void k_sort_rows(const i64* __restrict__ rrowptr, i64 n, u32* __restrict__ rcol,
                                                   double* __restrict__ rval) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const i64 b = rrowptr[i], e = rrowptr[i + 1];
  for (i64 a = b + 1; a < e; a++) {
    const u32 kc = rcol[a]; const double kv = rval[a];
    i64 q = a - 1;
    while (q >= b && rcol[q] > kc) { rcol[q + 1] = rcol[q]; rval[q + 1] = rval[q]; q--; }
    rcol[q + 1] = kc; rval[q + 1] = kv;
  }
}

// dense variants for offset transforms (Transform.Apply densifies, kmerLr_transform.go:596-609)
__global__ __launch_bounds__(256) void k_dense_init(i64 n, int c1, i64 colstride, const double* __restrict__ scale,
                                                    const double* __restrict__ offset, i64* __restrict__ rrowptr,
                                                    u32* __restrict__ rcol, double* __restrict__ rval,
                                                    u32* __restrict__ rcrow, double* __restrict__ rcval) {
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t <= n && t < n + 1) rrowptr[t] = t * (i64)c1;
  if (t < n * (i64)c1) {
    const i64 i = t / c1; const int p = (int)(t % c1);
    double v = 0.0 - offset[p];
    if (scale) v = v * scale[p];
    rcol[t] = (u32)p; rval[t] = v;
    rcrow[(i64)p * colstride + i] = (u32)i; rcval[(i64)p * colstride + i] = v;
  }
}

template <typename VT>
__global__ __launch_bounds__(256) void k_dense_overwrite(const i64* __restrict__ src_colptr, const u32* __restrict__ src_len,
                                                         const u32* __restrict__ crow, const VT* __restrict__ cval,
                                                         const i64* __restrict__ sel, int c1, i64 colstride,
                                                         const double* __restrict__ scale, const double* __restrict__ offset,
                                                         double* __restrict__ rval, double* __restrict__ rcval) {
  const int p = blockIdx.x;
  const i64 sb = src_colptr[sel[p]];
  const i64 len = src_len[p];
  for (i64 e = threadIdx.x; e < len; e += 256) {
    const i64 i = crow[sb + e];
    double v = (double)cval[sb + e] - offset[p];
    if (scale) v = v * scale[p];
    rval[i * c1 + p] = v;
    rcval[(i64)p * colstride + i] = v;
  }
}

int build_reduced(klr_ctx* ctx, const klr_mat* full, const i64* cols, i64 c1, const double* scale,
                  const double* offset, bool need_csc, klr_mat** out) {
  (void)need_csc;
  *out = nullptr;
  KLR_ARG(ctx, c1 >= 1 && cols[0] == 0);
  for (i64 p = 1; p < c1; p++) KLR_ARG(ctx, cols[p] > cols[p - 1] && cols[p] < full->dim);
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  const i64 n = full->n;
  klr_mat* r = new klr_mat();
  r->ctx = ctx; r->device = ctx->device; r->n = n; r->dim = c1; r->n_total = full->n_total; r->val_f64 = true;
  r->n1_local = full->n1_local; r->n_total_all = full->n_total_all; r->h_labels = full->h_labels;
  r->n_active_local = full->n_active_local;
  r->h_colptr.assign((size_t)c1 + 1, 0);
  r->h_collen.assign((size_t)c1, 0);
  r->h_colnz.assign((size_t)c1, 0);
  const bool dense = offset != nullptr;
  const i64 colstride = (n + 3) & ~(i64)3;
  bool any_zero = false;
  for (i64 p = 0; p < c1; p++) {
    const i64 j = cols[p];
    const u32 len = dense ? (u32)n : full->h_colnz[j];
    if (full->h_colnz[j] != full->h_collen[j]) any_zero = true;
    r->h_collen[p] = len; r->h_colnz[p] = len;
    r->h_colptr[p + 1] = r->h_colptr[p] + (dense ? colstride : (i64)((len + 3u) & ~3u));
    r->nnz += len;
  }
  r->nnz_pad = r->h_colptr[c1];
  r->avg_row = n ? (double)r->nnz / (double)n : 0.0;
  // chunks of the column sums (k_xt_s, k_irls_cols): columns cut at RED_CHUNK entries -- short chunks: the sums over
  // a reduced matrix are latency bound (a lane group walks its chunk in dependent gather round trips), and the
  // bias column alone is n entries
  const i64 RED_CHUNK = 512;
  i64 nch = 0;
  for (i64 p = 0; p < c1; p++) nch += ceil_div(r->h_colptr[p + 1] - r->h_colptr[p], RED_CHUNK);
  r->nchunks = nch;
  r->group = pick_group(nch ? (double)r->nnz_pad / (double)nch : 4.0);

  // ---- ONE pooled block for every device array of the matrix, ONE pinned staging copy for the host tables:
  // (eleven hipMalloc + as many hipFree and nine pageable copies per epoch were 1 ms of host time)
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t nn = (size_t)std::max<i64>(n, 1), nz = (size_t)std::max<i64>(r->nnz, 1), nzp = (size_t)std::max<i64>(r->nnz_pad, 1);
  const size_t nchs = (size_t)std::max<i64>(nch, 1);
  size_t o = 0;
  const size_t o_rowptr = o; o += al((nn + 1) * 8);
  const size_t o_col = o;    o += al(nz * 4);
  const size_t o_val = o;    o += al(nz * 8);
  const size_t o_crow = o;   o += al(nzp * 4);
  const size_t o_cval = o;   o += al(nzp * 8);
  const size_t o_lab = o;    o += al(nn);
  const size_t o_act = o;    o += al(nn);
  // tables, in staging order: colptr | colchunk | chunk_begin | sel | scale | offset | chunk_len | src_len | dst_len
  const size_t o_tab = o;
  size_t t = 0;
  const size_t t_colptr = t;   t += (size_t)(c1 + 1) * 8;
  const size_t t_colchunk = t; t += (size_t)(c1 + 1) * 8;
  const size_t t_cbegin = t;   t += nchs * 8;
  const size_t t_sel = t;      t += (size_t)c1 * 8;
  const size_t t_scale = t;    t += (size_t)c1 * 8;
  const size_t t_off = t;      t += (size_t)c1 * 8;
  const size_t t_clen = t;     t += nchs * 4;
  const size_t t_srclen = t;   t += (size_t)c1 * 4;
  const size_t t_dstlen = t;   t += (size_t)c1 * 4;
  const size_t tab_bytes = t;
  o += al(tab_bytes);

  int rc = KLR_OK;
  do {
    if ((rc = pool_get(ctx, o, &r->arena, &r->arena_cap))) break;
    char* base = (char*)r->arena;
    r->rowptr = (i64*)(base + o_rowptr); r->col = (u32*)(base + o_col); r->val = base + o_val;
    r->crow = (u32*)(base + o_crow); r->cval = base + o_cval;
    r->labels = (uint8_t*)(base + o_lab); r->active_slot = (uint8_t*)(base + o_act);
    char* dtab = base + o_tab;
    r->colptr = (i64*)(dtab + t_colptr); r->colchunk = (i64*)(dtab + t_colchunk);
    r->chunk_begin = (i64*)(dtab + t_cbegin); r->chunk_len = (u32*)(dtab + t_clen);
    const i64* d_sel = (const i64*)(dtab + t_sel);
    const double* d_scale = (const double*)(dtab + t_scale);
    const double* d_off = (const double*)(dtab + t_off);
    const u32* d_srclen = (const u32*)(dtab + t_srclen);
    const u32* d_dstlen = (const u32*)(dtab + t_dstlen);
    r->d_collen = (u32*)(dtab + t_dstlen);
    char* htab;
    if ((rc = stage_get(ctx, tab_bytes, (void**)&htab))) break;
    {
      i64* h_colptr = (i64*)(htab + t_colptr); i64* h_colchunk = (i64*)(htab + t_colchunk);
      i64* h_cbegin = (i64*)(htab + t_cbegin); u32* h_clen = (u32*)(htab + t_clen);
      i64* h_sel = (i64*)(htab + t_sel); double* h_scale = (double*)(htab + t_scale); double* h_off = (double*)(htab + t_off);
      u32* h_srclen = (u32*)(htab + t_srclen); u32* h_dstlen = (u32*)(htab + t_dstlen);
      i64 k = 0;
      for (i64 p = 0; p < c1; p++) {
        h_colptr[p] = r->h_colptr[p]; h_colchunk[p] = k;
        for (i64 b = r->h_colptr[p]; b < r->h_colptr[p + 1]; b += RED_CHUNK) {
          h_cbegin[k] = b; h_clen[k] = (u32)std::min(RED_CHUNK, r->h_colptr[p + 1] - b); k++;
        }
        h_sel[p] = cols[p]; h_scale[p] = scale ? scale[p] : 1.0; h_off[p] = offset ? offset[p] : 0.0;
        h_srclen[p] = full->h_collen[cols[p]]; h_dstlen[p] = r->h_collen[p];
      }
      h_colptr[c1] = r->h_colptr[c1]; h_colchunk[c1] = k;
    }
    hipStream_t st = ctx->stream;
    hipError_t e = hipMemcpyAsync(dtab, htab, tab_bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipEventRecord(ctx->stage_ev, st);
    if (e == hipSuccess && n) e = hipMemcpyAsync(r->labels, full->labels, (size_t)n, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && n && full->active) {
      r->active = r->active_slot;
      e = hipMemcpyAsync(r->active, full->active, (size_t)n, hipMemcpyDeviceToDevice, st);
    }
    if (e != hipSuccess) { rc = klr_fail(ctx, KLR_EHIP, "select_columns copy failed: %s", hipGetErrorString(e)); break; }
    const double* scp = scale ? d_scale : nullptr;
    if (dense) {
      const i64 tot = std::max<i64>(n * c1, n + 1);
      hipLaunchKernelGGL(k_dense_init, dim3((unsigned)ceil_div(tot, 256)), dim3(256), 0, st, n, (int)c1, colstride, scp,
                         d_off, r->rowptr, r->col, (double*)r->val, r->crow, (double*)r->cval);
      // padding rows of the dense CSC columns
      if (colstride != n) {
        // entries [n, colstride) of every column: row 0 / value 0
        for (i64 p = 0; p < c1; p++) {
          (void)hipMemsetAsync(r->crow + p * colstride + n, 0, (size_t)(colstride - n) * 4, st);
          (void)hipMemsetAsync((double*)r->cval + p * colstride + n, 0, (size_t)(colstride - n) * 8, st);
        }
      }
      if (full->val_f64)
        hipLaunchKernelGGL(k_dense_overwrite<double>, dim3((unsigned)c1), dim3(256), 0, st, full->colptr, d_srclen, full->crow,
                           (const double*)full->cval, d_sel, (int)c1, colstride, scp, d_off, (double*)r->val, (double*)r->cval);
      else
        hipLaunchKernelGGL(k_dense_overwrite<float>, dim3((unsigned)c1), dim3(256), 0, st, full->colptr, d_srclen, full->crow,
                           (const float*)full->cval, d_sel, (int)c1, colstride, scp, d_off, (double*)r->val, (double*)r->cval);
    } else {
      // row marks: bit masks (sorted rows for free) when they are small, else counters + cursor + in-row sort
      const int W = (int)ceil_div(c1, (i64)64);
      const char* ss_env = ctx_opt(ctx, "KLR_SUBSELECT");         // bin | flat | sort (tests); default: the first that applies
      const bool use_bin = !any_zero && n > 0 && c1 <= RB_CMAX && n < ((i64)1 << 32) && (!ss_env || ss_env[0] == 'b');
      if (use_bin) {
        const i64 nblk = ceil_div(n, (i64)RB_ROWS);
        char* wsb;
        const size_t m_bytes = al(nn * (size_t)W * 8), t_bytes = al((size_t)nblk * 4), s_bytes = al((size_t)nblk * c1 * 8);
        if ((rc = ws_get(ctx, WS_TMP1, m_bytes + t_bytes + s_bytes, (void**)&wsb))) break;
        unsigned long long* gmask = (unsigned long long*)wsb;
        u32* blocktot = (u32*)(wsb + m_bytes);
        u32* seg = (u32*)(wsb + m_bytes + t_bytes);
        const size_t ldsb = (size_t)RB_ROWS * W * 8 + (size_t)RB_TCAP * 12 + (size_t)(RB_ROWS + 1) * 4;
        if ((rc = ctx_func_lds(ctx, (const void*)k_red_bin_b, (int)((size_t)RB_ROWS * RB_WMAX * 8 + (size_t)RB_TCAP * 12 + (size_t)(RB_ROWS + 1) * 4)))) break;
        if (full->val_f64)
          hipLaunchKernelGGL(k_red_bin_a<double>, dim3((unsigned)nblk), dim3(1024), 0, st, full->colptr, full->crow, (const double*)full->cval,
                             d_sel, r->colptr, d_dstlen, (int)c1, scp, n, W, r->crow, (double*)r->cval, gmask, blocktot, seg);
        else
          hipLaunchKernelGGL(k_red_bin_a<float>, dim3((unsigned)nblk), dim3(1024), 0, st, full->colptr, full->crow, (const float*)full->cval,
                             d_sel, r->colptr, d_dstlen, (int)c1, scp, n, W, r->crow, (double*)r->cval, gmask, blocktot, seg);
        hipLaunchKernelGGL(k_red_bin_b, dim3((unsigned)nblk), dim3(1024), ldsb, st, r->colptr, (int)c1, n, W, r->crow,
                           (const double*)r->cval, (const unsigned long long*)gmask, (const u32*)blocktot, (const u32*)seg, r->rowptr,
                           r->col, (double*)r->val);
        e = hipGetLastError();
        if (e != hipSuccess) { rc = klr_fail(ctx, KLR_EHIP, "select_columns launch failed: %s", hipGetErrorString(e)); break; }
        break;
      }
      const bool use_mask = W <= RED_MASK_WORDS_MAX && !(ss_env && ss_env[0] == 's');
      u32* cnt; unsigned long long* mask = nullptr;
      const size_t cnt_bytes = al(nn * sizeof(u32));
      if ((rc = ws_get(ctx, WS_TMP1, cnt_bytes + (use_mask ? nn * (size_t)W * 8 : 0), (void**)&cnt))) break;
      if (use_mask) { mask = (unsigned long long*)((char*)cnt + cnt_bytes); (void)hipMemsetAsync(mask, 0, nn * (size_t)W * 8, st); }
      else (void)hipMemsetAsync(cnt, 0, nn * sizeof(u32), st);
      const unsigned eb = (unsigned)std::max<i64>(1, ceil_div(r->nnz_pad, (i64)256));
      const size_t lds = c1 + 1 <= RED_LDS_COLS ? (size_t)(c1 + 1) * 8 : 0;
      if (any_zero) {
        // explicit zeros in a selected column are dropped (:332): compact per column, then mark
        if (full->val_f64)
          hipLaunchKernelGGL(k_gather_cols_nz<double>, dim3((unsigned)c1), dim3(64), 0, st, full->colptr, d_srclen, full->crow,
                             (const double*)full->cval, d_sel, r->colptr, scp, r->crow, (double*)r->cval);
        else
          hipLaunchKernelGGL(k_gather_cols_nz<float>, dim3((unsigned)c1), dim3(64), 0, st, full->colptr, d_srclen, full->crow,
                             (const float*)full->cval, d_sel, r->colptr, scp, r->crow, (double*)r->cval);
        hipLaunchKernelGGL((k_red_gather<float, false>), dim3(eb), dim3(256), lds, st, full->colptr, full->crow, (const float*)nullptr,
                           d_sel, r->colptr, d_dstlen, (int)c1, scp, r->crow, (double*)r->cval, mask, W, cnt);
      } else if (full->val_f64) {
        hipLaunchKernelGGL((k_red_gather<double, true>), dim3(eb), dim3(256), lds, st, full->colptr, full->crow, (const double*)full->cval,
                           d_sel, r->colptr, d_dstlen, (int)c1, scp, r->crow, (double*)r->cval, mask, W, cnt);
      } else {
        hipLaunchKernelGGL((k_red_gather<float, true>), dim3(eb), dim3(256), lds, st, full->colptr, full->crow, (const float*)full->cval,
                           d_sel, r->colptr, d_dstlen, (int)c1, scp, r->crow, (double*)r->cval, mask, W, cnt);
      }
      if (use_mask && n) hipLaunchKernelGGL(k_red_count, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, mask, W, n, cnt);
      if ((rc = launch_exscan_u32(ctx, cnt, n, r->rowptr))) break;
      if (!use_mask) (void)hipMemsetAsync(cnt, 0, nn * sizeof(u32), st);
      hipLaunchKernelGGL(k_red_fill, dim3(eb), dim3(256), lds, st, r->colptr, d_dstlen, (int)c1, r->crow, (const double*)r->cval,
                         r->rowptr, mask, W, cnt, r->col, (double*)r->val);
      if (!use_mask && n)
        hipLaunchKernelGGL(k_sort_rows, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, r->rowptr, n, r->col, (double*)r->val);
    }
    e = hipGetLastError();
    if (e != hipSuccess) { rc = klr_fail(ctx, KLR_EHIP, "select_columns launch failed: %s", hipGetErrorString(e)); break; }
    // no synchronisation: the tables are part of the matrix's own block, everything downstream runs on ctx->stream
  } while (0);
  if (rc != KLR_OK) { std::string keep = ctx->err; mat_free(r); ctx->err = keep; return rc; }
  *out = r;
  return KLR_OK;
}

extern "C" int klr_mat_select_columns(klr_ctx* ctx, const klr_mat* full, const int64_t* cols, int64_t c_plus_1,
                                      const double* scale, const double* offset, klr_mat** reduced) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, full != nullptr && cols != nullptr && reduced != nullptr);
  return build_reduced(ctx, full, cols, c_plus_1, scale, offset, true, reduced);
}
