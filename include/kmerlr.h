/*
 * kmerlr.h -- C ABI of libkmerlr_hip.so: the MI355X (gfx950) implementation of kmerLr's
 * L1 logistic-regression hot path.  Plain pointers and sizes only; no torch/HIP types.
 *
 * The reference (pbenner/kmerLr @ v1, Go) has no FFI; the path sits behind Go method calls.
 * Each entry point below names the reference call site it replaces (file:line relative to
 * the reference tree).  INTEGRATION.md shows the cgo stub a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success and a nonzero KLR_E* code otherwise;
 *     klr_last_error(ctx) gives the message (ctx may be NULL for creation failures).
 *   - a klr_ctx is bound to one GPU and one HIP stream and is single-threaded; use one ctx
 *     per concurrent estimator (the reference runs one estimator per ensemble member / CV fold:
 *     kmerLr_estimator_ensemble.go:81, kmerLr_crossvalidation.go:176).
 *   - host buffers stay caller-owned; matrices are copied to HBM at creation
 *     (the reference's solver keeps references to caller rows instead, kmerLr_estimator.go:152).
 *   - rows use the reference layout: per row ascending column ids, col[rowptr[i]] == 0 is the
 *     bias entry with value 1.0 (kmerLr_data.go:199-200, scoresLr_data.go:108-117).
 *   - there is NO CPU fallback: without a HIP device klr_ctx_create fails.
 *
 * Sharding: a matrix holds rows [row0, row0+n) of a data set with n_total rows.  Every
 * quantity that is a sum over samples (gradient, loss, Gram) is computed as a local partial,
 * passed through the ctx's all-reduce hook (klr_ctx_set_allreduce / klr_ctx_init_rccl) if one
 * is installed, and finished identically on every rank.
 */
#ifndef KMERLR_H
#define KMERLR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KLR_OK          0
#define KLR_EINVAL      1   /* bad argument (the reference panics "internal error")      */
#define KLR_EHIP        2   /* HIP runtime error                                          */
#define KLR_ENODEV      3   /* no usable gfx950 device                                    */
#define KLR_ENOMEM      4
#define KLR_ECOMM       5   /* all-reduce hook / RCCL failure                             */

typedef struct klr_ctx klr_ctx;
typedef struct klr_mat klr_mat;

/* ---- context ------------------------------------------------------------------------- */
int         klr_ctx_create(int device_id, klr_ctx** out);
/* A run of consecutive reduced solves (klr_cd_solve) begins: the first batch of sweeps of a re-linearisation is sized by
 * what the same re-linearisation took in the solve before -- a matter of speed, the iterates do not depend on it; the call
 * forgets that history (klr_estimator_estimate makes it through the backend table). */
void        klr_solve_run_begin(klr_ctx* ctx);
void        klr_solve_run_end(klr_ctx* ctx);          /* later solves start without a history again */
/* A second context on the same device with the switches of `parent` (own stream, own workspaces), for concurrent work on
 * shared data from another host thread: one fold of a cross-validation per context, on klr_mat_view's of the data
 * (klr_config.threads_cv; the reference runs its folds in threads too, kmerLr_crossvalidation.go:176).  A context with
 * a communicator is not forked.  klr_ctx_release hands it back to the parent, which keeps it (warm workspaces) for the
 * next fork and destroys it when it is destroyed itself; klr_ctx_destroy ends it at once. */
int         klr_ctx_fork(klr_ctx* parent, klr_ctx** out);
void        klr_ctx_release(klr_ctx* forked);
void        klr_ctx_destroy(klr_ctx* ctx);
const char* klr_last_error(const klr_ctx* ctx);
/* run the ctx's kernels on an existing HIP stream (hipStream_t passed as void*); NULL = the
 * ctx's own stream.  Lets a host (e.g. torch) order its own work against the kernels. */
int         klr_ctx_set_stream(klr_ctx* ctx, void* hip_stream);
int         klr_ctx_synchronize(klr_ctx* ctx);

/* all-reduce hook: sums `count` elements of type dtype (KLR_F64 doubles or KLR_I64 int64) in place
 * across ranks.  `buf` is a DEVICE pointer; the call is made after the producing kernels were
 * enqueued on the ctx stream (also passed, as void*), and the hook must leave the result visible to
 * later work on that stream.  Count data is reduced as int64 fixed point, so that sum is exact and
 * independent of the number of ranks. */
#define KLR_F64 0
#define KLR_I64 1
typedef int (*klr_allreduce_fn)(void* user, void* dev_buf, int64_t count, int dtype, void* hip_stream);
int         klr_ctx_set_allreduce(klr_ctx* ctx, klr_allreduce_fn fn, void* user, int rank, int nranks);
/* Optional second pair of hooks for the full-width sweep's collective ("a direct reduce-scatter +
 * all-gather that uses all links ... prefer the latter").  With both installed (and klr_ctx_set_allreduce's nranks > 1)
 * klr_sweep_select reduce-scatters the int64 column sums -- rank r then owns the exact sums of columns
 * [r * slice, (r+1) * slice), slice = ceil((dim + 5) / nranks) -- selects its slice's 2N largest |g| locally and
 * all-gathers the nranks x 2N candidates (a few KB): the same selection and lambda as the all-reduce path, bit for bit,
 * for half the bytes on the links and 1/nranks of the conversion work.  Semantics:
 *   reduce_scatter: recv[i] = sum over ranks of send[rank * recv_count + i], i < recv_count (send holds nranks * recv_count
 *                   elements of dtype); out of place
 *   allgather:      recv[r * send_bytes + b] = rank r's send[b]
 * Both take DEVICE pointers and run in stream order on the ctx stream.  klr_ctx_init_rccl installs RCCL's. */
typedef int (*klr_reduce_scatter_fn)(void* user, const void* send_dev, void* recv_dev, int64_t recv_count, int dtype, void* hip_stream);
typedef int (*klr_allgather_fn)(void* user, const void* send_dev, void* recv_dev, int64_t send_bytes, void* hip_stream);
int         klr_ctx_set_collectives(klr_ctx* ctx, klr_reduce_scatter_fn rs, klr_allgather_fn ag, void* user);
/* built-in hook: RCCL over xGMI.  id = 128-byte ncclUniqueId obtained on rank 0 with
 * klr_rccl_unique_id and distributed by the launcher. */
int         klr_rccl_unique_id(void* id128);
int         klr_ctx_init_rccl(klr_ctx* ctx, const void* id128, int rank, int nranks);

/* built-in hooks for ranks that are processes of ONE host without a device interconnect between them (several ranks
 * on one GPU in tests; `kmerlr --ranks=N --transport=shm`): host-staged through a shared memory region of
 * klr_shm_region_bytes(nranks) bytes that the launcher maps MAP_SHARED (before fork, or by name) and initialises once
 * with klr_shm_region_init before any rank attaches.  Installs all-reduce, reduce-scatter and all-gather and runs the
 * collective self-check.  Sums are taken in rank order by every rank: all ranks hold bitwise the same result. */
int64_t     klr_shm_region_bytes(int nranks);
int         klr_shm_region_init(void* region, int64_t bytes, int nranks);
void        klr_shm_region_abort(void* region);      /* releases ranks waiting at a barrier of the region (a rank died) */
int         klr_ctx_init_shm(klr_ctx* ctx, void* region, int64_t bytes, int rank, int nranks);
/* sums a HOST buffer over the ranks through the installed hook (no-op without one): how a launcher collects per-row
 * results of the shards (each rank fills its own rows, zeros elsewhere).  dtype KLR_F64 | KLR_I64. */
int         klr_allreduce_host(klr_ctx* ctx, void* host_buf, int64_t count, int dtype);

/* what the ctx will reduce over: rank / nranks as given to the setters above; rccl_ranks = ncclCommCount of the built-in
 * communicator (0 when there is none).  Any output may be NULL. */
int         klr_ctx_comm_info(const klr_ctx* ctx, int* rank, int* nranks, int* rccl_ranks);

/* Collective self-check of whatever hooks are installed (klr_ctx_init_rccl runs it by itself and fails if it fails):
 * all-reduce of [1, rank, ...] as int64 and f64 must return [N, N(N-1)/2, ...], reduce-scatter and all-gather their
 * documented slices.  Every rank must call it.  The reduce-scatter form of the sweep's collective is used only on a
 * context that passed. */
int         klr_ctx_comm_selfcheck(klr_ctx* ctx);
/* full-width sweeps by the collective they posted, and how many reduce-scatter sweeps had to repeat as an all-reduce
 * (a rank's candidate list could not prove lambda: klr_ctx_set_collectives); checked = the self-check passed */
int         klr_ctx_comm_stats(const klr_ctx* ctx, int64_t* sweeps_allreduce, int64_t* sweeps_scatter,
                               int64_t* scatter_fallbacks, int* checked);
/* how many of the reduce-scatter sweeps ran as one launch per strip range with the collective of a range behind the next
 * range's launch (the overlapped form) */
int         klr_ctx_comm_ranged(const klr_ctx* ctx, int64_t* sweeps_ranged);

/* what the context measured before the first wide sweep it could reduce-scatter (averaged over the ranks): the installed
 * reduce-scatter on a vector of `elements` int64, the all-gather of the candidate blocks, ONE all-reduce of the same
 * vector; scatter_form = 1 when reduce-scatter + all-gather (+ 20 us of slice work) beat the all-reduce and that form is
 * used (-1: nothing measured yet); ranges = strip ranges of the overlapped sweep (1 = one launch).  Any output may be
 * NULL.  KLR_RS / KLR_SWEEP_RANGES override the choices. */
int         klr_ctx_comm_probe(const klr_ctx* ctx, int64_t* elements, double* reduce_scatter_ms, double* allgather_ms,
                               double* allreduce_ms, int* scatter_form, int* ranges);

/* KLR_* switches.  A context reads the process environment ONCE, in klr_ctx_create; later changes
 * go through these calls (value NULL unsets).  klr_ctx_reload_env re-reads the environment. */
int         klr_ctx_set_option(klr_ctx* ctx, const char* name, const char* value);
int         klr_ctx_reload_env(klr_ctx* ctx);

/* ---- sparse k-mer / score matrix ------------------------------------------------------- */
/* Replaces the []ConstVector the reference hands to SetSparseData (kmerLr_estimator.go:152)
 * and to featureSelector.gradient (kmerLr_feature_selection.go:221-229).
 * n rows of dimension dim = m+1; n_total = rows of the whole data set (0: = n). */
int  klr_mat_from_csr(klr_ctx* ctx, int64_t n, int64_t dim, const int64_t* rowptr,
                      const int64_t* col, const double* val, const uint8_t* labels,
                      int64_t n_total, klr_mat** out);
/* The same matrix from FASTA sequences, counted ON THE DEVICE (scan_sequences / convert_counts,
 * kmerLr_data.go:197-335): seqs = concatenated sequence bytes, seq_off[n+1]; k-mer lengths M..N (<= 12);
 * revcomp merges a k-mer with its reverse complement; binarize stores 1 for present classes.  Column ids: for
 * k = M..N in turn, classes in increasing base-4 code order (a=0,c=1,g=2,t=3; case folded; windows holding any
 * other character are skipped).  CSR, CSC and the tiled stream are all built on the device. */
int  klr_mat_from_sequences(klr_ctx* ctx, const char* seqs, const int64_t* seq_off, int64_t n, int M, int N,
                            int revcomp, int binarize, const uint8_t* labels, int64_t n_total, klr_mat** out);
/* k-mer equivalence classes: the counterpart of gonetics' KmerEquivalenceRelation / KmerCounter as the reference
 * builds them -- NewKmerCounter(M, N, complement, reverse, revcomp, maxAmbiguous, alphabet), kmerLr_learn.go:94,
 * kmerLr_features.go:204 (the module itself is not in the reference tree; semantics pinned by TestKmers1,
 * kmerLr_test.go:30-66).  A class is the orbit of a k-mer under the enabled operations; with the gapped alphabet a
 * k-mer may hold the gap letter 'n' (matches any nucleotide) at inner positions, at most max_ambiguous of them
 * (-1: no bound; one value, or one per length M..N).  Columns: lengths ascending, inside a length by the code of the
 * canonical member (a < c < g < t < n); class c is column c + 1 of the matrix.  Names are the members joined by '|'
 * ("kmer|revcomp").  klr_kmer_class_find takes any member as text (upper or lower case; anything after a '|' is
 * ignored, like KmerLrFeatures.ImportConfig, kmerLr_features.go:207-209) and returns the 0-based class or -1. */
#define KLR_ALPHABET_NUCLEOTIDE 0
#define KLR_ALPHABET_GAPPED     1    /* "gapped-nucleotide" */
#define KLR_ALPHABET_IUPAC      2    /* "iupac-nucleotide": a c g t r y s w k m b d h v n, every code matching its nucleotide set; an
                                      * ambiguous code counts towards max_ambiguous and, like the gap letter, never stands first or
                                      * last; k-mers up to 6 letters.  No fixture of the reference covers this alphabet: parity unpinned. */
#define KLR_KMER_MAX_LEN        12
#define KLR_KMER_INVALID        0xFFFFFFFFu
typedef struct {
  int M, N;
  int complement, reverse, revcomp;
  int alphabet;
  int n_max_ambiguous;          /* 0 (no bound), 1, or N - M + 1 */
  const int* max_ambiguous;
} klr_kmer_spec;
typedef struct klr_kmer_classes klr_kmer_classes;
int     klr_kmer_classes_create(const klr_kmer_spec* spec, klr_kmer_classes** out, char* errbuf, int64_t errcap);
void    klr_kmer_classes_destroy(klr_kmer_classes* kc);
int64_t klr_kmer_classes_count(const klr_kmer_classes* kc);
int     klr_kmer_class_name(const klr_kmer_classes* kc, int64_t cls, char* buf, int64_t cap);
int64_t klr_kmer_class_find(const klr_kmer_classes* kc, const char* kmer);
/* klr_mat_from_sequences for any class list (complement / reverse / revcomp, gapped alphabet, max-ambiguous):
 * every window of length k = M..N is counted once in the class of every k-mer of the alphabet it matches. */
int  klr_mat_from_sequences_classes(klr_ctx* ctx, const char* seqs, const int64_t* seq_off, int64_t n,
                                    const klr_kmer_classes* classes, int binarize, const uint8_t* labels,
                                    int64_t n_total, klr_mat** out);
void klr_mat_destroy(klr_mat* mat);
int  klr_mat_shape(const klr_mat* mat, int64_t* n, int64_t* dim, int64_t* nnz, int64_t* n_total);
/* how the matrix sits in HBM for the full-width sweep: tiled = 1 when the packed integer stream (count data; 2, 3 or
 * 4 bytes per stored unit entry) is in use, with its tile height, number of (column strip,
 * workgroup) pieces, stored entries incl. padding and value_bytes = bytes per padded entry - 3; tiled = 0: plain CSC
 * (u32 row + f32|f64 value) with its chunk count, lanes per chunk and value_bytes = 4|8. */
int  klr_mat_layout(const klr_mat* mat, int* tiled, int64_t* tile_rows, int64_t* nseg,
                    int64_t* padded_entries, int* lanes_per_segment, int* value_bytes);
/* copy a (reduced) matrix back in the reference layout; any output pointer may be NULL.
 * col/val need klr_mat_shape's nnz entries. */
int  klr_mat_to_csr(klr_ctx* ctx, const klr_mat* mat, int64_t* rowptr, int64_t* col, double* val);

/* Co-occurrence features (CoeffIndex, kmerLr_coefficients_index.go:26-54; LinearPdf / Gradient with
 * Cooccurrence, kmerLr_logistic_regression.go:69-85,183-195; featureSelection.Data :325-330).  For a base matrix
 * with nb features the result has klr_coeff_dim(nb) = nb(nb+1)/2 + 1 columns: bias, the nb singles at k+1, the
 * pairs k1 < k2 at klr_coeff_ind2sub(nb,k1,k2), value v_k1 * v_k2 for the stored entries of the row.  Built on
 * the device; labels, n_total and the row mask are inherited.  Base matrices wider than 65,535 features must be
 * pre-selected first (--co-preselection, kmerLr_estimator_ensemble.go:99-115). */
int     klr_mat_expand_cooccurrence(klr_ctx* ctx, const klr_mat* base, klr_mat** out);
int64_t klr_coeff_dim(int64_t nb);
int64_t klr_coeff_ind2sub(int64_t nb, int64_t k1, int64_t k2);
void    klr_coeff_sub2ind(int64_t nb, int64_t i, int64_t* k1, int64_t* k2);

/* Restrict every sum over samples (gradient, loss, Gram, class counts) to the rows with mask[i] != 0 --
 * the training / test split of a cross-validation fold or ensemble member (filterCvGroup,
 * kmerLr_crossvalidation.go:96-120) without re-uploading the matrix.  n_active_total = number of active
 * rows over all shards (takes the place of len(data) in 1/n and lambda = L1Reg/n).  mask = NULL restores
 * all rows.  Reduced matrices inherit the mask in force when they are created.  klr_log_pdf ignores it. */
int  klr_mat_set_active_rows(klr_ctx* ctx, klr_mat* mat, const uint8_t* mask, int64_t n_active_total);
/* A matrix for another context of the same device that SHARES the device storage of `mat` (which must outlive it) and has
 * its own row mask: what a concurrent fold works on.  Destroyed with klr_mat_destroy (frees the mask only). */
int         klr_mat_view(klr_ctx* ctx, const klr_mat* mat, klr_mat** out);

/* compute_class_weights (kmerLr_data.go:178-193) over the (all-reduced) label counts */
int  klr_class_weights(klr_ctx* ctx, const klr_mat* mat, double cw[2]);

/* ---- L1: logistic-regression arithmetic ------------------------------------------------ */
/* logisticRegression.Gradient with Lambda = 0 as called by featureSelector.gradient
 * (kmerLr_logistic_regression.go:151-248; kmerLr_feature_selection.go:221-229).
 * theta is given sparse: th_idx ascending column ids (may include 0), th_val, th_nnz.
 * scale/offset: TransformFull vectors of length dim or NULL (kmerLr_transform.go:260-268).
 * g_out: host, length dim. */
int  klr_gradient(klr_ctx* ctx, const klr_mat* mat, const int64_t* th_idx, const double* th_val,
                  int64_t th_nnz, const double cw[2], const double* scale, const double* offset,
                  double* g_out);
/* logisticRegression.Loss (kmerLr_logistic_regression.go:250-272); theta dense, length dim;
 * lambda NaN or 0 = no L1 term. */
int  klr_loss(klr_ctx* ctx, const klr_mat* mat, const double* theta, const double cw[2],
              double lambda, double* loss_out);
/* LinearPdf / LogPdf per row (kmerLr_logistic_regression.go:47-149; predict path
 * kmerLr_classifier.go): out[i] = log sigma(x_i . theta) for the local rows. */
int  klr_log_pdf(klr_ctx* ctx, const klr_mat* mat, const double* theta, double* out);

/* TransformFull.Fit (kmerLr_transform.go:59-252): column statistics over the resident matrix (all shards;
 * honours the active-row mask) -> Scale / Offset of length dim.
 * kind: 1 standardizer, 2 variance-scaler, 3 max-abs-scaler, 4 mean-scaler (KLR_TR_*); *has_offset = 1 when
 * the transform keeps an Offset (standardizer).  max-abs needs a max over ranks and is refused with an
 * all-reduce hook installed (the hook only sums). */
#define KLR_TR_STANDARDIZER  1
#define KLR_TR_VARIANCE      2
#define KLR_TR_MAXABS        3
#define KLR_TR_MEAN          4
int  klr_transform_fit(klr_ctx* ctx, const klr_mat* mat, int kind, double* scale_out, double* offset_out,
                       int* has_offset);

/* ---- L3: feature selection -------------------------------------------------------------- */
/* featureSelector.Select + computeLambda + NLargestAbsFloat64
 * (kmerLr_feature_selection.go:78-134,179-192; kmerLr_sort.go:120-131).
 *   g        gradient[1:], host, length m
 *   b        in/out mask, length m+1 (b[0] = 1 bias; b[j+1] = 1 where theta_j != 0)
 *   c_prev   nonzero non-bias coefficients restored; N = number of features to select
 *   lambda_prev, eps_lambda: the `ok` rule of :133
 * Tie rule (the reference's unstable sort leaves it open): larger |g| first, then LOWER
 * column index. */
int  klr_select(klr_ctx* ctx, const double* g, int64_t m, uint8_t* b, int64_t c_prev, int64_t N,
                double lambda_prev, double eps_lambda,
                double* lambda_out, int64_t* c_out, int* ok_out);
/* featureSelector.SelectFixed (kmerLr_feature_selection.go:136-159); g_full has length m+1 */
int  klr_select_fixed(klr_ctx* ctx, const double* g_full, int64_t m, uint8_t* b, int64_t c_prev,
                      double lambda, int64_t max_features, int64_t* c_out, int* ok_out);
/* featureSelection.Data followed by Transform.Apply (kmerLr_feature_selection.go:309-343,
 * kmerLr_transform.go:584-629): cols = ascending selected column ids incl. 0; scale/offset are
 * the SELECTED transform vectors (length c_plus_1) or NULL. */
int  klr_mat_select_columns(klr_ctx* ctx, const klr_mat* full, const int64_t* cols, int64_t c_plus_1,
                            const double* scale, const double* offset, klr_mat** reduced);

/* ---- fused leapfrog epoch step (device resident) ----------------------------------------- */
/* One full-width sweep as the leapfrog driver needs it (kmerLr_estimator.go:235):
 * X.theta / sigmoid / loss from the REDUCED matrix of the previous epoch (theta lives on its
 * columns), per-feature gradient over the FULL matrix, one all-reduce of [g | loss], then
 * top-2N selection and lambda on the device.  reduced may be NULL (theta = theta0 on the bias
 * only, first epoch).  red_cols[c_plus_1] maps reduced columns to full columns.  With a data transform
 * the reduced matrix already holds transformed values (klr_mat_select_columns) and scale/offset finish
 * the full-width gradient on the fly (kmerLr_logistic_regression.go:196-222).
 * Outputs as klr_select; loss_out = the mean log-loss without penalty; g_out (nullable, host, length dim) = the
 * gradient the selection saw, i.e. with scale/offset applied like klr_gradient's. */
int  klr_sweep_select(klr_ctx* ctx, const klr_mat* full, const klr_mat* reduced,
                      const int64_t* red_cols, const double* theta_red, int64_t c_plus_1,
                      const double cw[2], int64_t N, double lambda_prev, double eps_lambda,
                      const double* scale /* nullable, length dim: TransformFull.Scale  */,
                      const double* offset /* nullable, length dim: TransformFull.Offset */,
                      uint8_t* b, double* lambda_out, int64_t* c_out, int* ok_out,
                      double* loss_out, double* g_out /* nullable, length dim */);

/* ---- L2: solver -------------------------------------------------------------------------- */
/* hook record, kmerLr_estimator_hook.go:46-99 / Trace.Append (kmerLr_estimator_trace.go:80) */
typedef struct {
  int32_t iteration;
  int32_t nonzero;    /* nonzero coefficients incl. bias */
  double  change;     /* eval_stopping delta */
  double  lambda;     /* l1reg / n_total */
  double  loss;       /* NaN when eval_loss == 0 */
  double  seconds;
} klr_iter_rec;

/* IRLS + covariance-update cyclic coordinate descent with soft threshold: the algorithm of
 * estimate_coordinate / estimate_coordinate_loop (kmerLr_estimator_coordinate.go:31-139),
 * eval_stopping (kmerLr_estimator_proximal.go:30-52) and the hook's loss rule
 * (kmerLr_estimator_hook.go:94-98).  Takes the place of SetSparseData/Estimate/GetEstimate
 * (kmerLr_estimator.go:152-164).  theta: in warm start, out estimate, length dim(reduced).
 * l1reg = lambda * n_total.  recs may be NULL. */
int  klr_cd_solve(klr_ctx* ctx, const klr_mat* reduced, double* theta, const double cw[2],
                  double l1reg, double l2reg, double eps, double eps_loss, int eval_loss,
                  int64_t max_iter, klr_iter_rec* recs, int64_t recs_cap, int64_t* n_recs);

/* ---- synthetic workload (bench / tests) ----------------------------------- */
/* rows [row0,row0+n_local) of the deterministic synthetic k-mer count matrix, reference layout.
 * col/val need n_local*(draws+1) entries. */
int  klr_synth_csr(uint64_t seed, int64_t row0, int64_t n_local, int64_t n_total, int64_t m,
                   int64_t draws, int64_t S, int64_t* rowptr, int64_t* col, double* val,
                   uint8_t* labels, int threads);

/* ---- measurement -------------------------------------------------------------------------- */
/* HIP-event timing of the dominant kernel (the full-width column sweep) on the ctx stream:
 * accumulated since the last reset.  Events are recorded only while enabled. */
/* on = 0 off; 1 every launch of the sweep kernel is timed; k > 1: every k-th one (the events ride on the dispatch packet, but
 * each timed launch still costs a completion signal: a sampled timed region stays close to an untimed one) */
int  klr_prof_enable(klr_ctx* ctx, int on);
/* creates timing events ahead of a timed region (they are pooled and reused: klr_prof_read with reset hands them back) */
int  klr_prof_reserve(klr_ctx* ctx, int64_t events);
int  klr_prof_read(klr_ctx* ctx, int64_t* launches, double* total_ms, int reset);
/* while enabled, klr_cd_solve also records HIP events around the coordinate-descent sweep kernels of every batch
 * (soft-threshold sweeps only: no IRLS set-up, loss pass or host turnaround): sweeps executed, batches, their time. */
int  klr_prof_read_cd(klr_ctx* ctx, int64_t* sweeps, int64_t* batches, double* total_ms, int reset);
/* writes and copies `bytes` of scratch on the ctx stream so that whatever sat in the 256 MiB Infinity Cache is gone:
 * the next sweep reads its stream from HBM (bench.py's cold-stream figure). */
int  klr_prof_evict(klr_ctx* ctx, int64_t bytes);
/* the packed stream of a tiled matrix: format (0 u32, 1 u24, 2 r64, 3 q64/12, 4 q64/13), bytes of the stream, bytes of
 * one piece vector the sweep writes, bytes of the side tables the sweep reads.  Zeros for a plain matrix. */
int  klr_mat_stream_info(const klr_mat* mat, int* format, int64_t* stream_bytes, int64_t* piece_bytes, int64_t* table_bytes);

/* average milliseconds of the installed all-reduce hook alone on `count` elements of dtype (scratch buffer, ctx stream,
 * HIP events around `iters` calls); 0 without a hook.  Collective: every rank must call it. */
int  klr_allreduce_time(klr_ctx* ctx, int64_t count, int dtype, int iters, double* ms_avg);

const char* klr_version(void);

#ifdef __cplusplus
}
#endif
#endif
