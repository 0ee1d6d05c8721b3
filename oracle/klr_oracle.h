/*
 * klr_oracle.h -- CPU restatement of the kmerLr L1 logistic-regression hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under kmerlr_amd/ (the product) may include,
 * link or load this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / the timed CPU baseline.
 *
 * Parity pinning: the reference (pbenner/kmerLr @ v1, Go) cannot be built in this
 * image (no Go toolchain, four un-vendored modules), so this restatement is pinned
 * by the reference's own known answers (tests/test_oracle_golden.py):
 *   G1 gradient at theta=0 on scoresLr_test_{fg,bg}.table
 *   G2 first leapfrog lambda 2.496875e+00 (README.md:39) and picked columns {1,6}
 *   G3 loss 0.813659729805629 (scoresLr_test.go:59)
 *   G4 selected Index {1,6} (scoresLr_test.go:53-58)
 *   G7 LogPdf -0.6444834689451768 / -0.744447612033651, loss 0.644417014007959
 *      (kmerLr_test.go:201-257)
 * The solver trajectory is NOT pinned by the reference: its shipped solver is SAGA
 * from github.com/pbenner/autodiff v1.0.0 (absent here); klo_cd_solve restates the
 * intended algorithm of the unreferenced kmerLr_estimator_coordinate.go and is
 * checked against the KKT conditions and against the reference's theta goldens at
 * SAGA's early-stop accuracy (1e-2 relative).  "parity unpinned" for that part.
 *
 * All row data uses the reference layout: per-row ascending int64 indices + f64
 * values (16 B/nnz), index 0 = bias (kmerLr_data.go:199-200).
 */
#ifndef KLR_ORACLE_H
#define KLR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int64_t        n;       /* rows (samples)                                   */
  int64_t        dim;     /* columns incl. bias = m+1                         */
  const int64_t* rowptr;  /* n+1                                              */
  const int64_t* col;     /* ascending per row, col[rowptr[i]] == 0           */
  const double*  val;
} klo_csr;

/* mirrors klr_iter_rec of include/kmerlr.h (hook record, kmerLr_estimator_hook.go:46-99) */
typedef struct {
  int32_t iteration;
  int32_t nonzero;   /* nonzero coefficients INCLUDING the bias (hook.go:69-73) */
  double  change;    /* delta of eval_stopping                                 */
  double  lambda;    /* l1reg / n                                              */
  double  loss;      /* NaN when eval_loss == 0                                */
  double  seconds;
} klo_iter_rec;

#define KLO_IRLS_WMIN 1e-12   /* guard for the unguarded 1/w of coordinate.go:100-105 */

/* row sharding for the multi-rank driver tests: rows here are a shard of n_total rows; sums over
 * samples pass through fn (in place, all ranks) before use.  n_total = 0 / fn = NULL: off. */
typedef void (*klo_allreduce_fn)(double* buf, int64_t count, void* user);
void   klo_set_sharding(int64_t n_total, klo_allreduce_fn fn, void* user);

double klo_log_add(double a, double b);
double klo_linear_pdf(const klo_csr* x, int64_t i, const double* theta,
                      const double* scale, const double* offset);
double klo_class_log_pdf(const klo_csr* x, int64_t i, const double* theta,
                         const double* scale, const double* offset, int y);
void   klo_class_weights(const uint8_t* labels, int64_t n, double cw[2]);

/* bench.py's all-core figure only: the Gradient loop with rows split over OpenMP threads (no transform) */
int    klo_gradient_mt(const klo_csr* x, const uint8_t* labels, const double* theta,
                       const double cw[2], double* g, int threads);
int    klo_gradient(const klo_csr* x, const uint8_t* labels, const double* theta,
                    const double cw[2], double lambda,
                    const double* scale, const double* offset, double* g);
double klo_loss(const klo_csr* x, const uint8_t* labels, const double* theta,
                const double cw[2], double lambda,
                const double* scale, const double* offset);

/* order[0..m) = permutation of 0..m-1 sorted by |g| descending, ties by lower index */
void   klo_sort_abs_desc(const double* g, int64_t m, int64_t* order);
double klo_compute_lambda(const double* g, int64_t m, const int64_t* order, int64_t N);
int    klo_select(const double* g, int64_t m, uint8_t* b, int64_t c_prev, int64_t N,
                  double lambda_prev, double eps_lambda,
                  double* lambda_out, int64_t* c_out);
int    klo_select_fixed(const double* g_full, int64_t m, uint8_t* b, int64_t c_prev,
                        double lambda, int64_t max_features, int64_t* c_out);

/* featureSelection.Data (+ Transform.Apply).  Outputs are malloc'ed; free with klo_free. */
int    klo_select_columns(const klo_csr* x, const int64_t* cols, int64_t c_plus_1,
                          const double* scale, const double* offset,
                          int64_t** rowptr_out, int64_t** col_out, double** val_out);
void   klo_free(void* p);

/* CoeffIndex (kmerLr_coefficients_index.go:26-54): parameter index of a feature pair in co-occurrence mode.
 * n = number of base features (without bias); singles at k+1, pairs (k1<k2) after them, lexicographic. */
int64_t klo_coeff_dim(int64_t n);
int64_t klo_coeff_ind2sub(int64_t n, int64_t k1, int64_t k2);
void    klo_coeff_sub2ind(int64_t n, int64_t i, int64_t* k1, int64_t* k2);
/* The co-occurrence feature matrix written out: row = {0:1, k+1: v_k, Ind2Sub(k1,k2): v_k1*v_k2} for the stored
 * entries of the base row (LinearPdf / Gradient with Cooccurrence, kmerLr_logistic_regression.go:69-85,183-195;
 * featureSelection.Data :325-330).  Outputs are malloc'ed; free with klo_free. */
int    klo_expand_cooccurrence(const klo_csr* x, int64_t** rowptr_out, int64_t** col_out, double** val_out);

int    klo_eval_stopping(const double* xs, const double* x1, int64_t len,
                         double epsilon, double* delta_out);

int    klo_cd_solve(const klo_csr* x, const uint8_t* labels, double* theta,
                    const double cw[2], double l1reg, double l2reg,
                    double eps, double eps_loss, int eval_loss, int64_t max_iter,
                    klo_iter_rec* recs, int64_t recs_cap, int64_t* n_recs);

/* TransformFull.Fit; kind 1 standardizer, 2 variance-scaler, 3 max-abs-scaler, 4 mean-scaler */
int    klo_transform_fit(const klo_csr* x, int kind, double* scale, double* offset, int* has_offset);

/* proximal-gradient step of kmerLr_estimator_proximal.go:84-98 (one iteration) */
void   klo_prox_step(double* theta, const double* g, int64_t len, double step, double l1reg);

/* deterministic synthetic generator of SURVEY.md section 8(d); see oracle/klr_synth.c */
uint64_t klo_splitmix64(uint64_t x);

#ifdef __cplusplus
}
#endif
#endif
