/*
 * klr_oracle.c -- CPU restatement (plain C, IEEE f64, single thread, reference loop
 * order) of the kmerLr hot path.  TEST INFRASTRUCTURE ONLY -- see klr_oracle.h.
 *
 * Every function cites the reference file:line (relative to /root/reference) whose
 * behaviour it restates.  Nothing here is copied from the reference (which is Go).
 */
#ifdef _OPENMP
#include <omp.h>
#endif
#include "klr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---------------------------------------------------------------------------------- */
/* row sharding (tests of the multi-rank driver only): when set, the rows held here are a shard of */
/* a data set with n_total rows and every sum over samples goes through the hook before use.      */
/* ---------------------------------------------------------------------------------- */
static int64_t g_n_total = 0;
static klo_allreduce_fn g_ar = NULL;
static void* g_ar_user = NULL;

void klo_set_sharding(int64_t n_total, klo_allreduce_fn fn, void* user) {
  g_n_total = n_total; g_ar = fn; g_ar_user = user;
}
static inline double eff_n(int64_t n) { return (double)(g_n_total > 0 ? g_n_total : n); }
static inline void ar(double* buf, int64_t cnt) { if (g_ar) g_ar(buf, cnt, g_ar_user); }

/* ---------------------------------------------------------------------------------- */
/* third-party arithmetic restated from its published semantics                       */
/* ---------------------------------------------------------------------------------- */

/* github.com/pbenner/autodiff v1.0.0, logarithmetic.LogAdd: log(exp(a)+exp(b)),
 * evaluated as max + log1p(exp(min-max)).  Call sites: kmerLr_logistic_regression.go:141,143,
 * kmerLr_estimator_coordinate.go:99.  Pinned by the loss goldens G3/G7. */
double klo_log_add(double a, double b) {
  if (a > b) { double t = a; a = b; b = t; }
  if (isinf(a) && a < 0) return b;
  return b + log1p(exp(a - b));
}

/* ---------------------------------------------------------------------------------- */
/* L1: logistic-regression arithmetic (kmerLr_logistic_regression.go)                 */
/* ---------------------------------------------------------------------------------- */

/* TransformFull.Apply, kmerLr_transform.go:260-268 */
static inline double apply_tr(double v, int64_t j, const double* scale, const double* offset) {
  if (offset) v = v - offset[j];
  if (scale)  v = v * scale[j];
  return v;
}

/* logisticRegression.LinearPdf, kmerLr_logistic_regression.go:47-136 (non-co-occurrence) */
double klo_linear_pdf(const klo_csr* x, int64_t i, const double* theta,
                      const double* scale, const double* offset) {
  const int64_t  b = x->rowptr[i], e = x->rowptr[i + 1];
  const int64_t* ci = x->col + b;
  const double*  cv = x->val + b;
  const int64_t  q = e - b;
  double r = theta[0];                                   /* :52 */
  if (q == 0 || ci[0] != 0) return NAN;                  /* :53 panics; NaN here */
  if (!scale && !offset) {
    for (int64_t j = 1; j < q; j++) r += cv[j] * theta[ci[j]];                 /* :66-68 */
  } else if (!offset) {
    for (int64_t j = 1; j < q; j++) r += (cv[j] * scale[ci[j]]) * theta[ci[j]]; /* :88-90 */
  } else {
    /* iterate over the full vector to account for zeros, :110-116 */
    int64_t j = 1;
    for (int64_t k = 1; k < x->dim; k++) {
      if (j < q && ci[j] == k) { r += apply_tr(cv[j], k, scale, offset) * theta[k]; j++; }
      else                     { r += apply_tr(0.0,   k, scale, offset) * theta[k]; }
    }
  }
  return r;
}

/* ClassLogPdf, kmerLr_logistic_regression.go:138-145 */
double klo_class_log_pdf(const klo_csr* x, int64_t i, const double* theta,
                         const double* scale, const double* offset, int y) {
  double r = klo_linear_pdf(x, i, theta, scale, offset);
  return y ? -klo_log_add(0.0, -r) : -klo_log_add(0.0, r);
}

/* compute_class_weights, kmerLr_data.go:178-193 */
void klo_class_weights(const uint8_t* labels, int64_t n, double cw[2]) {
  int64_t n1 = 0, n0 = 0;
  for (int64_t i = 0; i < n; i++) { if (labels[i]) n1++; else n0++; }
  if (g_ar) { double c[2] = {(double)n0, (double)n1}; ar(c, 2); n0 = (int64_t)c[0]; n1 = (int64_t)c[1]; }
  cw[0] = (double)(n0 + n1) / (double)(2 * n0);
  cw[1] = (double)(n0 + n1) / (double)(2 * n1);
}

/* logisticRegression.Gradient, kmerLr_logistic_regression.go:151-248 (non-co-occurrence).
 * g has length dim and is overwritten. */
int klo_gradient(const klo_csr* x, const uint8_t* labels, const double* theta,
                 const double cw[2], double lambda,
                 const double* scale, const double* offset, double* g) {
  const int64_t n = x->n, dim = x->dim;
  if (n == 0 && !g_ar) return 1;                                          /* :152-154 */
  const double nn = eff_n(n);                                             /* len(data) of the whole set */
  for (int64_t j = 0; j < dim; j++) g[j] = 0.0;                           /* :162-164 */
  for (int64_t i = 0; i < n; i++) {                                       /* :166 */
    const double r = klo_class_log_pdf(x, i, theta, scale, offset, 1);    /* :168 LogPdf */
    const int64_t  b = x->rowptr[i], q = x->rowptr[i + 1] - b;
    const int64_t* ci = x->col + b;
    const double*  cv = x->val + b;
    double w;
    if (labels[i]) w = 1.0 / nn * cw[1] * (exp(r) - 1.0);                 /* :175 */
    else           w = 1.0 / nn * cw[0] * (exp(r));                       /* :177 */
    if (!scale && !offset) {
      for (int64_t j = 0; j < q; j++) g[ci[j]] += w * cv[j];              /* :180-182 */
    } else if (!offset) {
      for (int64_t j = 0; j < q; j++) g[ci[j]] += w * (cv[j] * scale[ci[j]]); /* :198-200 */
    } else {
      int64_t j = 0;                                                      /* :216-222 */
      for (int64_t k = 0; k < dim; k++) {
        if (j < q && ci[j] == k) { g[k] += w * apply_tr(cv[j], k, scale, offset); j++; }
        else                     { g[k] += w * apply_tr(0.0,   k, scale, offset); }
      }
    }
  }
  ar(g, dim);
  if (!isnan(lambda) && lambda != 0.0) {                                  /* :237-246 */
    for (int64_t j = 1; j < dim; j++) {
      if (theta[j] < 0) g[j] -= lambda; else if (theta[j] > 0) g[j] += lambda;
    }
  }
  return 0;
}

/* The same loop (kmerLr_logistic_regression.go:166-182, no transform) with the rows split over `threads` OpenMP
 * threads, every thread adding into a gradient vector of its own, the vectors summed in thread order afterwards.
 * NOT what the reference does (its loop is single-threaded): bench.py's "all_cores" figure, i.e. what a
 * multi-threaded port of the reference's loop reaches on the host.  Sum order differs from klo_gradient by the
 * blocking, so results agree to rounding, not bit for bit. */
int klo_gradient_mt(const klo_csr* x, const uint8_t* labels, const double* theta,
                    const double cw[2], double* g, int threads) {
  const int64_t n = x->n, dim = x->dim;
  if (n == 0) return 1;
  if (threads < 1) threads = 1;
  const double nn = eff_n(n);
  double* gt = (double*)calloc((size_t)threads * (size_t)dim, sizeof(double));
  if (!gt) return 2;
#pragma omp parallel num_threads(threads)
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num(), T = omp_get_num_threads();
#else
    const int t = 0, T = 1;
#endif
    double* gl = gt + (size_t)t * (size_t)dim;
    const int64_t r0 = n * t / T, r1 = n * (t + 1) / T;
    for (int64_t i = r0; i < r1; i++) {
      const double r = klo_class_log_pdf(x, i, theta, NULL, NULL, 1);
      const int64_t  b = x->rowptr[i], q = x->rowptr[i + 1] - b;
      const int64_t* ci = x->col + b;
      const double*  cv = x->val + b;
      const double w = labels[i] ? 1.0 / nn * cw[1] * (exp(r) - 1.0) : 1.0 / nn * cw[0] * (exp(r));
      for (int64_t j = 0; j < q; j++) gl[ci[j]] += w * cv[j];
    }
  }
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int64_t j = 0; j < dim; j++) {
    double v = 0.0;
    for (int t = 0; t < threads; t++) v += gt[(size_t)t * (size_t)dim + (size_t)j];
    g[j] = v;
  }
  free(gt);
  ar(g, dim);
  return 0;
}

/* logisticRegression.Loss, kmerLr_logistic_regression.go:250-272 */
double klo_loss(const klo_csr* x, const uint8_t* labels, const double* theta,
                const double cw[2], double lambda,
                const double* scale, const double* offset) {
  const int64_t n = x->n;
  if (n == 0 && !g_ar) return 0.0;
  double r = 0.0;
  for (int64_t i = 0; i < n; i++) {                                       /* :258-264 */
    const int y = labels[i] != 0;
    r -= cw[y] * klo_class_log_pdf(x, i, theta, scale, offset, y);
  }
  ar(&r, 1);
  r = r / eff_n(n);                                                       /* :265 */
  if (!isnan(lambda) && lambda != 0.0) {
    for (int64_t j = 1; j < x->dim; j++) r += lambda * fabs(theta[j]);    /* :267-269 */
  }
  return r;
}

/* ---------------------------------------------------------------------------------- */
/* L3: feature selection (kmerLr_feature_selection.go, kmerLr_sort.go)                */
/* ---------------------------------------------------------------------------------- */

typedef struct { double a; int64_t i; } absidx;

static int cmp_abs_desc(const void* pa, const void* pb) {
  const absidx* x = (const absidx*)pa; const absidx* y = (const absidx*)pb;
  const int xn = isnan(x->a), yn = isnan(y->a);
  if (xn || yn) { if (xn != yn) return xn ? 1 : -1; }
  else if (x->a != y->a) return x->a > y->a ? -1 : 1;
  return (x->i > y->i) - (x->i < y->i);
}

/* NLargestAbsFloat64 / AbsFloatInt.SortReverse, kmerLr_sort.go:85-131.  The reference's
 * sort.Sort is unstable, so its order among equal |g| is toolchain dependent; the rule
 * here (and in the HIP path) is: larger |g| first, then LOWER column index; NaN last. */
void klo_sort_abs_desc(const double* g, int64_t m, int64_t* order) {
  absidx* t = (absidx*)malloc((size_t)(m > 0 ? m : 1) * sizeof(absidx));
  for (int64_t j = 0; j < m; j++) { t[j].a = fabs(g[j]); t[j].i = j; }
  qsort(t, (size_t)m, sizeof(absidx), cmp_abs_desc);
  for (int64_t j = 0; j < m; j++) order[j] = t[j].i;
  free(t);
}

/* featureSelector.computeLambda, kmerLr_feature_selection.go:179-192.  g = gradient[1:]
 * (length m), order = klo_sort_abs_desc(g).  The reference hands in the first 2N sorted
 * entries, so `N > len(g)` means N > min(m, 2N), i.e. N > m. */
double klo_compute_lambda(const double* g, int64_t m, const int64_t* order, int64_t N) {
  const int64_t len = m < 2 * N ? m : 2 * N;
  if (N > len) return 0.0;                                                /* :180-182 */
  const double v = fabs(g[order[N - 1]]);                                 /* :183 */
  double w = 0.0;
  for (int64_t k = 0; k < m; k++) {                                       /* :186-190 */
    const double t = fabs(g[k]);
    if (t > w && t < v) w = t;
  }
  return (v + w) / 2.0;                                                   /* :191 */
}

/* featureSelector.Select, kmerLr_feature_selection.go:78-134.
 *   g          gradient[1:] at the restored theta (length m)
 *   b          in: mask of length m+1 after restoreNonzero (:194-219; b[0]=1, b[j]=1 where
 *              theta_j != 0); out: mask after selection
 *   c_prev     number of nonzero non-bias coefficients restored
 *   lambda_prev the estimator's L1Reg of the previous epoch (units: lambda*n, as in the
 *              reference call kmerLr_estimator.go:235)
 * returns ok (:133). */
int klo_select(const double* g, int64_t m, uint8_t* b, int64_t c_prev, int64_t N,
               double lambda_prev, double eps_lambda,
               double* lambda_out, int64_t* c_out) {
  int64_t* order = (int64_t*)malloc((size_t)(m > 0 ? m : 1) * sizeof(int64_t));
  klo_sort_abs_desc(g, m, order);                                         /* :93 */
  const int64_t len = m < 2 * N ? m : 2 * N;
  int64_t c = c_prev;
  int ok = 0;
  for (int64_t k = 0; k < len; k++) {                                     /* :95-105 */
    if (c >= N) break;
    if (!b[order[k] + 1] && g[order[k]] != 0.0) { ok = 1; b[order[k] + 1] = 1; c++; }
  }
  for (int64_t k = 0; k < len; k++) {                                     /* :107-116 */
    if (c >= N) break;
    if (!b[order[k] + 1]) { b[order[k] + 1] = 1; c++; }
  }
  if (c > N) ok = 1;                                                      /* :117-119 */
  const double l = klo_compute_lambda(g, m, order, N);                    /* :130 */
  free(order);
  *lambda_out = l; *c_out = c;
  return ok || (eps_lambda > 0.0 && fabs(lambda_prev - l) >= eps_lambda); /* :133 */
}

/* featureSelector.SelectFixed, kmerLr_feature_selection.go:136-159.  g_full = full
 * gradient (length m+1, index 0 = bias). */
int klo_select_fixed(const double* g_full, int64_t m, uint8_t* b, int64_t c_prev,
                     double lambda, int64_t max_features, int64_t* c_out) {
  int ok = 0; int64_t c = c_prev;
  for (int64_t k = 1; k < m + 1; k++) {                                   /* :146 */
    if (fabs(g_full[k]) >= lambda && !b[k]) { ok = 1; b[k] = 1; c++; }    /* :147-151 */
    if (max_features > 0 && c >= max_features && ok) break;               /* :152-154 */
  }
  *c_out = c;
  return ok;
}

/* CoeffIndex.Dim / Ind2Sub / Sub2Ind, kmerLr_coefficients_index.go:30-54 */
int64_t klo_coeff_dim(int64_t n) { return (n + 1) * n / 2 + 1; }                       /* :30-33 */
int64_t klo_coeff_ind2sub(int64_t n, int64_t k1, int64_t k2) {                          /* :35-42 */
  if (k1 == k2) return k1 + 1;
  return n + (n * (n - 1) / 2) - (n - k1) * ((n - k1) - 1) / 2 + k2 - k1;
}
void klo_coeff_sub2ind(int64_t n, int64_t i, int64_t* k1, int64_t* k2) {                /* :44-54 */
  if (i < n) { *k1 = i; *k2 = i; return; }
  i -= n;
  const int64_t a = n - 2 - (int64_t)floor(sqrt((double)(-8 * i + 4 * n * (n - 1) - 7)) / 2.0 - 0.5);
  *k1 = a;
  *k2 = i + a + 1 - n * (n - 1) / 2 + (n - a) * ((n - a) - 1) / 2;
}

/* The explicit form of what LinearPdf / Gradient compute on the fly with Cooccurrence set
 * (kmerLr_logistic_regression.go:69-85 and :183-195): every pair j1 < j2 of stored non-bias entries of a row
 * contributes v[j1]*v[j2] at parameter Ind2Sub(i[j1]-1, i[j2]-1).  Pairs come out in ascending column order
 * because Ind2Sub is lexicographic in (k1, k2). */
int klo_expand_cooccurrence(const klo_csr* x, int64_t** rowptr_out, int64_t** col_out, double** val_out) {
  const int64_t n = x->n, nb = x->dim - 1;
  int64_t* rp = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));
  rp[0] = 0;
  for (int64_t i = 0; i < n; i++) {
    const int64_t q = x->rowptr[i + 1] - x->rowptr[i] - 1;            /* stored non-bias entries */
    rp[i + 1] = rp[i] + 1 + q + q * (q - 1) / 2;
  }
  const int64_t cap = rp[n];
  int64_t* ci = (int64_t*)malloc((size_t)(cap > 0 ? cap : 1) * sizeof(int64_t));
  double*  cv = (double*)malloc((size_t)(cap > 0 ? cap : 1) * sizeof(double));
  for (int64_t i = 0; i < n; i++) {
    const int64_t b = x->rowptr[i], e = x->rowptr[i + 1];
    int64_t o = rp[i];
    for (int64_t p = b; p < e; p++) { ci[o] = x->col[p]; cv[o] = x->val[p]; o++; }       /* bias + singles */
    for (int64_t p1 = b + 1; p1 < e; p1++)
      for (int64_t p2 = p1 + 1; p2 < e; p2++) {
        ci[o] = klo_coeff_ind2sub(nb, x->col[p1] - 1, x->col[p2] - 1);
        cv[o] = x->val[p1] * x->val[p2];
        o++;
      }
  }
  *rowptr_out = rp; *col_out = ci; *val_out = cv;
  return 0;
}

/* featureSelection.Data, kmerLr_feature_selection.go:309-343, followed by Transform.Apply
 * (kmerLr_transform.go:584-629) with the already-selected scale/offset (length c+1).
 * cols = ascending selected column ids incl. 0. */
int klo_select_columns(const klo_csr* x, const int64_t* cols, int64_t c1,
                       const double* scale, const double* offset,
                       int64_t** rowptr_out, int64_t** col_out, double** val_out) {
  const int64_t n = x->n;
  int64_t* rp = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));
  int64_t cap = 0;
  /* pass 1: count */
  rp[0] = 0;
  for (int64_t i = 0; i < n; i++) {
    int64_t cnt = 0;
    if (offset) cnt = c1;                                                 /* transform.go:600-609 densifies */
    else {
      int64_t j = x->rowptr[i], e = x->rowptr[i + 1];
      for (int64_t p = 0; p < c1; p++) {
        while (j < e && x->col[j] < cols[p]) j++;
        if (j < e && x->col[j] == cols[p] && x->val[j] != 0.0) cnt++;     /* :332 */
      }
    }
    rp[i + 1] = rp[i] + cnt;
  }
  cap = rp[n];
  int64_t* ci = (int64_t*)malloc((size_t)(cap > 0 ? cap : 1) * sizeof(int64_t));
  double*  cv = (double*)malloc((size_t)(cap > 0 ? cap : 1) * sizeof(double));
  for (int64_t i = 0; i < n; i++) {
    int64_t o = rp[i];
    int64_t j = x->rowptr[i], e = x->rowptr[i + 1];
    for (int64_t p = 0; p < c1; p++) {
      while (j < e && x->col[j] < cols[p]) j++;
      const int present = (j < e && x->col[j] == cols[p]);
      if (offset) {
        double v = present ? x->val[j] : 0.0;
        v = v - offset[p];
        if (scale) v = v * scale[p];
        ci[o] = p; cv[o] = v; o++;
      } else if (present && x->val[j] != 0.0) {
        ci[o] = p;                                                        /* :333 remapped index */
        cv[o] = scale ? x->val[j] * scale[p] : x->val[j];                 /* transform.go:622-625 */
        o++;
      }
    }
  }
  *rowptr_out = rp; *col_out = ci; *val_out = cv;
  return 0;
}

void klo_free(void* p) { free(p); }

/* ---------------------------------------------------------------------------------- */
/* L2: solver (kmerLr_estimator_coordinate.go, _proximal.go, _hook.go)                */
/* ---------------------------------------------------------------------------------- */

/* eval_stopping, kmerLr_estimator_proximal.go:30-52 */
int klo_eval_stopping(const double* xs, const double* x1, int64_t len,
                      double epsilon, double* delta_out) {
  double max_x = 0.0, max_delta = 0.0, delta;
  for (int64_t i = 0; i < len; i++) {
    if (isnan(x1[i])) { *delta_out = NAN; return 1; }                     /* :36-38 */
    max_x     = fmax(max_x, fabs(x1[i]));
    max_delta = fmax(max_delta, fabs(x1[i] - xs[i]));
  }
  delta = (max_x != 0.0) ? max_delta / max_x : max_delta;                 /* :42-46 */
  *delta_out = delta;
  if ((max_x != 0.0 && max_delta / max_x <= epsilon) ||
      (max_x == 0.0 && max_delta == 0.0)) return 1;                       /* :47-50 */
  return 0;
}

typedef struct {
  const klo_csr* x; const uint8_t* labels; const double* cw;
  double eps_loss; int eval_loss;
  double loss_old, loss_new;
  klo_iter_rec* recs; int64_t cap; int64_t count;
  struct timespec t0;
} hook_state;

/* NewHook closure, kmerLr_estimator_hook.go:33-101 (adaptive step size is SAGA-only) */
static int hook_call(hook_state* h, const double* theta, double change, double l1reg, int64_t iteration) {
  const double lambda = l1reg / eff_n(h->x->n);                           /* :47 */
  double t = h->loss_old; h->loss_old = h->loss_new; h->loss_new = t;     /* :48 */
  if (h->eval_loss) h->loss_new = klo_loss(h->x, h->labels, theta, h->cw, lambda, NULL, NULL); /* :50-51 */
  int64_t nz = 0;
  for (int64_t j = 0; j < h->x->dim; j++) if (theta[j] != 0.0) nz++;      /* :69-73 */
  if (h->recs && h->count < h->cap) {                                     /* :74-76 */
    struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
    klo_iter_rec* r = &h->recs[h->count];
    r->iteration = (int32_t)iteration; r->nonzero = (int32_t)nz; r->change = change;
    r->lambda = lambda; r->loss = h->loss_new;
    r->seconds = (double)(t1.tv_sec - h->t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - h->t0.tv_nsec);
  }
  h->count++;
  if (h->eps_loss == 0.0) return 0;                                       /* :94-95 */
  return fabs(h->loss_old - h->loss_new) < h->eps_loss;                   /* :97 (NaN compares false) */
}

/* estimate_coordinate_loop, kmerLr_estimator_coordinate.go:31-84: covariance-update
 * cyclic coordinate descent on the weighted least-squares surrogate. */
static int64_t coordinate_loop(const klo_csr* x, const double* z, const double* w,
                               double* theta0, double* theta1, double l1reg, double l2reg,
                               double eps, int64_t max_iter, int64_t iter,
                               hook_state* h, int* hook_stop,
                               double* xy, double* norm, double* G) {
  const int64_t d = x->dim;
  memset(xy, 0, (size_t)d * sizeof(double));
  memset(norm, 0, (size_t)d * sizeof(double));
  memset(G, 0, (size_t)(d * d) * sizeof(double));
  for (int64_t i = 0; i < x->n; i++) {                                    /* :39-52 */
    for (int64_t a = x->rowptr[i]; a < x->rowptr[i + 1]; a++) {
      const int64_t j1 = x->col[a]; const double v1 = x->val[a];
      xy[j1]   += w[i] * z[i] * v1;                                       /* :43 */
      norm[j1] += w[i] * v1 * v1;                                         /* :45 */
      for (int64_t bq = x->rowptr[i]; bq < x->rowptr[i + 1]; bq++)        /* :47-50 */
        G[j1 * d + x->col[bq]] += w[i] * v1 * x->val[bq];
    }
  }
  ar(xy, d); ar(norm, d); ar(G, d * d);
  for (; iter < max_iter; iter++) {                                       /* :53 */
    for (int64_t j = 0; j < d; j++) {                                     /* :55-72 */
      double t = xy[j] + norm[j] * theta1[j];
      for (int64_t k = 0; k < d; k++) t -= G[j * d + k] * theta1[k];
      if (j > 0) {
        if (t >= 0.0) t =  fmax(fabs(t) - l1reg, 0.0);
        else          t = -fmax(fabs(t) - l1reg, 0.0);
      }
      t /= (norm[j] + l2reg);
      theta0[j] = theta1[j];
      theta1[j] = t;
    }
    double delta;
    if (klo_eval_stopping(theta0, theta1, d, eps, &delta)) break;         /* :74-75 */
    if (hook_call(h, theta1, delta, l1reg, iter)) { *hook_stop = 1; break; } /* :78-80 */
  }
  return iter;
}

/* estimate_coordinate, kmerLr_estimator_coordinate.go:86-139 -- the INTENDED algorithm:
 * IRLS outer loop + covariance-update CD.  Deliberate differences from the dead code, all
 * needed for it to converge at all (DESIGN.md "solver semantics"):
 *   - real copies where the reference aliases one backing array (:89-91, which makes
 *     eval_stopping compare a vector with itself and stop after one sweep);
 *   - the caller's class weights (the reference recomputes balanced weights at :88 while
 *     its selector and hook use obj.ClassWeights);
 *   - w guarded by KLO_IRLS_WMIN (the reference divides by an unguarded p(1-p), :101-105);
 *   - the hook's loss verdict inside the sweep loop ends the SWEEP loop only (:78-80 as
 *     written); the outer loop then stops on eval_stopping over the whole IRLS iteration
 *     (:119-120) or when the loss moved less than eps_loss between two consecutive IRLS
 *     iterations.  (As written, :123-125 re-runs the same hook on the same theta, whose
 *     loss difference is exactly 0, so the reference would stop after ONE re-weighting.)
 * Objective: sum_i cw_i l_i(theta) + l1reg*|theta_1:|_1 + l2reg/2 |theta|^2, l1reg = lambda*n.
 * One record per sweep (the hook call of :78).  Returns 0. */
int klo_cd_solve(const klo_csr* x, const uint8_t* labels, double* theta,
                 const double cw[2], double l1reg, double l2reg,
                 double eps, double eps_loss, int eval_loss, int64_t max_iter,
                 klo_iter_rec* recs, int64_t recs_cap, int64_t* n_recs) {
  const int64_t n = x->n, d = x->dim;
  if (max_iter <= 0) max_iter = INT32_MAX;
  double* w  = (double*)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
  double* z  = (double*)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
  double* tp = (double*)malloc((size_t)d * sizeof(double));
  double* t0 = (double*)malloc((size_t)d * sizeof(double));
  double* xy = (double*)malloc((size_t)d * sizeof(double));
  double* nm = (double*)malloc((size_t)d * sizeof(double));
  double* G  = (double*)malloc((size_t)(d * d) * sizeof(double));
  hook_state h; memset(&h, 0, sizeof(h));
  h.x = x; h.labels = labels; h.cw = cw; h.eps_loss = eps_loss; h.eval_loss = eval_loss;
  h.loss_old = NAN; h.loss_new = NAN; h.recs = recs; h.cap = recs_cap; h.count = 0;
  clock_gettime(CLOCK_MONOTONIC, &h.t0);
  double loss_outer_prev = NAN;
  for (int64_t iter = 0; iter < max_iter; iter++) {                       /* :95 */
    for (int64_t i = 0; i < n; i++) {                                     /* :97-111 */
      const double r = klo_linear_pdf(x, i, theta, NULL, NULL);
      const double p = exp(-klo_log_add(0.0, -r));
      double wi = p * (1.0 - p);
      if (wi < KLO_IRLS_WMIN) wi = KLO_IRLS_WMIN;
      z[i] = r + ((labels[i] ? 1.0 : 0.0) - p) / wi;
      w[i] = wi * cw[labels[i] != 0];
    }
    memcpy(tp, theta, (size_t)d * sizeof(double));                        /* :113-115 */
    int hook_stop = 0;
    iter = coordinate_loop(x, z, w, t0, theta, l1reg, l2reg, eps, max_iter, iter,
                           &h, &hook_stop, xy, nm, G);                    /* :116 */
    double delta;
    if (klo_eval_stopping(tp, theta, d, eps, &delta)) break;              /* :119-120 */
    if (eval_loss && eps_loss != 0.0) {
      const double lc = klo_loss(x, labels, theta, cw, l1reg / eff_n(n), NULL, NULL);
      if (fabs(loss_outer_prev - lc) < eps_loss) break;
      loss_outer_prev = lc;
    }
  }
  if (n_recs) *n_recs = h.count;
  free(w); free(z); free(tp); free(t0); free(xy); free(nm); free(G);
  return 0;
}

/* one iteration of estimate_proximal, kmerLr_estimator_proximal.go:84-98 */
void klo_prox_step(double* theta, const double* g, int64_t len, double step, double l1reg) {
  for (int64_t k = 0; k < len; k++) {
    double t = theta[k] - step * g[k];
    if (k > 0) {
      if (t >= 0.0) t =  fmax(fabs(t) - step * l1reg, 0.0);
      else          t = -fmax(fabs(t) - step * l1reg, 0.0);
    }
    theta[k] = t;
  }
}

/* TransformFull.Fit (kmerLr_transform.go:59-252, non-co-occurrence).  kind: 1 standardizer, 2 variance-scaler,
 * 3 max-abs-scaler, 4 mean-scaler.  scale/offset have length dim; *has_offset tells whether Offset is kept.
 * With sharding on, n and every per-column sum are over all ranks (the max through a sum of indicator-free
 * two-pass is not needed here: max-abs uses the hook on a copy raised to... see klo code). */
int klo_transform_fit(const klo_csr* x, int kind, double* scale, double* offset, int* has_offset) {
  const int64_t n = x->n, dim = x->dim;
  const double nn = eff_n(n);
  for (int64_t j = 0; j < dim; j++) { scale[j] = 0.0; offset[j] = 0.0; }
  *has_offset = 0;
  if (kind == 1 || kind == 2) {                                          /* fitStandardizer :59-146 */
    double* k = (double*)calloc((size_t)dim, sizeof(double));
    for (int64_t i = 0; i < n; i++)
      for (int64_t a = x->rowptr[i] + 1; a < x->rowptr[i + 1]; a++) offset[x->col[a]] += x->val[a];   /* :87-89 */
    ar(offset, dim);
    for (int64_t j = 1; j < dim; j++) offset[j] /= nn;                   /* :103-105 */
    for (int64_t i = 0; i < n; i++)
      for (int64_t a = x->rowptr[i] + 1; a < x->rowptr[i + 1]; a++) {    /* :116-120 */
        const int64_t j = x->col[a];
        k[j] += 1.0;
        scale[j] += (x->val[a] - offset[j]) * (x->val[a] - offset[j]);
      }
    ar(k, dim); ar(scale, dim);
    for (int64_t j = 1; j < dim; j++) {                                  /* :134-143 */
      const double sj = scale[j] + (nn - k[j]) * offset[j] * offset[j];
      scale[j] = (sj == 0.0) ? 1.0 : sqrt(sj / (nn - 1.0));
      scale[j] = 1.0 / scale[j];
    }
    offset[0] = 0.0; scale[0] = 1.0;                                     /* :144-145 */
    free(k);
    *has_offset = (kind == 1);                                           /* FitVarianceScaler drops Offset :157 */
    if (kind == 2) for (int64_t j = 0; j < dim; j++) offset[j] = 0.0;
  } else if (kind == 3) {                                                /* FitMaxAbsScaler :161-204 */
    for (int64_t i = 0; i < n; i++)
      for (int64_t a = x->rowptr[i] + 1; a < x->rowptr[i + 1]; a++)
        scale[x->col[a]] = fmax(scale[x->col[a]], fabs(x->val[a]));
    for (int64_t j = 1; j < dim; j++) scale[j] = 1.0 / scale[j];
    scale[0] = 1.0;
  } else if (kind == 4) {                                                /* FitMeanScaler :206-250 */
    for (int64_t i = 0; i < n; i++)
      for (int64_t a = x->rowptr[i] + 1; a < x->rowptr[i + 1]; a++) scale[x->col[a]] += x->val[a];
    ar(scale, dim);
    for (int64_t j = 1; j < dim; j++) scale[j] = nn / scale[j];
    scale[0] = 1.0;
  } else return 1;
  return 0;
}

uint64_t klo_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}
