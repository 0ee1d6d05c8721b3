/*
 * klr_oracle_backend.c -- a klr_backend table (include/kmerlr_driver.h) served by the CPU oracle.
 * TEST INFRASTRUCTURE ONLY: it lets the CPU test-suite run the product's host driver
 * (kmerlr_amd/csrc/klr_driver.cpp, compiled with g++ into oracle/_build/libklr_oracle_driver.so)
 * end to end, single rank or row-sharded over gloo ranks.  It is never loaded by the product.
 */
#include "klr_oracle.h"
#include "../include/kmerlr_driver.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

struct klr_ctx { char err[256]; };
struct klr_mat {
  klo_csr x;                 /* the rows that take part in sums over samples (all rows, or the active subset) */
  int64_t* rowptr; int64_t* col; double* val; uint8_t* labels; int64_t n_total;
  /* all rows (owner arrays) */
  klo_csr all; int64_t* a_rowptr; int64_t* a_col; double* a_val; uint8_t* a_labels; int64_t n_total_all;
  uint8_t* mask;
  int shared;                /* ob_mat_view: the owner arrays belong to another matrix */
};

static const char* ob_last_error(const klr_ctx* c) { return c ? c->err : ""; }

klr_mat* klo_backend_mat(int64_t n, int64_t dim, const int64_t* rowptr, const int64_t* col, const double* val,
                         const uint8_t* labels, int64_t n_total) {
  klr_mat* m = (klr_mat*)calloc(1, sizeof(klr_mat));
  const int64_t nnz = rowptr[n];
  m->rowptr = (int64_t*)malloc((size_t)(n + 1) * 8); memcpy(m->rowptr, rowptr, (size_t)(n + 1) * 8);
  m->col = (int64_t*)malloc((size_t)(nnz > 0 ? nnz : 1) * 8); memcpy(m->col, col, (size_t)nnz * 8);
  m->val = (double*)malloc((size_t)(nnz > 0 ? nnz : 1) * 8); memcpy(m->val, val, (size_t)nnz * 8);
  m->labels = (uint8_t*)malloc((size_t)(n > 0 ? n : 1)); memcpy(m->labels, labels, (size_t)n);
  m->x.n = n; m->x.dim = dim; m->x.rowptr = m->rowptr; m->x.col = m->col; m->x.val = m->val;
  m->n_total = n_total ? n_total : n;
  m->all = m->x; m->a_rowptr = m->rowptr; m->a_col = m->col; m->a_val = m->val; m->a_labels = m->labels;
  m->n_total_all = m->n_total; m->mask = NULL;
  return m;
}

static void drop_view(klr_mat* m) {
  if (m->rowptr != m->a_rowptr) { free(m->rowptr); free(m->col); free(m->val); free(m->labels); }
  m->rowptr = m->a_rowptr; m->col = m->a_col; m->val = m->a_val; m->labels = m->a_labels;
  m->x = m->all; m->n_total = m->n_total_all;
  free(m->mask); m->mask = NULL;
}

static void ob_mat_destroy(klr_mat* m) {
  if (!m) return;
  drop_view(m);
  if (m->shared) { free(m); return; }
  free(m->a_rowptr); free(m->a_col); free(m->a_val); free(m->a_labels); free(m);
}

/* klr_mat_set_active_rows: the active rows as a filtered copy (what filterCvGroup builds upstream) */
static int ob_set_active_rows(klr_ctx* c, klr_mat* m, const uint8_t* mask, int64_t n_active_total) {
  (void)c;
  drop_view(m);
  if (!mask) return 0;
  const int64_t n = m->all.n;
  int64_t na = 0, nnz = 0;
  for (int64_t i = 0; i < n; i++) if (mask[i]) { na++; nnz += m->a_rowptr[i + 1] - m->a_rowptr[i]; }
  m->rowptr = (int64_t*)malloc((size_t)(na + 1) * 8);
  m->col = (int64_t*)malloc((size_t)(nnz > 0 ? nnz : 1) * 8);
  m->val = (double*)malloc((size_t)(nnz > 0 ? nnz : 1) * 8);
  m->labels = (uint8_t*)malloc((size_t)(na > 0 ? na : 1));
  m->mask = (uint8_t*)malloc((size_t)(n > 0 ? n : 1)); memcpy(m->mask, mask, (size_t)n);
  int64_t r = 0, o = 0;
  m->rowptr[0] = 0;
  for (int64_t i = 0; i < n; i++) if (mask[i]) {
    for (int64_t a = m->a_rowptr[i]; a < m->a_rowptr[i + 1]; a++) { m->col[o] = m->a_col[a]; m->val[o] = m->a_val[a]; o++; }
    m->labels[r] = m->a_labels[i]; r++; m->rowptr[r] = o;
  }
  m->x.n = na; m->x.dim = m->all.dim; m->x.rowptr = m->rowptr; m->x.col = m->col; m->x.val = m->val;
  m->n_total = n_active_total;
  return 0;
}

static int ob_log_pdf(klr_ctx* c, const klr_mat* m, const double* theta, double* out) {
  (void)c;
  for (int64_t i = 0; i < m->all.n; i++) out[i] = klo_class_log_pdf(&m->all, i, theta, NULL, NULL, 1);
  return 0;
}

static int ob_mat_shape(const klr_mat* m, int64_t* n, int64_t* dim, int64_t* nnz, int64_t* n_total) {
  if (n) *n = m->all.n;
  if (dim) *dim = m->all.dim;
  if (nnz) *nnz = m->all.rowptr[m->all.n];
  if (n_total) *n_total = m->n_total;
  return 0;
}

static int ob_class_weights(klr_ctx* c, const klr_mat* m, double cw[2]) {
  (void)c; klo_class_weights(m->labels, m->x.n, cw); return 0;
}

static int ob_gradient(klr_ctx* c, const klr_mat* m, const int64_t* ti, const double* tv, int64_t tn, const double cw[2],
                       const double* scale, const double* offset, double* g) {
  (void)c;
  double* th = (double*)calloc((size_t)m->x.dim, 8);
  for (int64_t k = 0; k < tn; k++) th[ti[k]] = tv[k];
  klo_gradient(&m->x, m->labels, th, cw, 0.0, scale, offset, g);
  free(th);
  return 0;
}

static int ob_loss(klr_ctx* c, const klr_mat* m, const double* theta, const double cw[2], double lambda, double* out) {
  (void)c; *out = klo_loss(&m->x, m->labels, theta, cw, lambda, NULL, NULL); return 0;
}

static int ob_select_columns(klr_ctx* c, const klr_mat* m, const int64_t* cols, int64_t c1, const double* scale,
                             const double* offset, klr_mat** out) {
  int64_t *rp, *ci; double* cv;
  klo_select_columns(&m->all, cols, c1, scale, offset, &rp, &ci, &cv);
  klr_mat* r = (klr_mat*)calloc(1, sizeof(klr_mat));
  r->rowptr = rp; r->col = ci; r->val = cv;
  r->labels = (uint8_t*)malloc((size_t)(m->all.n > 0 ? m->all.n : 1)); memcpy(r->labels, m->a_labels, (size_t)m->all.n);
  r->x.n = m->all.n; r->x.dim = c1; r->x.rowptr = rp; r->x.col = ci; r->x.val = cv; r->n_total = m->n_total_all;
  r->all = r->x; r->a_rowptr = rp; r->a_col = ci; r->a_val = cv; r->a_labels = r->labels; r->n_total_all = m->n_total_all;
  if (m->mask) ob_set_active_rows(c, r, m->mask, m->n_total);        /* reduced matrices inherit the split */
  *out = r;
  return 0;
}

static int ob_expand_cooccurrence(klr_ctx* c, const klr_mat* m, klr_mat** out) {
  int64_t *rp, *ci; double* cv;
  klo_expand_cooccurrence(&m->all, &rp, &ci, &cv);
  klr_mat* r = (klr_mat*)calloc(1, sizeof(klr_mat));
  r->rowptr = rp; r->col = ci; r->val = cv;
  r->labels = (uint8_t*)malloc((size_t)(m->all.n > 0 ? m->all.n : 1)); memcpy(r->labels, m->a_labels, (size_t)m->all.n);
  r->x.n = m->all.n; r->x.dim = klo_coeff_dim(m->all.dim - 1); r->x.rowptr = rp; r->x.col = ci; r->x.val = cv;
  r->n_total = m->n_total_all;
  r->all = r->x; r->a_rowptr = rp; r->a_col = ci; r->a_val = cv; r->a_labels = r->labels; r->n_total_all = m->n_total_all;
  if (m->mask) ob_set_active_rows(c, r, m->mask, m->n_total);
  *out = r;
  return 0;
}

static int ob_sweep_select(klr_ctx* c, const klr_mat* full, const klr_mat* red, const int64_t* red_cols, const double* theta_red,
                           int64_t c1, const double cw[2], int64_t N, double lambda_prev, double eps_lambda,
                           const double* scale, const double* offset, uint8_t* b, double* lambda_out, int64_t* c_out,
                           int* ok_out, double* loss_out, double* g_out) {
  (void)c;
  const int64_t dim = full->x.dim;
  double* th = (double*)calloc((size_t)dim, 8);
  double* g = (double*)malloc((size_t)dim * 8);
  int64_t c_prev = 0;
  memset(b, 0, (size_t)dim); b[0] = 1;
  th[0] = theta_red[0];
  if (red) for (int64_t p = 1; p < c1; p++) {
    th[red_cols[p]] = theta_red[p];
    if (theta_red[p] != 0.0) { b[red_cols[p]] = 1; c_prev++; }
  }
  klo_gradient(&full->x, full->labels, th, cw, 0.0, scale, offset, g);
  if (loss_out) *loss_out = klo_loss(&full->x, full->labels, th, cw, 0.0, scale, offset);
  *ok_out = klo_select(g + 1, dim - 1, b, c_prev, N, lambda_prev, eps_lambda, lambda_out, c_out);
  if (g_out) memcpy(g_out, g, (size_t)dim * 8);
  free(th); free(g);
  return 0;
}

static int ob_cd_solve(klr_ctx* c, const klr_mat* red, double* theta, const double cw[2], double l1reg, double l2reg,
                       double eps, double eps_loss, int eval_loss, int64_t max_iter, klr_iter_rec* recs, int64_t cap,
                       int64_t* nrecs) {
  (void)c;
  return klo_cd_solve(&red->x, red->labels, theta, cw, l1reg, l2reg, eps, eps_loss, eval_loss, max_iter,
                      (klo_iter_rec*)recs, cap, nrecs);
}

/* concurrent folds (klr_config.threads_cv): a context and a view of the data per host thread */
static int ob_ctx_fork(klr_ctx* c, klr_ctx** out) { (void)c; *out = (klr_ctx*)calloc(1, sizeof(klr_ctx)); return *out ? 0 : 1; }
static void ob_ctx_release(klr_ctx* c) { free(c); }
static int ob_mat_view(klr_ctx* c, const klr_mat* m, klr_mat** out) {
  (void)c;
  klr_mat* v = (klr_mat*)calloc(1, sizeof(klr_mat));
  if (!v) return 1;
  v->all = m->all; v->a_rowptr = m->a_rowptr; v->a_col = m->a_col; v->a_val = m->a_val; v->a_labels = m->a_labels;
  v->n_total_all = m->n_total_all;
  v->rowptr = v->a_rowptr; v->col = v->a_col; v->val = v->a_val; v->labels = v->a_labels;
  v->x = v->all; v->n_total = v->n_total_all; v->mask = NULL; v->shared = 1;
  *out = v;
  return 0;
}

void klo_backend_table(klr_backend* out) {
  static klr_ctx ctx;
  out->ctx = &ctx;
  out->last_error = ob_last_error;
  out->mat_shape = ob_mat_shape;
  out->mat_destroy = ob_mat_destroy;
  out->class_weights = ob_class_weights;
  out->gradient = ob_gradient;
  out->loss = ob_loss;
  out->select_columns = ob_select_columns;
  out->sweep_select = ob_sweep_select;
  out->cd_solve = ob_cd_solve;
  out->set_active_rows = ob_set_active_rows;
  out->log_pdf = ob_log_pdf;
  out->expand_cooccurrence = ob_expand_cooccurrence;
  out->transform_fit = 0;                                /* the driver then keeps the caller's transform for every fold */
  out->ctx_fork = ob_ctx_fork;
  out->ctx_release = ob_ctx_release;
  out->mat_view = ob_mat_view;
  out->solve_run_begin = 0;
  out->solve_run_end = 0;
}

void klo_backend_mat_destroy(klr_mat* m) { ob_mat_destroy(m); }
