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
  klo_csr x; int64_t* rowptr; int64_t* col; double* val; uint8_t* labels; int64_t n_total;
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
  return m;
}

static void ob_mat_destroy(klr_mat* m) {
  if (!m) return;
  free(m->rowptr); free(m->col); free(m->val); free(m->labels); free(m);
}

static int ob_mat_shape(const klr_mat* m, int64_t* n, int64_t* dim, int64_t* nnz, int64_t* n_total) {
  if (n) *n = m->x.n;
  if (dim) *dim = m->x.dim;
  if (nnz) *nnz = m->x.rowptr[m->x.n];
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
  (void)c;
  int64_t *rp, *ci; double* cv;
  klo_select_columns(&m->x, cols, c1, scale, offset, &rp, &ci, &cv);
  klr_mat* r = (klr_mat*)calloc(1, sizeof(klr_mat));
  r->rowptr = rp; r->col = ci; r->val = cv;
  r->labels = (uint8_t*)malloc((size_t)(m->x.n > 0 ? m->x.n : 1)); memcpy(r->labels, m->labels, (size_t)m->x.n);
  r->x.n = m->x.n; r->x.dim = c1; r->x.rowptr = rp; r->x.col = ci; r->x.val = cv; r->n_total = m->n_total;
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
}

void klo_backend_mat_destroy(klr_mat* m) { ob_mat_destroy(m); }
