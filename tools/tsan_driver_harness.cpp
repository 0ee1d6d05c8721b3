// ThreadSanitizer harness (host code only): the driver's threaded cross-validation and ensemble on the oracle backend
#include "../include/kmerlr_driver.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
extern "C" {
void klo_synth_csr(uint64_t seed, int64_t row0, int64_t n_local, int64_t n_total, int64_t m, int64_t draws, int64_t S,
                   int64_t* rowptr, int64_t* col, double* val, uint8_t* labels);
klr_mat* klo_backend_mat(int64_t n, int64_t dim, const int64_t* rowptr, const int64_t* col, const double* val, const uint8_t* labels, int64_t n_total);
void klo_backend_table(klr_backend* out);
void klo_backend_mat_destroy(klr_mat* m);
}
int main() {
  const int64_t n = 300, m = 48, draws = 20;
  std::vector<int64_t> rp(n + 1), col(n * (draws + 1)); std::vector<double> val(n * (draws + 1)); std::vector<uint8_t> lab(n);
  klo_synth_csr(41, 0, n, n, m, draws, 6, rp.data(), col.data(), val.data(), lab.data());
  klr_mat* mat = klo_backend_mat(n, m + 1, rp.data(), col.data(), val.data(), lab.data(), n);
  klr_backend be; memset(&be, 0, sizeof(be)); klo_backend_table(&be);
  for (int variant = 0; variant < 3; variant++) {
    klr_config cfg; klr_config_default(&cfg);
    cfg.threads_cv = 3; cfg.threads = 3; cfg.save_trace = 1;
    if (variant == 1) { cfg.validation_size = 0.2; cfg.ensemble_size = 2; }
    if (variant == 2) { cfg.cooccurrence = 1; cfg.copreselection = 5; }
    klr_estimator* e = nullptr;
    if (klr_estimator_create(&be, &cfg, &e) != 0) return 2;
    const int64_t la[2] = {3, 6};
    std::vector<double> pred(2 * n), ltr(2 * 5), lte(2 * 5); std::vector<int32_t> g(n);
    int rc = klr_estimator_crossvalidate(e, mat, 0, n, 5, 9, la, 2, nullptr, nullptr, pred.data(), g.data(), ltr.data(), lte.data());
    printf("variant %d cv rc %d classifiers %lld\n", variant, rc, (long long)klr_estimator_n_classifiers(e));
    if (variant == 0) { rc = klr_estimator_estimate_ensemble(e, mat, 0, n, 4, 2, la, 2, nullptr, nullptr); printf("ensemble rc %d\n", rc); }
    klr_estimator_destroy(e);
  }
  klo_backend_mat_destroy(mat);
  return 0;
}
