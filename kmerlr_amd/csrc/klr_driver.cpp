// klr_driver.cpp -- leapfrog path driver, feature-selection bookkeeping, hook trace and the .trace /
// .path / model writers (include/kmerlr_driver.h).  Backend agnostic: plain C++17, no HIP; every
// device operation goes through the klr_backend table.
//
// Mirrors KmerLrEstimator / ScoresLrEstimator (kmerLr_estimator.go:147-270,
// scoresLr_estimator.go:126-240), featureSelection.Theta (kmerLr_feature_selection.go:299-307),
// Trace.Export (kmerLr_estimator_trace.go:49-78) and RegularizationPath
// (kmerLr_estimator_path.go:41-117).
#include "../../include/kmerlr_driver.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Classifier {
  std::vector<int64_t> cols;
  std::vector<double> theta;
  double lambda = 0.0;
  std::vector<double> scale, offset;
};

struct Path {   // ScoresRegularizationPath / KmerRegularizationPath
  std::vector<double> lambda, norm;
  std::vector<std::vector<double>> theta;
  std::vector<int64_t> idx;
  std::map<int64_t, size_t> index;

  void append(double lam, const std::vector<int64_t>& feat, const std::vector<double>& th) {   // path.go:79-110
    double v = 0.0;
    std::vector<double> t(index.size(), 0.0);
    for (size_t j = 0; j < th.size(); j++) {
      if (th[j] == 0.0) continue;
      v += std::fabs(th[j]);
      auto it = index.find(feat[j]);
      if (it == index.end()) {
        index[feat[j]] = index.size();
        idx.push_back(feat[j]);
        t.push_back(th[j]);
      } else {
        t[it->second] = th[j];
      }
    }
    lambda.push_back(lam); norm.push_back(v); theta.push_back(t);
    for (auto& row : theta) row.resize(t.size(), 0.0);    // augment previous entries with zeroes
  }
};

}  // namespace

struct klr_estimator {
  klr_backend be;
  klr_config cfg;
  std::string err;
  // estimator state, kept across lambda_auto values (warm start, kmerLr_estimator.go:263-267)
  std::vector<int64_t> cols{0};
  std::vector<double> theta{0.0};
  double l1reg = 0.0;
  double cw[2] = {1.0, 1.0};
  std::vector<Classifier> out;
  std::vector<klr_iter_rec> trace;
  Path path;
  std::vector<double> ep_lambda;
  std::vector<int64_t> ep_nonzero;
  // k-mer counter of the data (klr_estimator_set_kmer_features): kmerLr models are written with the reference's
  // KmerLrFeatures parameters
  std::vector<klr_iter_rec> recs;     // scratch of solve_reduced
  std::vector<klr_estimator*> folds;  // estimators of the last cross-validation / ensemble, one per fold (klr_estimator_fold)
  std::vector<int64_t> base_cols;     // a fold with its own co-preselection: the columns of the data its pair matrix was built from
  bool has_kmer = false;
  klr_kmer_features kf{};
  std::vector<int> kf_max_amb;
  std::string kf_alphabet;
};

static int fail(klr_estimator* e, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  e->err = buf;
  return code;
}

static int be_fail(klr_estimator* e, int code, const char* what) {
  const char* m = e->be.last_error ? e->be.last_error(e->be.ctx) : "";
  return fail(e, code, "%s failed: %s", what, m ? m : "");
}

static void say(const klr_estimator* e, int level, const char* fmt, ...) {   // PrintStderr, kmerLr_io.go:30-34
  if (e->cfg.verbose < level) return;
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
}

extern "C" void klr_config_default(klr_config* c) {
  memset(c, 0, sizeof(*c));
  c->lambda = NAN;
  c->epsilon_loss = 1e-8;       // kmerLr_learn.go:136
  c->eval_loss = 1;             // :331-333
}

extern "C" int klr_estimator_create(const klr_backend* be, const klr_config* cfg, klr_estimator** out) {
  if (!be || !cfg || !out) return KLR_EINVAL;
  if (!be->sweep_select || !be->select_columns || !be->cd_solve || !be->mat_shape || !be->mat_destroy ||
      !be->gradient || !be->class_weights)
    return KLR_EINVAL;
  klr_estimator* e = new klr_estimator();
  e->be = *be;
  e->cfg = *cfg;
  if (e->cfg.epsilon_loss != 0.0) e->cfg.eval_loss = 1;      // kmerLr_learn.go:331-333
  *out = e;
  return KLR_OK;
}

static void drop_folds(klr_estimator* e) {
  for (klr_estimator* f : e->folds) klr_estimator_destroy(f);
  e->folds.clear();
}

extern "C" void klr_estimator_destroy(klr_estimator* e) {
  if (!e) return;
  drop_folds(e);
  delete e;
}

extern "C" klr_estimator* klr_estimator_fold(const klr_estimator* e, int fold) {
  if (!e || fold < 0 || fold >= (int)e->folds.size()) return nullptr;
  return e->folds[(size_t)fold];
}

extern "C" int klr_estimator_base_cols(const klr_estimator* e, const int64_t** cols, int64_t* n) {
  if (!e) return KLR_EINVAL;
  if (cols) *cols = e->base_cols.empty() ? nullptr : e->base_cols.data();
  if (n) *n = (int64_t)e->base_cols.size();
  return KLR_OK;
}

// a fresh estimator for a fold or ensemble member: the configuration AND the starting model of `e`
// (NewKmerLrEnsembleEstimator(config, classifier, icv), kmerLr_estimator_ensemble.go:35-41)
static klr_estimator* fresh_like(klr_estimator* e) {
  klr_estimator* fe = nullptr;
  if (klr_estimator_create(&e->be, &e->cfg, &fe) != KLR_OK) return nullptr;
  fe->cols = e->cols; fe->theta = e->theta;
  fe->has_kmer = e->has_kmer; fe->kf = e->kf; fe->kf_max_amb = e->kf_max_amb; fe->kf_alphabet = e->kf_alphabet;
  return fe;
}

extern "C" const char* klr_estimator_error(const klr_estimator* e) { return e ? e->err.c_str() : "estimator is NULL"; }

extern "C" int klr_estimator_set_model(klr_estimator* e, const int64_t* cols, const double* theta, int64_t c1) {
  if (!e) return KLR_EINVAL;
  if (!cols || !theta || c1 < 1 || cols[0] != 0) return fail(e, KLR_EINVAL, "model must start with the bias column");
  for (int64_t p = 1; p < c1; p++) if (cols[p] <= cols[p - 1]) return fail(e, KLR_EINVAL, "model columns must ascend");
  e->cols.assign(cols, cols + c1);
  e->theta.assign(theta, theta + c1);
  return KLR_OK;
}

// selected transform = TransformFull.Select(b) (kmerLr_transform.go:290-306)
static void select_transform(const std::vector<int64_t>& cols, const double* scale, const double* offset,
                             std::vector<double>& sc, std::vector<double>& of) {
  sc.clear(); of.clear();
  if (scale) for (int64_t c : cols) sc.push_back(scale[c]);
  if (offset) for (int64_t c : cols) of.push_back(offset[c]);
}

// After a selection mask b: new column list and featureSelection.Theta() (:299-307): the restored
// nonzero coefficients on their columns, zero on the new ones.
static void adopt_selection(klr_estimator* e, const std::vector<uint8_t>& b) {
  std::vector<int64_t> nc;
  std::vector<double> nt;
  size_t q = 0;
  // the mask has a million entries at configs[3] and ~N set ones: skip the empty 8-byte words
  const int64_t nb = (int64_t)b.size();
  for (int64_t j0 = 0; j0 < nb; j0 += 8) {
    if (j0 + 8 <= nb) { uint64_t wd; memcpy(&wd, b.data() + j0, 8); if (wd == 0) continue; }
    const int64_t j1 = std::min(nb, j0 + 8);
    for (int64_t j = j0; j < j1; j++) {
      if (!b[j]) continue;
      while (q < e->cols.size() && e->cols[q] < j) q++;
      double t = 0.0;
      if (q < e->cols.size() && e->cols[q] == j) t = e->theta[q];
      nc.push_back(j); nt.push_back(t);
    }
  }
  e->cols.swap(nc); e->theta.swap(nt);
}

// KmerLrEstimator.estimate (kmerLr_estimator.go:147-173): solve on the reduced data, append the path
static int solve_reduced(klr_estimator* e, const klr_mat* red, int64_t n_total, double l1reg, std::vector<double>& theta) {
  // the hook records of one solve: kept in the estimator (an epoch allocates nothing), none asked for when nobody reads them
  const bool want = e->cfg.save_trace || e->cfg.verbose > 1;
  const int64_t cap = want ? (1 << 16) : 0;
  if (want && e->recs.size() < (size_t)cap) e->recs.resize((size_t)cap);
  std::vector<klr_iter_rec>& recs = e->recs;
  int64_t nrec = 0;
  int rc = e->be.cd_solve(e->be.ctx, red, theta.data(), e->cw, l1reg, e->cfg.l2reg, e->cfg.epsilon, e->cfg.epsilon_loss,
                          e->cfg.eval_loss, e->cfg.max_iterations, want ? recs.data() : nullptr, cap, &nrec);
  if (rc != KLR_OK) return be_fail(e, rc, "cd_solve");
  const int64_t keep = std::min(nrec, cap);
  if (e->cfg.save_trace || e->cfg.verbose > 1) {
    for (int64_t i = 0; i < keep; i++) {
      e->trace.push_back(recs[(size_t)i]);                                          // hook.go:74-76
      if (e->cfg.verbose > 1) {                                                     // hook.go:77-92
        const klr_iter_rec& r = recs[(size_t)i];
        fprintf(stdout, "{ iteration: %d, coefficients: %d, lambda: %e, ", r.iteration, r.nonzero - 1, r.lambda);
        if (e->cfg.eval_loss) fprintf(stdout, "loss: %f, ", r.loss);
        fprintf(stdout, "delta-theta: %e, time: %.6fs }\n", r.change, r.seconds);
      }
    }
  }
  (void)n_total;
  return KLR_OK;
}

static int estimate_common(klr_estimator* e, const klr_mat* data, const double* scale, const double* offset,
                           int64_t lambda_auto /* 0: fixed lambda */) {
  int64_t n = 0, dim = 0, nnz = 0, n_total = 0;
  if (e->be.mat_shape(data, &n, &dim, &nnz, &n_total) != KLR_OK) return fail(e, KLR_EINVAL, "bad data matrix");
  if (n_total == 0) return fail(e, KLR_EINVAL, "no training data given");        // kmerLr_learn.go:100-102
  for (int64_t c : e->cols) if (c >= dim) return fail(e, KLR_EINVAL, "model column %lld outside the data", (long long)c);
  const int64_t m = dim - 1;
  const int64_t N = lambda_auto;                                                    // n_params :91-101
  // SetLabels (kmerLr_estimator.go:219): class weights; Balance -> compute_class_weights, else [1,1]
  if (e->cfg.balance) {
    if (e->be.class_weights(e->be.ctx, data, e->cw) != KLR_OK) return be_fail(e, KLR_EHIP, "class_weights");
  } else { e->cw[0] = 1.0; e->cw[1] = 1.0; }

  if (e->be.solve_run_begin) e->be.solve_run_begin(e->be.ctx);
  klr_mat* red = nullptr;
  std::vector<double> sc, of;
  int rc = KLR_OK;
  bool have = false;
  auto rebuild = [&]() -> int {
    if (red) { e->be.mat_destroy(red); red = nullptr; }
    select_transform(e->cols, scale, offset, sc, of);
    int r = e->be.select_columns(e->be.ctx, data, e->cols.data(), (int64_t)e->cols.size(), sc.empty() ? nullptr : sc.data(),
                                 of.empty() ? nullptr : of.data(), &red);
    return r == KLR_OK ? KLR_OK : be_fail(e, r, "select_columns");
  };
  // warm start: theta lives on e->cols; its reduced rows are needed for X.theta
  if (e->cols.size() > 1) { if ((rc = rebuild()) != KLR_OK) return rc; }

  std::vector<uint8_t> b((size_t)dim);
  double prev_lambda = NAN;
  std::vector<int64_t> prev_cols;
  int stuck = 0;
  // -vvv: where the host's time goes (seconds in the backend's sweep_select / select_columns / reduced solve, and in between)
  double t_sweep = 0.0, t_sub = 0.0, t_solve = 0.0;
  const auto t_begin = std::chrono::steady_clock::now();
  auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  for (int64_t epoch = 0; e->cfg.max_epochs == 0 || epoch < e->cfg.max_epochs; epoch++) {   // :225
    // The reference's loop has no bound (kmerLr_estimator.go:225): on degenerate data -- a training set of one class, whose
    // intercept runs off -- it selects new features at lambda -> 0 for ever.  A library must not: a run that is still
    // going after a million epochs ends with an error (--max-epochs is the reference's own bound).
    if (epoch >= 1000000) { rc = fail(e, KLR_EINVAL, "the leapfrog loop did not terminate after %lld epochs (lambda = %g): degenerate training data? (--max-epochs bounds it)", (long long)epoch, e->l1reg / (double)n_total); break; }
    if (have && lambda_auto) {
      int64_t d = 0;
      for (size_t j = 1; j < e->theta.size(); j++) if (e->theta[j] != 0.0) d++;             // r.Nonzero()
      if (d > N) say(e, 1, "Estimated classifier has %lld non-zero coefficients, removing %lld features...\n", (long long)d, (long long)(d - N));
      else       say(e, 1, "Estimated classifier has %lld non-zero coefficients, selecting %lld new features...\n", (long long)d, (long long)(N - d));
    }
    double lambda = 0.0; int64_t c = 0; int ok = 0;
    if (lambda_auto) {
      // featureSelector.Select (kmerLr_estimator.go:235)
      const auto t0 = std::chrono::steady_clock::now();
      rc = e->be.sweep_select(e->be.ctx, data, red, e->cols.data(), e->theta.data(), (int64_t)e->cols.size(), e->cw, N,
                              e->l1reg, e->cfg.epsilon_lambda, scale, offset, b.data(), &lambda, &c, &ok, nullptr, nullptr);
      t_sweep += since(t0);
      if (rc != KLR_OK) { rc = be_fail(e, rc, "sweep_select"); break; }
    } else {
      // featureSelector.SelectFixed (kmerLr_feature_selection.go:136-159)
      std::vector<int64_t> ti; std::vector<double> tv;
      std::fill(b.begin(), b.end(), 0); b[0] = 1;
      for (size_t p = 0; p < e->cols.size(); p++) {
        if (p == 0 || e->theta[p] != 0.0) { ti.push_back(e->cols[p]); tv.push_back(e->theta[p]); }
        if (p > 0 && e->theta[p] != 0.0) { b[(size_t)e->cols[p]] = 1; c++; }                  // restoreNonzero
      }
      std::vector<double> g((size_t)dim);
      rc = e->be.gradient(e->be.ctx, data, ti.data(), tv.data(), (int64_t)ti.size(), e->cw, scale, offset, g.data());
      if (rc != KLR_OK) { rc = be_fail(e, rc, "gradient"); break; }
      lambda = e->cfg.lambda;
      for (int64_t k = 1; k < m + 1; k++) {                                                   // :146-155
        if (std::fabs(g[(size_t)k]) >= lambda && !b[(size_t)k]) { ok = 1; b[(size_t)k] = 1; c++; }
        if (e->cfg.max_features > 0 && c >= e->cfg.max_features && ok) break;
      }
    }
    if (!ok && have) break;                                                                   // :236-238
    e->l1reg = lambda * (double)n_total;                                                      // :239
    adopt_selection(e, b);                                                                    // :240-242
    { const auto t0 = std::chrono::steady_clock::now(); rc = rebuild(); t_sub += since(t0); }  // :244 selection.Data
    if (rc != KLR_OK) break;
    if (lambda_auto) say(e, 1, "Estimating parameters with lambda=%e...\n", lambda);          // :246
    else say(e, 1, "Estimating parameters with lambda=%e and %lld features...\n", lambda, (long long)(e->cols.size() - 1));
    { const auto t0 = std::chrono::steady_clock::now(); rc = solve_reduced(e, red, n_total, e->l1reg, e->theta); t_solve += since(t0); }   // :247
    if (rc != KLR_OK) break;
    if (e->cfg.save_path) {                                                                   // :168-170
      std::vector<int64_t> feat; std::vector<double> th;
      for (size_t p = 1; p < e->cols.size(); p++) { feat.push_back(e->cols[p] - 1); th.push_back(e->theta[p]); }
      e->path.append(e->l1reg / (double)n_total, feat, th);
    }
    int64_t nz = 0;
    for (size_t j = 1; j < e->theta.size(); j++) if (e->theta[j] != 0.0) nz++;
    e->ep_lambda.push_back(lambda); e->ep_nonzero.push_back(nz);
    // Epochs that end where the one before ended -- the same columns, lambda within 1e-12 of the last -- while the selection
    // still reports a change will go on for ever: a model with more non-zero coefficients than lambda_auto asks for (a
    // warm start from a larger model): nothing is ever removed, `c > N` keeps `ok` set (kmerLr_feature_selection.go:117-119)
    // and lambda creeps to a fixed point.  The reference spins there; three such epochs in a row end the run with an error.
    if (have && lambda_auto && nz > N && e->cols == prev_cols && std::fabs(lambda - prev_lambda) <= 1e-12 * std::fabs(lambda)) stuck++;
    else stuck = 0;
    if (stuck >= 3) {
      rc = fail(e, KLR_EINVAL, "the leapfrog loop makes no progress (lambda = %g, %lld non-zero coefficients for lambda_auto = %lld): "
                               "a model with more non-zero coefficients than features to select?", lambda, (long long)nz, (long long)N);
      break;
    }
    prev_lambda = lambda; prev_cols = e->cols;
    have = true;
  }
  say(e, 3, "leapfrog loop: %.3f ms in all, sweep + selection %.3f, column sub-select %.3f, reduced solve %.3f, the driver itself %.3f\n",
      since(t_begin) * 1e3, t_sweep * 1e3, t_sub * 1e3, t_solve * 1e3, (since(t_begin) - t_sweep - t_sub - t_solve) * 1e3);
  if (rc == KLR_OK && have) {
    Classifier cl;
    cl.cols = e->cols; cl.theta = e->theta; cl.lambda = e->l1reg / (double)n_total;
    select_transform(e->cols, scale, offset, cl.scale, cl.offset);
    if (e->cfg.penalty_free && lambda_auto) {                                                 // :249-252 reestimate
      say(e, 1, "Re-estimating parameters without penalty...\n");
      std::vector<double> t = e->theta;
      rc = solve_reduced(e, red, n_total, 0.0, t);
      if (rc == KLR_OK) cl.theta = t;
    }
    if (rc == KLR_OK) e->out.push_back(cl);
  } else if (rc == KLR_OK) {
    rc = fail(e, KLR_EINVAL, "no classifier was estimated (max_epochs = %lld)", (long long)e->cfg.max_epochs);
  }
  if (red) e->be.mat_destroy(red);
  if (e->be.solve_run_end) e->be.solve_run_end(e->be.ctx);
  return rc;
}

extern "C" int klr_estimator_estimate(klr_estimator* e, const klr_mat* data, const int64_t* lambda_auto,
                                      int64_t n_lambda_auto, const double* scale, const double* offset) {
  if (!e) return KLR_EINVAL;
  if (!data) return fail(e, KLR_EINVAL, "data is NULL");
  e->out.clear(); e->ep_lambda.clear(); e->ep_nonzero.clear();
  if (!std::isnan(e->cfg.lambda)) {                                                           // :258-261
    for (int64_t i = 0; i < n_lambda_auto; i++)
      if (lambda_auto[i] != 0) return fail(e, KLR_EINVAL, "options --lambda and --lambda-auto are incompatible");
    return estimate_common(e, data, scale, offset, 0);
  }
  if (n_lambda_auto <= 0 || !lambda_auto) return fail(e, KLR_EINVAL, "lambda_auto is empty");
  std::vector<int64_t> la(lambda_auto, lambda_auto + n_lambda_auto);
  std::sort(la.begin(), la.end());                                                            // kmerLr_learn.go:275
  for (int64_t N : la) {                                                                      // :263-267
    if (N <= 0) return fail(e, KLR_EINVAL, "lambda_auto must be positive");
    say(e, 1, "Estimating classifier with %lld non-zero coefficients...\n", (long long)N);
    int rc = estimate_common(e, data, scale, offset, N);
    if (rc != KLR_OK) return rc;
  }
  return KLR_OK;
}

extern "C" int64_t klr_estimator_n_classifiers(const klr_estimator* e) { return e ? (int64_t)e->out.size() : 0; }

extern "C" int klr_estimator_classifier(const klr_estimator* e, int64_t i, int64_t* c1, const int64_t** cols,
                                        const double** theta, double* lambda, const double** scale, const double** offset) {
  if (!e || i < 0 || i >= (int64_t)e->out.size()) return KLR_EINVAL;
  const Classifier& c = e->out[(size_t)i];
  if (c1) *c1 = (int64_t)c.cols.size();
  if (cols) *cols = c.cols.data();
  if (theta) *theta = c.theta.data();
  if (lambda) *lambda = c.lambda;
  if (scale) *scale = c.scale.empty() ? nullptr : c.scale.data();
  if (offset) *offset = c.offset.empty() ? nullptr : c.offset.data();
  return KLR_OK;
}

extern "C" int klr_estimator_trace(const klr_estimator* e, const klr_iter_rec** recs, int64_t* n) {
  if (!e) return KLR_EINVAL;
  if (recs) *recs = e->trace.data();
  if (n) *n = (int64_t)e->trace.size();
  return KLR_OK;
}

extern "C" int klr_estimator_epochs(const klr_estimator* e, const double** lambdas, const int64_t** nonzero, int64_t* n) {
  if (!e) return KLR_EINVAL;
  if (lambdas) *lambdas = e->ep_lambda.data();
  if (nonzero) *nonzero = e->ep_nonzero.data();
  if (n) *n = (int64_t)e->ep_lambda.size();
  return KLR_OK;
}

// ---- cross-validation ---------------------------------------------------------------------------------
static inline uint64_t sm64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// getCvGroups (kmerLr_crossvalidation.go:56-94): groups = i mod fold (-1 without folds), validation flags per block of
// `fold` rows, then the shuffle of :42-52 (swap a random remaining element to the end, both arrays alike) with Go's
// r.Intn(n) replaced by splitmix64(seed * GOLD + draw) mod n.
extern "C" void klr_cv_groups_val(int64_t n_all, int k_fold, double val_ratio, int64_t seed, int64_t row0, int64_t n_local,
                                  int32_t* groups, uint8_t* validation) {
  std::vector<int32_t> g((size_t)std::max<int64_t>(n_all, 1));
  std::vector<uint8_t> v((size_t)std::max<int64_t>(n_all, 1), 0);
  const int64_t fold = k_fold <= 1 ? 1 : k_fold;
  for (int64_t i = 0; i < n_all; i += fold)
    for (int64_t j = 0; j < fold && i + j < n_all; j++) {
      g[(size_t)(i + j)] = k_fold <= 1 ? -1 : (int32_t)j;
      v[(size_t)(i + j)] = ((double)i < val_ratio * (double)n_all && i + fold < n_all) ? 1 : 0;
    }
  if (seed != -1 && (k_fold > 1 || val_ratio > 0.0)) {
    uint64_t draw = 0;
    for (int64_t n = n_all; n > 0; n--) {
      const int64_t r = (int64_t)(sm64((uint64_t)seed * 0x9E3779B97F4A7C15ull + draw++) % (uint64_t)n);
      std::swap(g[(size_t)(n - 1)], g[(size_t)r]);
      std::swap(v[(size_t)(n - 1)], v[(size_t)r]);
    }
  }
  if (groups) for (int64_t i = 0; i < n_local; i++) groups[i] = g[(size_t)(row0 + i)];
  if (validation) for (int64_t i = 0; i < n_local; i++) validation[i] = v[(size_t)(row0 + i)];
}

extern "C" void klr_cv_groups(int64_t n_all, int k_fold, int64_t seed, int64_t row0, int64_t n_local, int32_t* groups) {
  klr_cv_groups_val(n_all, k_fold, 0.0, seed, row0, n_local, groups, nullptr);
}

// Worker contexts for jobs that run at once on one device (folds of a cross-validation, members of an ensemble): every
// worker has a forked context and a view of the data -- shared storage, its own row mask.  Fewer than two workers (the
// backend cannot fork, a context with a communicator): the caller runs its jobs in turn on e's own context.
struct Workers { std::vector<klr_ctx*> ctx; std::vector<klr_mat*> data; };
static int workers_open(klr_estimator* e, klr_mat* data, int T, Workers& W) {
  if (T < 2 || !(e->be.ctx_fork && e->be.ctx_release && e->be.mat_view)) return 0;
  for (int t = 0; t < T; t++) {
    klr_ctx* c = nullptr; klr_mat* v = nullptr;
    if (e->be.ctx_fork(e->be.ctx, &c) != KLR_OK || !c) break;             // e.g. a context with a communicator
    if (e->be.mat_view(c, data, &v) != KLR_OK || !v) { e->be.ctx_release(c); break; }
    W.ctx.push_back(c); W.data.push_back(v);
  }
  if (W.ctx.size() < 2) {
    for (size_t t = 0; t < W.ctx.size(); t++) { e->be.mat_destroy(W.data[t]); e->be.ctx_release(W.ctx[t]); }
    W.ctx.clear(); W.data.clear();
    return 0;
  }
  return (int)W.ctx.size();
}
static void workers_close(klr_estimator* e, Workers& W) {
  for (size_t t = 0; t < W.ctx.size(); t++) { e->be.mat_destroy(W.data[t]); e->be.ctx_release(W.ctx[t]); }
  W.ctx.clear(); W.data.clear();
}
// job(t, j): job j on worker t; false ends the run (the other workers finish the job they are on)
template <class F>
static void workers_run(const Workers& W, int njobs, F&& job) {
  std::atomic<int> next(0);
  std::atomic<bool> failed(false);
  auto worker = [&](int t) {
    for (;;) {
      const int j = next.fetch_add(1);
      if (j >= njobs || failed.load()) break;
      if (!job(t, j)) failed.store(true);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < (int)W.ctx.size(); t++) th.emplace_back(worker, t);
  worker(0);
  for (std::thread& x : th) x.join();
}

// what the folds of one cross-validation share (read-only while they run)
struct CvShared {
  int64_t n = 0, dim = 0, n_all = 0, row0 = 0, seed = 0, nla = 0;
  int k_fold = 0, nfold = 0, E = 1;
  bool val = false, copre = false;
  const int64_t* lambda_auto = nullptr;
  const double* scale = nullptr; const double* offset = nullptr;
  const std::vector<int32_t>* all = nullptr; const std::vector<uint8_t>* vall = nullptr;
  double* predictions = nullptr; double* loss_train = nullptr; double* loss_test = nullptr;
};

// One fold (crossvalidation's loop body, kmerLr_crossvalidation.go:176-195) on e's backend and `data`: the fold's estimator
// is appended to e->folds, its records to e->trace, the classifiers it keeps go to kept_out.  Folds write disjoint slots of
// predictions / loss_train / loss_test; with threads_cv every fold runs on an estimator, a context and a view of its own.
static int cv_fold(klr_estimator* e, klr_mat* data, const CvShared& S, int f, std::vector<Classifier>& kept_out) {
  const int64_t n = S.n, dim = S.dim, n_all = S.n_all, row0 = S.row0, seed = S.seed, nla = S.nla;
  const int nfold = S.nfold, E = S.E;
  const bool val = S.val, copre = S.copre;
  const int64_t* lambda_auto = S.lambda_auto;
  const double* scale = S.scale; const double* offset = S.offset;
  const std::vector<int32_t>& all = *S.all; const std::vector<uint8_t>& vall = *S.vall;
  const int32_t* grp = all.data() + row0;
  const uint8_t* vfl = vall.data() + row0;
  double* predictions = S.predictions; double* loss_train = S.loss_train; double* loss_test = S.loss_test;
  std::vector<uint8_t> mask((size_t)std::max<int64_t>(n, 1));
  std::vector<double> fsc, fof;                       // per-fold transform (validation sets)
  int rc = KLR_OK;
  // classifier.Loss on a row set ({Kmer,Scores}LrEnsemble.Loss, kmerLr_classifier_ensemble.go:108-123): Lambda = config.Lambda
  // (NaN with --lambda-auto: no penalty term), class weights [1,1] unless --balance (then those of the row set itself)
  auto set_loss = [&](klr_mat* red, const std::vector<uint8_t>& rows, int64_t n_rows, const std::vector<double>& theta, double* out) -> int {
    if (e->be.set_active_rows(e->be.ctx, red, rows.data(), n_rows) != KLR_OK) return be_fail(e, KLR_EHIP, "set_active_rows");
    double cwl[2] = {1.0, 1.0};
    if (e->cfg.balance && e->be.class_weights(e->be.ctx, red, cwl) != KLR_OK) return be_fail(e, KLR_EHIP, "class_weights");
    if (e->be.loss(e->be.ctx, red, theta.data(), cwl, e->cfg.lambda, out) != KLR_OK) return be_fail(e, KLR_EHIP, "loss");
    return KLR_OK;
  };
  // {Kmer,Scores}LrEnsemble.Summarize (kmerLr_classifier_ensemble.go:64-104)
  auto summarize = [&](const std::vector<double>& x) -> double {
    if (x.empty()) return NAN;
    if (x.size() == 1 && E == 1) return x[0];
    double r;
    switch (e->cfg.ensemble_summary) {
      case KLR_SUMMARY_PRODUCT: r = 1.0; for (double v : x) r *= v; return r;
      case KLR_SUMMARY_MIN: r = x[0]; for (double v : x) if (r > v) r = v; return r;
      case KLR_SUMMARY_MAX: r = x[0]; for (double v : x) if (r < v) r = v; return r;
      default: r = 0.0; for (double v : x) r += v; return r / (double)x.size();
    }
  };
  klr_mat* fold_ex = nullptr;          // the pair matrix of the fold (co-preselection inside the fold)
  do {
    const int32_t ff = f;             // without folds every group id is -1: no row is a test row (crossvalidation :183-186)
    // the three row sets of fold f over the WHOLE data set (filterCvGroup :96-120): test, validation, training
    int64_t n_test = 0, n_val = 0;
    for (int64_t i = 0; i < n_all; i++) { if (all[(size_t)i] == ff) n_test++; else if (val && vall[(size_t)i]) n_val++; }
    const int64_t n_train = n_all - n_test - n_val;
    auto is_test = [&](int64_t i) { return grp[i] == ff; };
    auto is_val = [&](int64_t i) { return !is_test(i) && val && vfl[i]; };
    const double* sc = scale; const double* of = offset;
    if (val && e->cfg.data_transform != 0 && e->be.transform_fit) {
      // transform.Fit(append(data_train.Data, data_test.Data...)) (kmerLr_estimator_ensemble.go:116-117): every row but the validation set
      for (int64_t i = 0; i < n; i++) mask[(size_t)i] = !is_val(i);
      if (e->be.set_active_rows(e->be.ctx, data, mask.data(), n_train + n_test) != KLR_OK) { rc = be_fail(e, KLR_EHIP, "set_active_rows"); break; }
      fsc.assign((size_t)dim, 0.0); fof.assign((size_t)dim, 0.0);
      int has_off = 0;
      if (e->be.transform_fit(e->be.ctx, data, e->cfg.data_transform, fsc.data(), fof.data(), &has_off) != KLR_OK) { rc = be_fail(e, KLR_EHIP, "transform_fit"); break; }
      sc = fsc.data(); of = has_off ? fof.data() : nullptr;
    }
    // the fold's estimator (learn_parameters, kmerLr_learn.go:37-38): estimates itself without an ensemble, else holds
    // the members (NewKmerLrEnsembleEstimator: ensemble_size estimators that start from the same model)
    klr_estimator* fe = fresh_like(e);
    if (!fe) { rc = fail(e, KLR_EINVAL, "cannot create fold estimator"); break; }
    e->folds.push_back(fe);
    // Co-preselection inside the fold ({Kmer,Scores}LrEstimatorEnsemble.Estimate, kmerLr_estimator_ensemble.go:99-115): `data`
    // is the BASE matrix; on the fold's training rows a first leapfrog run without co-occurrence picks cfg.copreselection
    // features (transform: the one fitted above on the training + test rows, or the caller's), the base data are restricted
    // to them and the pair matrix of that restriction takes the place of the data for the rest of the fold; its own
    // transform is fitted on its training + test rows (:116-117).
    klr_mat* data_f = data;
    int64_t dim_f = dim;
    std::vector<double> xsc, xof;
    if (copre) {
      for (int64_t i = 0; i < n; i++) mask[(size_t)i] = !is_test(i) && !is_val(i);
      if (e->be.set_active_rows(e->be.ctx, data, mask.data(), n_train) != KLR_OK) { rc = be_fail(e, KLR_EHIP, "set_active_rows"); break; }
      fe->base_cols.assign((size_t)dim, 0);
      int64_t nb1 = 0;
      char eb[512] = "";
      klr_config c1 = e->cfg; c1.verbose = 0;
      rc = klr_cooccurrence_prepare(&e->be, &c1, data, sc, of, &fold_ex, fe->base_cols.data(), &nb1, eb, sizeof(eb));
      if (rc != KLR_OK) { fail(e, rc, "co-preselection of fold %d failed: %s", f + 1, eb); break; }
      fe->base_cols.resize((size_t)nb1);
      data_f = fold_ex;
      if (e->be.mat_shape(data_f, nullptr, &dim_f, nullptr, nullptr) != KLR_OK) { rc = fail(e, KLR_EINVAL, "bad pair matrix"); break; }
      sc = nullptr; of = nullptr;
      if (e->cfg.data_transform != 0) {
        if (!e->be.transform_fit) { rc = fail(e, KLR_EINVAL, "co-preselection per fold with a data transform needs a backend that can fit it"); break; }
        for (int64_t i = 0; i < n; i++) mask[(size_t)i] = !is_val(i);
        if (e->be.set_active_rows(e->be.ctx, data_f, mask.data(), n_train + n_test) != KLR_OK) { rc = be_fail(e, KLR_EHIP, "set_active_rows"); break; }
        xsc.assign((size_t)dim_f, 0.0); xof.assign((size_t)dim_f, 0.0);
        int has_off = 0;
        if (e->be.transform_fit(e->be.ctx, data_f, e->cfg.data_transform, xsc.data(), xof.data(), &has_off) != KLR_OK) { rc = be_fail(e, KLR_EHIP, "transform_fit"); break; }
        sc = xsc.data(); of = has_off ? xof.data() : nullptr;
      }
    }
    // member groups (estimate_ensemble :75): getCvGroups over the TRAINING rows in data order; member k trains on the
    // training rows whose group is not k.  A local row's position among all training rows comes from the global arrays.
    std::vector<int32_t> wgrp;
    std::vector<int64_t> tpos;
    if (E > 1) {
      if (n_train < E) { rc = fail(e, KLR_EINVAL, "not enough training samples (%lld) for an ensemble of %d", (long long)n_train, E); break; }
      wgrp.resize((size_t)n_train);
      klr_cv_groups(n_train, E, seed, 0, n_train, wgrp.data());
      tpos.assign((size_t)std::max<int64_t>(n, 1), -1);
      int64_t cnt = 0;
      for (int64_t gi = 0; gi < row0 + n; gi++) {
        const bool tr = all[(size_t)gi] != ff && !(val && vall[(size_t)gi]);
        if (gi >= row0 && tr) tpos[(size_t)(gi - row0)] = cnt;
        if (tr) cnt++;
      }
    }
    std::vector<klr_estimator*> members;
    for (int k = 0; k < E && rc == KLR_OK; k++) {
      int64_t n_k = n_train;
      if (E > 1) { n_k = 0; for (int32_t g : wgrp) if (g != k) n_k++; }
      for (int64_t i = 0; i < n; i++) mask[(size_t)i] = !is_test(i) && !is_val(i) && (E == 1 || wgrp[(size_t)tpos[(size_t)i]] != k);
      if (e->be.set_active_rows(e->be.ctx, data_f, mask.data(), n_k) != KLR_OK) { rc = be_fail(e, KLR_EHIP, "set_active_rows"); break; }
      klr_estimator* me = fe;
      if (E > 1) {
        me = fresh_like(e);
        if (!me) { rc = fail(e, KLR_EINVAL, "cannot create member estimator"); break; }
        fe->folds.push_back(me);
      }
      members.push_back(me);
      rc = klr_estimator_estimate(me, data_f, lambda_auto, nla, sc, of);
      if (rc != KLR_OK) { e->err = me->err; break; }
      for (auto& r : me->trace) { e->trace.push_back(r); if (E > 1) fe->trace.push_back(r); }
    }
    if (rc != KLR_OK) break;
    const size_t ncls = members[0]->out.size();                        // lambda_auto values (1 with a fixed lambda)
    // reduced data of member k's classifier j (SelectData)
    std::vector<klr_mat*> reds((size_t)E * ncls, nullptr);
    for (int k = 0; k < E && rc == KLR_OK; k++)
      for (size_t j = 0; j < ncls && rc == KLR_OK; j++) {
        const Classifier& c = members[(size_t)k]->out[j];
        if (e->be.select_columns(e->be.ctx, data_f, c.cols.data(), (int64_t)c.cols.size(), c.scale.empty() ? nullptr : c.scale.data(),
                                 c.offset.empty() ? nullptr : c.offset.data(), &reds[(size_t)k * ncls + j]) != KLR_OK) rc = be_fail(e, KLR_EHIP, "select_columns");
      }
    // the ensemble's loss on a row set: the summary of its members' losses (KmerLrEnsemble.Loss :108-123)
    auto ens_loss = [&](size_t j, const std::vector<uint8_t>& rows, int64_t n_rows, double* out) -> int {
      std::vector<double> lv((size_t)E, 0.0);
      for (int k = 0; k < E; k++) {
        const int r_ = set_loss(reds[(size_t)k * ncls + j], rows, n_rows, members[(size_t)k]->out[j].theta, &lv[(size_t)k]);
        if (r_ != KLR_OK) return r_;
      }
      *out = summarize(lv);
      return KLR_OK;
    };
    size_t j_first = 0, j_end = ncls;
    if (rc == KLR_OK && val) {
      // "if validation data is available, select best classifier" (kmerLr_estimator_ensemble.go:119-135)
      size_t best = 0; double vbest = INFINITY;
      for (int64_t i = 0; i < n; i++) mask[(size_t)i] = is_val(i);
      for (size_t j = 0; j < ncls && rc == KLR_OK; j++) {
        double v = 0.0;
        if (n_val > 0) rc = ens_loss(j, mask, n_val, &v);
        if (rc != KLR_OK) break;
        if (v < vbest) { vbest = v; best = j; }
        say(e, 1, "> Classifier %zu has loss %g...\n", j, v);
      }
      say(e, 1, "> Selecting classifier %zu\n", best);
      j_first = best; j_end = best + 1;
    }
    std::vector<double> lpk((size_t)E * (size_t)std::max<int64_t>(n, 1));
    for (size_t j = j_first; j < j_end && rc == KLR_OK; j++) {
      const size_t jo = val ? 0 : j;                                   // output row
      // loss_train on the training rows (classifier.Loss, kmerLr_estimator_ensemble.go:143-145)
      for (int64_t i = 0; i < n; i++) mask[(size_t)i] = !is_test(i) && !is_val(i);
      rc = ens_loss(j, mask, n_train, &loss_train[jo * nfold + f]);
      // predictions of the test rows: the summary of the members' log pdf per row (KmerLrEnsemble.Predict :124-137)
      if (rc == KLR_OK && n_test > 0) {
        for (int k = 0; k < E && rc == KLR_OK; k++)
          if (e->be.log_pdf(e->be.ctx, reds[(size_t)k * ncls + j], members[(size_t)k]->out[j].theta.data(), lpk.data() + (size_t)k * (size_t)n) != KLR_OK)
            rc = be_fail(e, KLR_EHIP, "log_pdf");
        std::vector<double> t((size_t)E);
        if (rc == KLR_OK) for (int64_t i = 0; i < n; i++) if (is_test(i)) {
          for (int k = 0; k < E; k++) t[(size_t)k] = lpk[(size_t)k * (size_t)n + (size_t)i];
          predictions[jo * n + i] = summarize(t);
        }
      }
      // loss_test on the test rows (no rows: 0, kmerLr_logistic_regression.go:251-253)
      if (rc == KLR_OK) {
        loss_test[jo * nfold + f] = 0.0;
        if (n_test > 0) {
          for (int64_t i = 0; i < n; i++) mask[(size_t)i] = is_test(i);
          rc = ens_loss(j, mask, n_test, &loss_test[jo * nfold + f]);
        }
      }
    }
    for (klr_mat* r : reds) if (r) e->be.mat_destroy(r);
    if (fold_ex) { e->be.mat_destroy(fold_ex); fold_ex = nullptr; }
    if (rc != KLR_OK) break;
    // what the fold keeps: member-major, kept classifier jo of member k at k * n_kept + jo
    {
      std::vector<Classifier> kept;
      for (int k = 0; k < E; k++) for (size_t j = j_first; j < j_end; j++) kept.push_back(members[(size_t)k]->out[j]);
      fe->out = kept;
      kept_out = kept;
    }
  } while (0);
  if (fold_ex) e->be.mat_destroy(fold_ex);
  return rc;
}

extern "C" int klr_estimator_crossvalidate(klr_estimator* e, klr_mat* data, int64_t row0, int64_t n_all, int k_fold,
                                           int64_t seed, const int64_t* lambda_auto, int64_t nla, const double* scale,
                                           const double* offset, double* predictions, int32_t* groups_out,
                                           double* loss_train, double* loss_test) {
  if (!e) return KLR_EINVAL;
  const double vr = e->cfg.validation_size;
  const bool val = vr > 0.0;
  if (!data || k_fold < 1 || (k_fold < 2 && !val) || !lambda_auto || nla < 1 || !loss_train || !loss_test || (k_fold > 1 && !predictions))
    return fail(e, KLR_EINVAL, "crossvalidate: invalid argument");
  if (!e->be.set_active_rows || !e->be.log_pdf) return fail(e, KLR_EINVAL, "backend lacks set_active_rows / log_pdf");
  int64_t n = 0, dim = 0;
  if (e->be.mat_shape(data, &n, &dim, nullptr, nullptr) != KLR_OK) return fail(e, KLR_EINVAL, "bad data matrix");
  if (n_all < k_fold) return fail(e, KLR_EINVAL, "not enough training samples (%lld) for %d-fold cross-validation", (long long)n_all, k_fold);
  if (val && n_all < 2 * (int64_t)k_fold)
    return fail(e, KLR_EINVAL, "not enough training samples for creating validation set with %d-fold cross-validation", k_fold);
  std::vector<int32_t> all((size_t)n_all);
  std::vector<uint8_t> vall((size_t)n_all);
  klr_cv_groups_val(n_all, k_fold, vr, seed, 0, n_all, all.data(), vall.data());
  const int32_t* grp = all.data() + row0;
  if (groups_out) for (int64_t i = 0; i < n; i++) groups_out[i] = grp[i];
  int rc = KLR_OK;
  e->out.clear();
  drop_folds(e);
  std::vector<Classifier> all_cls;
  const int nfold = k_fold <= 1 ? 1 : k_fold;
  CvShared S;
  S.n = n; S.dim = dim; S.n_all = n_all; S.row0 = row0; S.seed = seed; S.nla = nla;
  S.k_fold = k_fold; S.nfold = nfold; S.E = std::max(1, e->cfg.ensemble_size);
  S.val = val; S.copre = e->cfg.cooccurrence && e->cfg.copreselection > 0 && e->cfg.copreselection < dim - 1;
  S.lambda_auto = lambda_auto; S.scale = scale; S.offset = offset; S.all = &all; S.vall = &vall;
  S.predictions = predictions; S.loss_train = loss_train; S.loss_test = loss_test;
  std::vector<std::vector<Classifier>> kept((size_t)nfold);
  // Folds at once (cfg.threads_cv, the reference's --threads-cv): every worker thread forks a context and takes a view of
  // the data (shared storage, its own row mask), a fold runs on an estimator of its own; the results are put together in
  // fold order afterwards, so nothing depends on the number of threads.  One rank only: the folds of several ranks would
  // have to post their collectives in one agreed order.
  Workers W;
  const int T = workers_open(e, data, std::min(std::max(1, e->cfg.threads_cv), nfold), W);
  if (T == 0 && e->cfg.threads_cv > 1 && nfold > 1) say(e, 1, "Cross-validation folds run one after the other (this context is not forked)\n");
  if (T == 0) {
    for (int f = 0; f < nfold && rc == KLR_OK; f++) rc = cv_fold(e, data, S, f, kept[(size_t)f]);
    e->be.set_active_rows(e->be.ctx, data, nullptr, 0);
  } else {
    std::vector<klr_estimator*> west((size_t)nfold, nullptr);
    std::vector<int> wrc((size_t)nfold, KLR_OK);
    workers_run(W, nfold, [&](int t, int f) {
      klr_estimator* w = fresh_like(e);
      if (!w) { wrc[(size_t)f] = KLR_EINVAL; return false; }
      w->be.ctx = W.ctx[(size_t)t];
      west[(size_t)f] = w;
      wrc[(size_t)f] = cv_fold(w, W.data[(size_t)t], S, f, kept[(size_t)f]);
      return wrc[(size_t)f] == KLR_OK;
    });
    // in fold order: the fold estimators (handed to e, on e's backend from now on), the records, the first error
    for (int f = 0; f < nfold; f++) {
      klr_estimator* w = west[(size_t)f];
      if (rc == KLR_OK && wrc[(size_t)f] != KLR_OK) { rc = wrc[(size_t)f]; e->err = w ? w->err : std::string("cannot create fold estimator"); }
      if (!w) { if (rc == KLR_OK) rc = fail(e, KLR_EINVAL, "fold %d did not run", f + 1); continue; }
      if (rc == KLR_OK) {
        for (klr_estimator* fe : w->folds) {
          fe->be = e->be;
          for (klr_estimator* me : fe->folds) me->be = e->be;
          e->folds.push_back(fe);
        }
        w->folds.clear();
        for (auto& r : w->trace) e->trace.push_back(r);
      }
      klr_estimator_destroy(w);
    }
    workers_close(e, W);
  }
  if (rc == KLR_OK) for (auto& k : kept) for (const Classifier& c : k) all_cls.push_back(c);
  if (rc == KLR_OK) e->out.swap(all_cls);        // fold-major: classifier of fold f, lambda_auto j at f*nla + j (f alone with a validation
                                                 // set); with an ensemble the folds' member-major lists one after the other
  return rc;
}

extern "C" int klr_save_crossvalidation(const char* filename, const double* predictions, const uint8_t* labels,
                                        const int32_t* groups, int64_t n, int k_fold) {
  FILE* f = fopen(filename, "w");
  if (!f) return KLR_EINVAL;
  fprintf(f, "%15s\t%6s\n", "prediction", "labels");                                  // crossvalidation.go:133
  for (int g = 0; g < k_fold; g++)                                                        // folds joined in order :196-207
    for (int64_t i = 0; i < n; i++)
      if (groups[i] == g) fprintf(f, "%15e\t%6d\n", predictions[i], labels[i] ? 1 : 0);
  fclose(f);
  return KLR_OK;
}

extern "C" int klr_save_crossvalidation_loss(const char* filename, const double* loss_train, const double* loss_test, int k_fold) {
  FILE* f = fopen(filename, "w");
  if (!f) return KLR_EINVAL;
  fprintf(f, "%15s\t%15s\n", "train", "test");                                         // :158
  for (int g = 0; g < k_fold; g++) fprintf(f, "%15e\t%15e\n", loss_train[g], loss_test[g]);
  fclose(f);
  return KLR_OK;
}

// format_duration, kmerLr_estimator_trace.go:29-36
static std::string format_duration(double seconds) {
  const double hours_total = seconds / 3600.0;
  const int days = (int)(hours_total / 24.0);
  const int hours = (int)std::fmod(hours_total, 24.0);
  const int minutes = (int)std::fmod(seconds / 60.0, 60.0);
  const int secs = (int)std::fmod(seconds, 60.0);
  const int msecs = (int)std::fmod(std::floor(seconds * 1000.0), 1000.0);
  char buf[64];
  snprintf(buf, sizeof(buf), "%02d:%02d:%02d:%02d.%03d", days, hours, minutes, secs, msecs);
  return buf;
}

extern "C" int klr_estimator_save_trace(const klr_estimator* e, const char* filename) {
  if (!e || !filename) return KLR_EINVAL;
  FILE* f = fopen(filename, "w");
  if (!f) return KLR_EINVAL;
  bool any_loss = false;
  for (const auto& r : e->trace) if (!std::isnan(r.loss)) any_loss = true;
  const bool any_lambda = !e->trace.empty();
  fprintf(f, "%15s %9s %12s %8s", "duration", "iteration", "change", "nonzero");       // trace.go:61
  if (any_lambda) fprintf(f, " %12s", "lambda");
  if (any_loss) fprintf(f, " %12s", "loss");
  fprintf(f, "\n");
  for (const auto& r : e->trace) {                                                       // :69-77
    fprintf(f, "%15s %9d %12e %8d", format_duration(r.seconds).c_str(), r.iteration, r.change, r.nonzero);
    if (any_lambda) fprintf(f, " %12e", r.lambda);
    if (any_loss) fprintf(f, " %12e", r.loss);
    fprintf(f, "\n");
  }
  fclose(f);
  return KLR_OK;
}

extern "C" int klr_estimator_save_path(const klr_estimator* e, const char* filename) {
  if (!e || !filename) return KLR_EINVAL;
  FILE* f = fopen(filename, "w");
  if (!f) return KLR_EINVAL;
  fprintf(f, "%13s %13s %s\n", "lambda", "norm", "theta");                              // path.go:55
  for (size_t i = 0; i < e->path.lambda.size(); i++) {                                   // :58-71
    fprintf(f, "%13e %13e", e->path.lambda[i], e->path.norm[i]);
    for (size_t j = 0; j < e->path.theta[i].size(); j++) fprintf(f, j == 0 ? " %e" : ",%e", e->path.theta[i][j]);
    fprintf(f, "\n");
  }
  fclose(f);
  return KLR_OK;
}

static void json_str(FILE* f, const char* s) {
  fputc('"', f);
  for (; *s; s++) { if (*s == '"' || *s == '\\') fputc('\\', f); fputc(*s, f); }
  fputc('"', f);
}

extern "C" int klr_estimator_set_kmer_features(klr_estimator* e, const klr_kmer_features* kf) {
  if (!e || !kf) return KLR_EINVAL;
  e->kf = *kf;
  e->kf_max_amb.assign(kf->max_ambiguous ? kf->max_ambiguous : nullptr, kf->max_ambiguous ? kf->max_ambiguous + kf->n_max_ambiguous : nullptr);
  e->kf_alphabet = kf->alphabet ? kf->alphabet : "nucleotide alphabet";
  e->kf.max_ambiguous = nullptr; e->kf.alphabet = nullptr;
  e->has_kmer = true;
  return KLR_OK;
}

static bool all_finite(const std::vector<double>& v) {
  for (double x : v) if (!std::isfinite(x)) return false;
  return true;
}

// "Name" and "Parameters" of a model file.  index = 0-based features of the data the model uses (ascending), features =
// pairs of positions in index (equal for a single feature), names = one string per feature of the DATA (nullable).
// ConfigDistribution{Name, Parameters, Distributions} as written by ExportDistribution (kmerLr_io.go:105-112): field
// order of Parameters as the exporting Go structs declare them (encoding/json keeps declaration order) --
// kmerLr_features.go:222-235 for kmerLr, scoresLr_feature.go:72-80 for scoresLr.  The envelope itself comes from the
// absent autodiff module: its exact spelling is unpinned.
static void write_head(FILE* f, const klr_estimator* e, const std::string& name, bool cooccurrence, const std::vector<int64_t>& index,
                       const std::vector<std::pair<size_t, size_t>>& features, const char* const* names) {
  const bool kmer = e->has_kmer && name.compare(0, 6, "kmerLr") == 0;
  fprintf(f, "{\n  \"Name\": ");
  json_str(f, name.c_str());
  fprintf(f, ",\n  \"Parameters\": {\n");
  auto put_features = [&]() {
    fprintf(f, "    \"Features\": [");
    for (size_t k = 0; k < features.size(); k++) fprintf(f, "%s[%zu,%zu]", k ? "," : "", features[k].first, features[k].second);
    fprintf(f, "]");
  };
  auto put_names = [&](const char* key) {
    fprintf(f, "    \"%s\": [", key);
    if (names) for (size_t k = 0; k < index.size(); k++) { if (k) fputc(',', f); json_str(f, names[index[k]]); }
    fprintf(f, "]");
  };
  if (kmer) {
    const klr_kmer_features& kf = e->kf;
    fprintf(f, "    \"M\": %d,\n    \"N\": %d,\n    \"Binarize\": %s,\n    \"Complement\": %s,\n    \"Cooccurrence\": %s,\n"
               "    \"Reverse\": %s,\n    \"Revcomp\": %s,\n    \"MaxAmbiguous\": ",
            kf.M, kf.N, kf.binarize ? "true" : "false", kf.complement ? "true" : "false", cooccurrence ? "true" : "false",
            kf.reverse ? "true" : "false", kf.revcomp ? "true" : "false");
    if (e->kf_max_amb.empty()) fprintf(f, "null");
    else { fputc('[', f); for (size_t k = 0; k < e->kf_max_amb.size(); k++) fprintf(f, "%s%d", k ? "," : "", e->kf_max_amb[k]); fputc(']', f); }
    fprintf(f, ",\n    \"Alphabet\": ");
    json_str(f, e->kf_alphabet.c_str());
    fprintf(f, ",\n");
    put_names("Kmers");
    fprintf(f, ",\n");
    put_features();
    fprintf(f, "\n  },\n");
    return;
  }
  fprintf(f, "    \"Cooccurrence\": %s,\n", cooccurrence ? "true" : "false");
  put_features();
  fprintf(f, ",\n    \"Index\": [");
  for (size_t k = 0; k < index.size(); k++) fprintf(f, "%s%lld", k ? "," : "", (long long)index[k]);
  fprintf(f, "],\n");
  put_names("Names");
  fprintf(f, "\n  },\n");
}

static void write_lr(FILE* f, const std::vector<double>& theta, bool last_comma) {
  fprintf(f, "    {\"Name\": \"vector:logistic regression\", \"Parameters\": [");
  for (size_t k = 0; k < theta.size(); k++) fprintf(f, "%s%.17g", k ? "," : "", theta[k]);
  fprintf(f, "], \"Distributions\": null}%s\n", last_comma ? "," : "");
}

static void write_transform_and_close(FILE* f, const std::vector<double>& offset, const std::vector<double>& scale) {
  fprintf(f, "    {\"Name\": \"transform\", \"Parameters\": {\"Offset\": [");
  for (size_t k = 0; k < offset.size(); k++) fprintf(f, "%s%.17g", k ? "," : "", offset[k]);
  fprintf(f, "], \"Scale\": [");
  for (size_t k = 0; k < scale.size(); k++) fprintf(f, "%s%.17g", k ? "," : "", scale[k]);
  fprintf(f, "]}, \"Distributions\": null}\n  ]\n}\n");
}

extern "C" int klr_estimator_save_model(const klr_estimator* e, int64_t i, const char* filename, const char* type_name,
                                        const char* const* names) {
  if (!e || !filename || i < 0 || i >= (int64_t)e->out.size()) return KLR_EINVAL;
  const Classifier& c = e->out[(size_t)i];
  if (!all_finite(c.theta) || !all_finite(c.scale) || !all_finite(c.offset)) return KLR_EINVAL;   // NaN is not JSON
  FILE* f = fopen(filename, "w");
  if (!f) return KLR_EINVAL;
  const size_t nf = c.cols.size() - 1;
  std::vector<int64_t> index(nf);
  std::vector<std::pair<size_t, size_t>> features(nf);
  for (size_t k = 0; k < nf; k++) { index[k] = c.cols[k + 1] - 1; features[k] = {k, k}; }       // non-co-occurrence: [k,k]
  write_head(f, e, type_name ? type_name : "scoresLr", false, index, features, names);
  fprintf(f, "  \"Distributions\": [\n");
  write_lr(f, c.theta, true);
  write_transform_and_close(f, c.offset, c.scale);
  fclose(f);
  return KLR_OK;
}

// ---------------------------------------------------------------------------------------------------
// co-occurrence mode
// ---------------------------------------------------------------------------------------------------
extern "C" int klr_cooccurrence_prepare(const klr_backend* be, const klr_config* cfg, const klr_mat* data, const double* pre_scale,
                                        const double* pre_offset, klr_mat** expanded, int64_t* base_cols, int64_t* n_base_cols,
                                        char* errbuf, int64_t errcap) {
  auto err = [&](const char* what, const char* detail) {
    if (errbuf && errcap > 0) snprintf(errbuf, (size_t)errcap, "%s%s%s", what, detail && *detail ? ": " : "", detail ? detail : "");
  };
  if (!be || !cfg || !data || !expanded || !base_cols || !n_base_cols) return KLR_EINVAL;
  *expanded = nullptr;
  if (!be->expand_cooccurrence) { err("this backend has no co-occurrence expansion", ""); return KLR_EINVAL; }
  int64_t n = 0, dim = 0;
  if (be->mat_shape(data, &n, &dim, nullptr, nullptr) != KLR_OK) { err("mat_shape failed", ""); return KLR_EINVAL; }
  const klr_mat* base = data;
  klr_mat* owned = nullptr;
  int64_t nb1 = dim;
  for (int64_t j = 0; j < dim; j++) base_cols[j] = j;
  if (cfg->copreselection > 0 && cfg->copreselection < dim - 1) {
    // estimate_loop(config, data_train, transform, config.Copreselection, false), then SelectData and Reset
    klr_config c2 = *cfg;
    c2.lambda = NAN; c2.cooccurrence = 0; c2.copreselection = 0; c2.save_trace = 0; c2.save_path = 0;
    klr_estimator* e = nullptr;
    int rc = klr_estimator_create(be, &c2, &e);
    if (rc != KLR_OK) { err("co-preselection: estimator_create failed", ""); return rc; }
    const int64_t q = cfg->copreselection;
    rc = klr_estimator_estimate(e, data, &q, 1, pre_scale, pre_offset);
    if (rc != KLR_OK) { err("co-preselection failed", klr_estimator_error(e)); klr_estimator_destroy(e); return rc; }
    int64_t c1 = 0; const int64_t* cols = nullptr;
    klr_estimator_classifier(e, 0, &c1, &cols, nullptr, nullptr, nullptr, nullptr);
    for (int64_t k = 0; k < c1; k++) base_cols[k] = cols[k];
    nb1 = c1;
    klr_estimator_destroy(e);
    if ((rc = be->select_columns(be->ctx, data, base_cols, nb1, nullptr, nullptr, &owned)) != KLR_OK) {
      err("co-preselection: select_columns failed", be->last_error ? be->last_error(be->ctx) : "");
      return rc;
    }
    base = owned;
  }
  *n_base_cols = nb1;
  const int rc = be->expand_cooccurrence(be->ctx, base, expanded);
  if (rc != KLR_OK) err("co-occurrence expansion failed", be->last_error ? be->last_error(be->ctx) : "");
  if (owned) be->mat_destroy(owned);
  return rc;
}

extern "C" int klr_estimator_save_model_pairs(const klr_estimator* e, int64_t i, const char* filename, const char* type_name,
                                              const char* const* names, const int64_t* pairs) {
  if (!e || !filename || !pairs || i < 0 || i >= (int64_t)e->out.size()) return KLR_EINVAL;
  const Classifier& c = e->out[(size_t)i];
  const size_t nf = c.cols.size() - 1;
  std::vector<int64_t> index(pairs, pairs + 2 * nf);
  std::sort(index.begin(), index.end());
  index.erase(std::unique(index.begin(), index.end()), index.end());
  auto pos = [&](int64_t f) { return (size_t)(std::lower_bound(index.begin(), index.end(), f) - index.begin()); };
  if (!all_finite(c.theta) || !all_finite(c.scale) || !all_finite(c.offset)) return KLR_EINVAL;
  FILE* f = fopen(filename, "w");
  if (!f) return KLR_EINVAL;
  std::vector<std::pair<size_t, size_t>> features(nf);
  for (size_t k = 0; k < nf; k++) features[k] = {pos(pairs[2 * k]), pos(pairs[2 * k + 1])};
  write_head(f, e, type_name ? type_name : "scoresLr", true, index, features, names);
  fprintf(f, "  \"Distributions\": [\n");
  write_lr(f, c.theta, true);
  write_transform_and_close(f, c.offset, c.scale);
  fclose(f);
  return KLR_OK;
}

// ---------------------------------------------------------------------------------------------------
// ensembles
// ---------------------------------------------------------------------------------------------------
extern "C" int klr_estimator_estimate_ensemble(klr_estimator* e, klr_mat* data, int64_t row0, int64_t n_all, int ensemble_size,
                                               int64_t seed, const int64_t* lambda_auto, int64_t nla, const double* scale,
                                               const double* offset) {
  if (!e) return KLR_EINVAL;
  if (!data || ensemble_size < 2 || !lambda_auto || nla < 1) return fail(e, KLR_EINVAL, "estimate_ensemble: invalid argument");
  if (!e->be.set_active_rows) return fail(e, KLR_EINVAL, "backend lacks set_active_rows");
  int64_t n = 0, dim = 0;
  if (e->be.mat_shape(data, &n, &dim, nullptr, nullptr) != KLR_OK) return fail(e, KLR_EINVAL, "bad data matrix");
  if (n_all < ensemble_size) return fail(e, KLR_EINVAL, "not enough training samples (%lld) for an ensemble of %d", (long long)n_all, ensemble_size);
  std::vector<int32_t> all((size_t)n_all);
  klr_cv_groups(n_all, ensemble_size, seed, 0, n_all, all.data());                // getCvGroups(len, EnsembleSize, ..., Seed) :75
  std::vector<int64_t> size((size_t)ensemble_size, 0);
  for (int32_t g : all) size[(size_t)g]++;
  const int32_t* grp = all.data() + row0;
  std::vector<uint8_t> mask((size_t)std::max<int64_t>(n, 1));
  std::vector<Classifier> all_cls;
  int rc = KLR_OK;
  e->out.clear();
  drop_folds(e);
  // member k on backend `be` and matrix `d` (e's own, or a worker's forked context and view)
  auto member = [&](const klr_backend& be, klr_mat* d, int k, klr_estimator** out, std::string& err) -> int {
    std::vector<uint8_t> mk((size_t)std::max<int64_t>(n, 1));
    for (int64_t i = 0; i < n; i++) mk[(size_t)i] = grp[i] != k;                   // filterCvGroup(data_train, groups, nil, k) :84
    if (be.set_active_rows(be.ctx, d, mk.data(), n_all - size[(size_t)k]) != KLR_OK) {
      err = std::string("set_active_rows failed: ") + (be.last_error ? be.last_error(be.ctx) : "");
      return KLR_EHIP;
    }
    klr_estimator* me = fresh_like(e);
    if (!me) { err = "cannot create member estimator"; return KLR_EINVAL; }
    me->be = be;
    *out = me;
    const int r = klr_estimator_estimate(me, d, lambda_auto, nla, scale, offset);
    if (r != KLR_OK) err = me->err;
    return r;
  };
  std::vector<klr_estimator*> mem((size_t)ensemble_size, nullptr);
  std::vector<int> mrc((size_t)ensemble_size, KLR_OK);
  std::vector<std::string> merr((size_t)ensemble_size);
  // members at once (cfg.threads: the reference runs them in config.Pool, kmerLr_estimator_ensemble.go:81), as the folds
  // of a cross-validation do
  Workers W;
  const int T = workers_open(e, data, std::min(std::max(1, e->cfg.threads), ensemble_size), W);
  if (T == 0) {
    for (int k = 0; k < ensemble_size; k++) {
      mrc[(size_t)k] = member(e->be, data, k, &mem[(size_t)k], merr[(size_t)k]);
      if (mrc[(size_t)k] != KLR_OK) break;
    }
    e->be.set_active_rows(e->be.ctx, data, nullptr, 0);
  } else {
    workers_run(W, ensemble_size, [&](int t, int k) {
      klr_backend be = e->be;
      be.ctx = W.ctx[(size_t)t];
      mrc[(size_t)k] = member(be, W.data[(size_t)t], k, &mem[(size_t)k], merr[(size_t)k]);
      return mrc[(size_t)k] == KLR_OK;
    });
    workers_close(e, W);
  }
  for (int k = 0; k < ensemble_size; k++) {
    klr_estimator* me = mem[(size_t)k];
    if (rc == KLR_OK && mrc[(size_t)k] != KLR_OK) { rc = mrc[(size_t)k]; e->err = merr[(size_t)k]; }
    if (!me) { if (rc == KLR_OK) rc = fail(e, KLR_EINVAL, "ensemble member %d did not run", k + 1); continue; }
    me->be = e->be;
    e->folds.push_back(me);
    if (rc != KLR_OK) continue;
    for (auto& r : me->trace) e->trace.push_back(r);                               // GetTrace :56-62
    for (int64_t j = 0; j < nla; j++) all_cls.push_back(me->out[(size_t)j]);
  }
  if (rc == KLR_OK) e->out.swap(all_cls);
  return rc;
}

static int save_ensemble_common(const klr_estimator* e, int64_t j, int ensemble_size, int64_t nla, const char* filename,
                                const char* type_name, const char* const* names, const char* summary, const int64_t* base_cols, int64_t nb1);

// CoeffIndex.Sub2Ind (kmerLr_coefficients_index.go:41-54; the library's klr_coeff_sub2ind, restated here because the driver
// is also built without the HIP library): column i of the pair matrix of n features -> (k1, k2), k1 <= k2
static void pair_of_column(int64_t n, int64_t i, int64_t* k1, int64_t* k2) {
  if (i < n) { *k1 = i; *k2 = i; return; }
  i -= n;
  const int64_t a = n - 2 - (int64_t)std::floor(std::sqrt((double)(-8 * i + 4 * n * (n - 1) - 7)) / 2.0 - 0.5);
  *k1 = a;
  *k2 = i + a + 1 - n * (n - 1) / 2 + (n - a) * ((n - a) - 1) / 2;
}

extern "C" int klr_estimator_save_ensemble(const klr_estimator* e, int64_t j, int ensemble_size, int64_t nla, const char* filename,
                                           const char* type_name, const char* const* names, const char* summary) {
  return save_ensemble_common(e, j, ensemble_size, nla, filename, type_name, names, summary, nullptr, 0);
}

// an ensemble of co-occurrence models: the members' columns are columns of the pair matrix of base_cols (nb1 columns incl.
// the bias, as klr_cooccurrence_prepare / klr_estimator_base_cols report them); Index holds the original features, Features
// the pair of every kept column (KmerLrFeatures of a co-occurrence ensemble, kmerLr_features.go:222-250)
extern "C" int klr_estimator_save_ensemble_pairs(const klr_estimator* e, int64_t j, int ensemble_size, int64_t nla, const char* filename,
                                                 const char* type_name, const char* const* names, const char* summary,
                                                 const int64_t* base_cols, int64_t nb1) {
  if (!base_cols || nb1 < 2) return KLR_EINVAL;
  return save_ensemble_common(e, j, ensemble_size, nla, filename, type_name, names, summary, base_cols, nb1);
}

static int save_ensemble_common(const klr_estimator* e, int64_t j, int ensemble_size, int64_t nla, const char* filename,
                                const char* type_name, const char* const* names, const char* summary, const int64_t* base_cols, int64_t nb1) {
  if (!e || !filename || ensemble_size < 1 || nla < 1 || j < 0 || j >= nla || (int64_t)e->out.size() < (int64_t)ensemble_size * nla)
    return KLR_EINVAL;
  // union of the members' features with a non-zero coefficient (AddKmerLr :277-286), ascending
  std::vector<int64_t> uni;
  for (int k = 0; k < ensemble_size; k++) {
    const Classifier& c = e->out[(size_t)(k * nla + j)];
    for (size_t p = 1; p < c.cols.size(); p++) if (c.theta[p] != 0.0) uni.push_back(c.cols[p]);
  }
  std::sort(uni.begin(), uni.end());
  uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
  const size_t nf = uni.size();
  std::vector<double> scale(nf + 1, 1.0), offset(nf + 1, 0.0);
  bool has_scale = false, has_offset = false;
  FILE* f = fopen(filename, "w");
  if (!f) return KLR_EINVAL;
  std::string name = type_name ? type_name : "scoresLr";
  if (summary && *summary) name += std::string(" [") + summary + "]";
  if (base_cols) {
    std::vector<int64_t> f1(nf), f2(nf), index;
    for (size_t k = 0; k < nf; k++) {
      int64_t k1 = 0, k2 = 0;
      if (uni[k] < 1 || uni[k] - 1 >= (nb1 - 1) * nb1 / 2) { fclose(f); return KLR_EINVAL; }
      pair_of_column(nb1 - 1, uni[k] - 1, &k1, &k2);
      if (k1 < 0 || k2 < k1 || k2 + 1 >= nb1) { fclose(f); return KLR_EINVAL; }
      f1[k] = base_cols[k1 + 1] - 1; f2[k] = base_cols[k2 + 1] - 1;
      index.push_back(f1[k]); index.push_back(f2[k]);
    }
    std::sort(index.begin(), index.end());
    index.erase(std::unique(index.begin(), index.end()), index.end());
    auto pos = [&](int64_t v) { return (size_t)(std::lower_bound(index.begin(), index.end(), v) - index.begin()); };
    std::vector<std::pair<size_t, size_t>> features(nf);
    for (size_t k = 0; k < nf; k++) features[k] = {pos(f1[k]), pos(f2[k])};
    write_head(f, e, name, true, index, features, names);
  } else {
    std::vector<int64_t> index(nf);
    std::vector<std::pair<size_t, size_t>> features(nf);
    for (size_t k = 0; k < nf; k++) { index[k] = uni[k] - 1; features[k] = {k, k}; }
    write_head(f, e, name, false, index, features, names);
  }
  fprintf(f, "  \"Distributions\": [\n");
  for (int k = 0; k < ensemble_size; k++) {
    const Classifier& c = e->out[(size_t)(k * nla + j)];
    std::vector<double> th(nf + 1, 0.0);
    th[0] = c.theta[0];
    for (size_t p = 1; p < c.cols.size(); p++) {
      if (c.theta[p] == 0.0) continue;
      const size_t u = (size_t)(std::lower_bound(uni.begin(), uni.end(), c.cols[p]) - uni.begin());
      th[u + 1] = c.theta[p];
      if (!c.scale.empty()) { scale[u + 1] = c.scale[p]; has_scale = true; }
      if (!c.offset.empty()) { offset[u + 1] = c.offset[p]; has_offset = true; }
    }
    fprintf(f, "    {\"Name\": \"vector:logistic regression\", \"Parameters\": [");
    for (size_t q = 0; q < th.size(); q++) fprintf(f, "%s%.17g", q ? "," : "", th[q]);
    fprintf(f, "], \"Distributions\": null},\n");
  }
  fprintf(f, "    {\"Name\": \"transform\", \"Parameters\": {\"Offset\": [");
  if (has_offset) for (size_t q = 0; q < offset.size(); q++) fprintf(f, "%s%.17g", q ? "," : "", offset[q]);
  fprintf(f, "], \"Scale\": [");
  if (has_scale) for (size_t q = 0; q < scale.size(); q++) fprintf(f, "%s%.17g", q ? "," : "", scale[q]);
  fprintf(f, "]}, \"Distributions\": null}\n  ]\n}\n");
  fclose(f);
  return KLR_OK;
}
