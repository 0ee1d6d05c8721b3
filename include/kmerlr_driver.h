/*
 * kmerlr_driver.h -- host side of the path above the kernels: the leapfrog path driver and feature
 * selection bookkeeping of KmerLrEstimator / ScoresLrEstimator, the hook records, and the .trace /
 * .path / model writers.  Plain C ABI.
 *
 * Mirrors (reference file:line):
 *   KmerLrEstimator.Estimate / estimate_loop / estimate_fixed / estimate / reestimate
 *                                   kmerLr_estimator.go:147-270, scoresLr_estimator.go:126-240
 *   featureSelection.Theta / restoreNonzero bookkeeping   kmerLr_feature_selection.go:194-219,299-307
 *   Trace.Export                    kmerLr_estimator_trace.go:49-78
 *   {Kmer,Scores}RegularizationPath kmerLr_estimator_path.go:41-117, scoresLr_estimator_path.go
 *   Config (hot-path fields)        kmerLr.go:32-62
 *
 * The driver is backend agnostic: it reaches the device only through the klr_backend function table,
 * whose entries have exactly the signatures of include/kmerlr.h.  The shipped table is
 * klr_hip_backend() (libkmerlr_hip.so); there is no CPU table in the product.
 */
#ifndef KMERLR_DRIVER_H
#define KMERLR_DRIVER_H

#include "kmerlr.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct klr_backend {
  klr_ctx* ctx;
  const char* (*last_error)(const klr_ctx*);
  int  (*mat_shape)(const klr_mat*, int64_t*, int64_t*, int64_t*, int64_t*);
  void (*mat_destroy)(klr_mat*);
  int  (*class_weights)(klr_ctx*, const klr_mat*, double[2]);
  int  (*gradient)(klr_ctx*, const klr_mat*, const int64_t*, const double*, int64_t, const double[2],
                   const double*, const double*, double*);
  int  (*loss)(klr_ctx*, const klr_mat*, const double*, const double[2], double, double*);
  int  (*select_columns)(klr_ctx*, const klr_mat*, const int64_t*, int64_t, const double*, const double*,
                         klr_mat**);
  int  (*sweep_select)(klr_ctx*, const klr_mat*, const klr_mat*, const int64_t*, const double*, int64_t,
                       const double[2], int64_t, double, double, const double*, const double*, uint8_t*,
                       double*, int64_t*, int*, double*, double*);
  int  (*cd_solve)(klr_ctx*, const klr_mat*, double*, const double[2], double, double, double, double, int,
                   int64_t, klr_iter_rec*, int64_t, int64_t*);
  int  (*set_active_rows)(klr_ctx*, klr_mat*, const uint8_t*, int64_t);
  int  (*log_pdf)(klr_ctx*, const klr_mat*, const double*, double*);
  int  (*expand_cooccurrence)(klr_ctx*, const klr_mat*, klr_mat**);      /* co-occurrence mode only; may be NULL */
  int  (*transform_fit)(klr_ctx*, const klr_mat*, int, double*, double*, int*);   /* per-fold TransformFull.Fit (validation sets only); may be NULL */
  /* concurrent folds (klr_config.threads_cv > 1); all three or none.  ctx_fork: another context for another host thread;
   * mat_view: the data for that context -- shared storage, its own row mask; ctx_release ends a forked context */
  int  (*ctx_fork)(klr_ctx*, klr_ctx**);
  void (*ctx_release)(klr_ctx*);
  int  (*mat_view)(klr_ctx*, const klr_mat*, klr_mat**);
  /* optional: a new run of consecutive reduced solves begins (klr_estimator_estimate).  The HIP backend sizes the first
   * batch of sweeps of a re-linearisation by what the same re-linearisation took in the solve before (the iterates do not
   * depend on it); the call forgets that history, so that a run is the same whatever the context did before */
  void (*solve_run_begin)(klr_ctx*);
  void (*solve_run_end)(klr_ctx*);       /* the run is over: later solves on the context start without a history again */
} klr_backend;

/* the HIP table for an existing context */
int klr_hip_backend(klr_ctx* ctx, klr_backend* out);

/* hot-path fields of Config (kmerLr.go:32-62); defaults = the CLI defaults (kmerLr_learn.go:116-149) */
typedef struct klr_config {
  double  lambda;            /* --lambda; NaN = use lambda_auto                      */
  int64_t max_features;      /* --max-features (fixed lambda only)                   */
  int64_t max_epochs;        /* --max-epochs, 0 = unbounded                          */
  int64_t max_iterations;    /* --max-iterations, 0 = solver default                 */
  double  epsilon;           /* --epsilon                                            */
  double  epsilon_lambda;    /* --epsilon-lambda                                     */
  double  epsilon_loss;      /* --epsilon-loss (default 1e-8)                        */
  int     eval_loss;         /* --eval-loss (forced on when epsilon_loss != 0)       */
  int     balance;           /* --balance: class weights n/(2 n_c), else [1,1]       */
  int     penalty_free;      /* --penalty-free                                       */
  int     save_trace;        /* --save-trace                                         */
  int     save_path;         /* --save-path                                          */
  int     verbose;           /* -v / -vv                                             */
  double  l2reg;             /* L2Reg of the estimator (0 in the reference's CLI)    */
  int     cooccurrence;      /* --co-occurrence (used by klr_cooccurrence_prepare's callers) */
  int64_t copreselection;    /* --co-preselection, 0 = none                          */
  double  validation_size;   /* --validation-size: fraction of the rows set aside to choose among the lambda_auto
                              * classifiers (getCvGroups, kmerLr_crossvalidation.go:56-90); 0 = none        */
  int     ensemble_size;     /* --ensemble-size inside klr_estimator_crossvalidate (0 / 1: none): every fold trains
                              * ensemble_size members, member k on the fold's training rows whose group (getCvGroups over
                              * the training rows, kmerLr_estimator_ensemble.go:75) is not k                */
  int     ensemble_summary;  /* --ensemble-summary of their predictions and losses: KLR_SUMMARY_*          */
  int     data_transform;    /* --data-transform as KLR_TR_* (0 none): only read where the driver has to FIT the
                              * transform itself, i.e. per fold when a validation set is split off
                              * (kmerLr_estimator_ensemble.go:116-117 fits on train + test rows)            */
  int     threads_cv;        /* --threads-cv: folds of klr_estimator_crossvalidate that run at once, each in its own
                              * host thread on a forked context and a view of the data (kmerLr_crossvalidation.go:176
                              * runs its folds in a thread pool); 0 / 1 = one after the other.  Needs the backend's
                              * ctx_fork / mat_view and a single rank; otherwise the folds run in turn.  The results
                              * do not depend on it.                                                          */
  int     threads;           /* --threads (config.Pool): members of klr_estimator_estimate_ensemble that run at once
                              * (kmerLr_estimator_ensemble.go:81), by the same means; 0 / 1 = in turn         */
} klr_config;

#define KLR_SUMMARY_MEAN    0   /* {Kmer,Scores}LrEnsemble.Summarize, kmerLr_classifier_ensemble.go:64-104 */
#define KLR_SUMMARY_PRODUCT 1
#define KLR_SUMMARY_MIN     2
#define KLR_SUMMARY_MAX     3

void klr_config_default(klr_config* cfg);

typedef struct klr_estimator klr_estimator;

/* NewKmerLrEstimator / NewScoresLrEstimator (kmerLr_estimator.go:45-67) */
int  klr_estimator_create(const klr_backend* be, const klr_config* cfg, klr_estimator** out);
void klr_estimator_destroy(klr_estimator* est);
const char* klr_estimator_error(const klr_estimator* est);
/* warm start from a model (kmerLr_estimator.go:62-64): cols ascending incl. 0 */
int  klr_estimator_set_model(klr_estimator* est, const int64_t* cols, const double* theta, int64_t c_plus_1);

/* KmerLrEstimator.Estimate (kmerLr_estimator.go:257-270): for every lambda_auto (ascending) run
 * estimate_loop on the SAME estimator (warm start); with cfg.lambda set run estimate_fixed instead.
 * scale/offset: TransformFull vectors (length dim) or NULL.  One classifier per lambda_auto is kept and
 * read back with klr_estimator_classifier. */
int  klr_estimator_estimate(klr_estimator* est, const klr_mat* data, const int64_t* lambda_auto,
                            int64_t n_lambda_auto, const double* scale, const double* offset);

/* i-th classifier: selected columns (ascending, incl. 0), coefficients, its lambda (= L1Reg/n) and the
 * selected transform.  Pointers stay valid until the next estimate/destroy; any may be NULL. */
int  klr_estimator_classifier(const klr_estimator* est, int64_t i, int64_t* c_plus_1, const int64_t** cols,
                              const double** theta, double* lambda, const double** scale, const double** offset);
int64_t klr_estimator_n_classifiers(const klr_estimator* est);

/* hook records of all solver iterations so far (Trace, kmerLr_estimator_trace.go:40-47) */
int  klr_estimator_trace(const klr_estimator* est, const klr_iter_rec** recs, int64_t* n);
/* number of leapfrog epochs of the last estimate call and their lambdas (the "Estimating parameters
 * with lambda=..." lines of kmerLr_estimator.go:246) */
int  klr_estimator_epochs(const klr_estimator* est, const double** lambdas, const int64_t** nonzero, int64_t* n);

/* k-fold cross-validation (crossvalidation / getCvGroups / filterCvGroup, kmerLr_crossvalidation.go:56-211,
 * learn_cv kmerLr_learn.go:71-87).  Folds are row masks on the resident matrix (no re-upload): fold f trains
 * a fresh estimator (cfg of `est`) on the rows with group != f, predicts the rows of group f
 * (Predict = log sigma(x.theta), kmerLr_classifier.go:107-123) and records train / test loss, for every
 * lambda_auto.  Group of GLOBAL row r: (r mod k) permuted by the documented shuffle (Go's
 * math/rand stream is not reproducible without Go); seed = -1: no shuffle, as in the reference.
 *   row0                global id of the first local row (sharded data), n_all = rows of the whole set
 *   predictions         out, n_lambda_auto x n_local (local rows; each row is tested in exactly one fold)
 *   groups              out, n_local (fold of every local row), nullable
 *   loss_train/test     out, n_lambda_auto x k_fold
 * Per-fold classifiers stay readable through klr_estimator_classifier (index fold*n_lambda_auto + j). */
int  klr_estimator_crossvalidate(klr_estimator* est, klr_mat* data, int64_t row0, int64_t n_all, int k_fold, int64_t seed,
                                 const int64_t* lambda_auto, int64_t n_lambda_auto, const double* scale,
                                 const double* offset, double* predictions, int32_t* groups, double* loss_train,
                                 double* loss_test);
/* fold assignment alone: groups[i] for global rows [row0, row0+n_local) */
void klr_cv_groups(int64_t n_all, int k_fold, int64_t seed, int64_t row0, int64_t n_local, int32_t* groups);
/* getCvGroups in full (kmerLr_crossvalidation.go:56-94): groups as above (-1 everywhere for k_fold <= 1) and the
 * validation flags -- row i of block [i0, i0 + fold) is flagged while i0 < val_ratio * n and i0 + fold < n --, both
 * permuted by the same shuffle.  validation may be NULL.  A row is in fold f's test set when groups == f, else in
 * its validation set when flagged, else in its training set (filterCvGroup :96-120). */
void klr_cv_groups_val(int64_t n_all, int k_fold, double val_ratio, int64_t seed, int64_t row0, int64_t n_local,
                       int32_t* groups, uint8_t* validation);
/* With cfg.validation_size > 0 klr_estimator_crossvalidate also accepts k_fold = 1 (no test set: the `learn` run of
 * kmerLr_learn.go:89-108 with --validation-size), estimates every lambda_auto on the training rows, keeps for each
 * fold ONLY the classifier with the lowest loss on the fold's validation rows (kmerLr_estimator_ensemble.go:119-135)
 * and fills row 0 of predictions / loss_train / loss_test alone; klr_estimator_n_classifiers is then k_fold (or 1).
 * With cfg.ensemble_size = E > 1 every fold trains E members (estimate_ensemble inside the fold,
 * kmerLr_estimator_ensemble.go:74-96,118); predictions and losses are the cfg.ensemble_summary of the members' values
 * (KmerLrEnsemble.Predict / .Loss); the fold's estimator (klr_estimator_fold) then holds the members' classifiers
 * member-major (member k, kept lambda_auto j at k * n_kept + j: klr_estimator_save_ensemble's layout).
 * Estimator of fold f after a cross-validation (its trace, path and classifiers: learn_parameters writes one
 * `_cv<f+1>` file set per fold, kmerLr_learn.go:45-66); owned by `est`, valid until its next call. */
klr_estimator* klr_estimator_fold(const klr_estimator* est, int fold);
/* With cfg.cooccurrence and cfg.copreselection = q > 0 klr_estimator_crossvalidate takes the BASE data and repeats the
 * co-occurrence set-up inside every fold as the reference does (kmerLr_estimator_ensemble.go:99-117): pre-selection of q
 * features on the fold's training rows, the pair matrix of the restricted data, its own transform (fitted by the backend
 * when cfg.data_transform is set; scale / offset of the call are the base data's transform for the pre-selection).  The
 * columns of a fold's classifiers are columns of ITS pair matrix: klr_estimator_base_cols on the fold's estimator gives
 * the base columns it was built from (as klr_cooccurrence_prepare's base_cols). */
int  klr_estimator_base_cols(const klr_estimator* fold_est, const int64_t** cols, int64_t* n);
/* SaveCrossvalidation / SaveCrossvalidationLoss (kmerLr_crossvalidation.go:122-165): rows in fold order */
int  klr_save_crossvalidation(const char* filename, const double* predictions, const uint8_t* labels,
                              const int32_t* groups, int64_t n, int k_fold);
int  klr_save_crossvalidation_loss(const char* filename, const double* loss_train, const double* loss_test, int k_fold);

/* Ensembles (estimate_ensemble, kmerLr_estimator_ensemble.go:74-96 / scoresLr_estimator_ensemble.go:74-96): member k
 * of `ensemble_size` >= 2 is trained by a fresh estimator on the rows whose group (klr_cv_groups with
 * k_fold = ensemble_size) is NOT k -- a row mask on the resident matrix -- for every lambda_auto, all members with
 * the same transform.  Classifiers are kept member-major (member k, lambda_auto j at k*n_lambda_auto + j). */
int  klr_estimator_estimate_ensemble(klr_estimator* est, klr_mat* data, int64_t row0, int64_t n_all, int ensemble_size,
                                     int64_t seed, const int64_t* lambda_auto, int64_t n_lambda_auto, const double* scale,
                                     const double* offset);
/* One model file for lambda_auto j of such an ensemble ({Kmer,Scores}LrEnsemble.AddKmerLr + ExportConfig,
 * kmerLr_classifier_ensemble.go:253-353,416-433): the union of the members' features (zero coefficients dropped),
 * one coefficient vector per member over that union, the shared transform; Name = "<type> [<summary>]". */
int  klr_estimator_save_ensemble(const klr_estimator* est, int64_t j, int ensemble_size, int64_t n_lambda_auto,
                                 const char* filename, const char* type_name, const char* const* names, const char* summary);
/* The same for members trained on a pair matrix (cfg.cooccurrence): base_cols / nb1 as klr_cooccurrence_prepare or
 * klr_estimator_base_cols report them; the file's Index holds the original features, Features the pair of every column. */
int  klr_estimator_save_ensemble_pairs(const klr_estimator* est, int64_t j, int ensemble_size, int64_t n_lambda_auto,
                                       const char* filename, const char* type_name, const char* const* names, const char* summary,
                                       const int64_t* base_cols, int64_t n_base_cols);

/* Co-occurrence set-up ({Kmer,Scores}LrEstimatorEnsemble.Estimate, kmerLr_estimator_ensemble.go:98-117,
 * scoresLr_estimator_ensemble.go:98-117).  With cfg.copreselection = q > 0 a first estimate_loop with lambda_auto = q
 * and WITHOUT co-occurrence runs on `data` (transform pre_scale / pre_offset, nullable), the data are restricted to
 * the q selected features (raw values, SelectData) and the estimator state is discarded (Reset); then the pair matrix
 * of the (restricted) data is written out (klr_mat_expand_cooccurrence) and returned -- an ordinary data matrix for
 * klr_estimator_estimate, klr_transform_fit, ... whose column j >= 1 is the feature pair
 *   (k1, k2) = klr_coeff_sub2ind(nb, j-1)  ->  original 0-based features (base_cols[k1+1]-1, base_cols[k2+1]-1).
 *   base_cols   out, capacity = dim of `data`: ascending columns of `data` kept (incl. 0); n_base_cols = nb + 1 */
int  klr_cooccurrence_prepare(const klr_backend* be, const klr_config* cfg, const klr_mat* data, const double* pre_scale,
                              const double* pre_offset, klr_mat** expanded, int64_t* base_cols, int64_t* n_base_cols,
                              char* errbuf, int64_t errcap);
/* model JSON with feature pairs (Cooccurrence = true): pairs[2k], pairs[2k+1] = 0-based original features of the
 * k-th selected column (equal for a single feature).  "Index" = the distinct features used, ascending; "Features" =
 * positions in Index (scoresLr_feature.go:75-80, kmerLr_features.go:224-235). */
int  klr_estimator_save_model_pairs(const klr_estimator* est, int64_t i, const char* filename, const char* type_name,
                                    const char* const* names, const int64_t* pairs);

/* Trace.Export (kmerLr_estimator_trace.go:49-78) */
int  klr_estimator_save_trace(const klr_estimator* est, const char* filename);
/* RegularizationPath.Export (kmerLr_estimator_path.go:41-73): lambda norm theta,... */
int  klr_estimator_save_path(const klr_estimator* est, const char* filename);
/* model JSON of classifier i: a ConfigDistribution {Name, Parameters, Distributions} (kmerLr_io.go:105-112).
 * Distributions = one "vector:logistic regression" per member, then "transform" {Offset, Scale}
 * (kmerLr_classifier_ensemble.go:416-424, kmerLr_transform.go:646-648).  Parameters:
 *   type "scoresLr":  {Cooccurrence, Features, Index, Names}                         (scoresLr_feature.go:72-86)
 *   type "kmerLr" after klr_estimator_set_kmer_features: the reference's KmerLrFeatures export
 *                     {M, N, Binarize, Complement, Cooccurrence, Reverse, Revcomp, MaxAmbiguous, Alphabet, Kmers,
 *                      Features} in that order (kmerLr_features.go:222-250); Kmers = the class strings
 *                     ("kmer|revcomp") of the model's features, Features = pairs of positions in Kmers -- the file
 *                     `loss` / `predict` rebuild their k-mer counter from (kmerLr_loss.go:53-57).
 * names: one string per feature of the data (class strings for kmerLr) or NULL.  A model with a non-finite
 * coefficient is refused (KLR_EINVAL): NaN is not JSON (Go's encoder fails the same way). */
typedef struct {
  int M, N;
  int binarize, complement, reverse, revcomp;
  int n_max_ambiguous;              /* entries of max_ambiguous (the reference's default --max-ambiguous=-1 gives [-1]) */
  const int* max_ambiguous;
  const char* alphabet;             /* "nucleotide alphabet", "gapped nucleotide alphabet" (Alphabet.String()) */
} klr_kmer_features;
int  klr_estimator_set_kmer_features(klr_estimator* est, const klr_kmer_features* kf);
int  klr_estimator_save_model(const klr_estimator* est, int64_t i, const char* filename, const char* type_name,
                              const char* const* names);

#ifdef __cplusplus
}
#endif
#endif
