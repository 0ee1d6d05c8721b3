// kmerlr -- command line of the MI355X build: `learn` and `loss` for --type=kmerLr|scoresLr, mirroring the
// reference CLI for the hot path (kmerLr.go:79-175, kmerLr_learn.go:112-338, scoresLr_learn.go:106-286,
// kmerLr_loss.go:32-106).  Host-side ingest (CSV tables, FASTA k-mer counts, transform fitting) feeds the
// C ABI of libkmerlr_hip.so; all training arithmetic runs on the GPU.  There is no CPU fallback.
//
//   kmerlr [--type=kmerLr|scoresLr] [-v|-vv] [--device=N] learn [options] [<M> <N>] FG BG BASENAME
//   kmerlr [--type=...] loss [--lambda=F] MODEL.json FG BG
#include <algorithm>
#include <cctype>
#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/kmerlr_driver.h"

struct DataSet {                       // ScoresDataSet / KmerDataSet: fg rows first (labels true)
  std::vector<int64_t> rowptr{0}, col;
  std::vector<double> val;
  std::vector<uint8_t> labels;
  int64_t dim = 0;
  std::vector<std::string> names;
};

[[noreturn]] static void die(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
  fputc('\n', stderr);
  exit(1);
}

static int g_verbose = 0;
static std::string g_seed = "1";
static void say(int level, const char* fmt, ...) {
  if (g_verbose < level) return;
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
}

// ---- scores tables: read_scores_table + convert_scores (scoresLr_data.go:56-145) ---------------------
static void read_scores(const std::string& fn, bool header, uint8_t label, DataSet& d) {
  say(1, "Reading scores from `%s'... ", fn.c_str());
  std::ifstream f(fn);
  if (!f) die("failed\ncannot open `%s'", fn.c_str());
  std::string line;
  bool need_header = header;
  int64_t lineno = 0;
  while (std::getline(f, line)) {
    lineno++;
    if (line.empty()) continue;
    std::vector<std::string> fields;
    { std::stringstream ss(line); std::string t; while (std::getline(ss, t, ',')) if (!t.empty()) fields.push_back(t); }
    if (need_header) { if (d.names.empty()) d.names = fields; need_header = false; continue; }
    if (d.dim == 0) d.dim = (int64_t)fields.size() + 1;
    if ((int64_t)fields.size() + 1 != d.dim) die("failed\nError: data has variable number of features");
    d.col.push_back(0); d.val.push_back(1.0);                      // bias
    for (size_t j = 0; j < fields.size(); j++) {
      char* end = nullptr;
      std::string t = fields[j];
      size_t a = t.find_first_not_of(" \t\r"), b = t.find_last_not_of(" \t\r");
      t = a == std::string::npos ? "" : t.substr(a, b - a + 1);
      const double v = strtod(t.c_str(), &end);
      if (t.empty() || *end) die("failed\nparsing scores failed at line `%lld'", (long long)lineno);
      d.col.push_back((int64_t)j + 1); d.val.push_back(v);         // every column stored, zeros included
    }
    d.rowptr.push_back((int64_t)d.col.size());
    d.labels.push_back(label);
  }
  say(1, "done\n");
}

// ---- k-mer classes (klr_kmer_classes: NewKmerCounter's arguments, kmerLr_learn.go:94,116-122,193-214); the counting
// itself runs on the device (klr_mat_from_sequences_classes)
struct KmerSpec {
  int M = 0, N = 0;
  bool revcomp = false, reverse = false, complement = false, binarize = false;
  std::string alphabet = "nucleotide";          // nucleotide | gapped-nucleotide | iupac-nucleotide  (Alphabet.String(): "<name> alphabet")
  std::vector<int> max_ambiguous{-1};           // --max-ambiguous: one value or one per length
  klr_kmer_classes* kc = nullptr;
  int64_t m = 0;
};

// alphabet_from_string (kmerLr_features.go:254-270): lower case, '-' -> ' ', optional " alphabet" suffix
static std::string alphabet_key(std::string a) {
  for (auto& c : a) { c = (char)tolower((unsigned char)c); if (c == '-') c = ' '; }
  const std::string suf = " alphabet";
  if (a.size() < suf.size() || a.compare(a.size() - suf.size(), suf.size(), suf) != 0) a += suf;
  return a;
}

static void kmer_prepare(KmerSpec& s, std::vector<std::string>* names) {
  klr_kmer_spec sp;
  memset(&sp, 0, sizeof(sp));
  sp.M = s.M; sp.N = s.N; sp.complement = s.complement; sp.reverse = s.reverse; sp.revcomp = s.revcomp;
  const std::string key = alphabet_key(s.alphabet);
  if (key == "nucleotide alphabet") sp.alphabet = KLR_ALPHABET_NUCLEOTIDE;
  else if (key == "gapped nucleotide alphabet") sp.alphabet = KLR_ALPHABET_GAPPED;
  else if (key == "iupac nucleotide alphabet" || key == "ambiguous nucleotide alphabet") sp.alphabet = KLR_ALPHABET_IUPAC;
  else die("invalid alphabet");
  s.alphabet = sp.alphabet == KLR_ALPHABET_IUPAC ? "iupac nucleotide alphabet" : key;
  // a scalar -1 (the default) bounds nothing; other values go to the class tables
  const bool unbounded = s.max_ambiguous.size() == 1 && s.max_ambiguous[0] < 0;
  sp.n_max_ambiguous = unbounded ? 0 : (int)s.max_ambiguous.size();
  sp.max_ambiguous = unbounded ? nullptr : s.max_ambiguous.data();
  char eb[256] = "";
  if (klr_kmer_classes_create(&sp, &s.kc, eb, sizeof(eb)) != 0) die("%s", eb);
  s.m = klr_kmer_classes_count(s.kc);
  if (names) {
    names->clear();
    char buf[128];
    for (int64_t c = 0; c < s.m; c++) { klr_kmer_class_name(s.kc, c, buf, sizeof(buf)); names->push_back(buf); }
  }
}

static void parse_max_ambiguous(const std::string& v, int M, int N, std::vector<int>& out) {     // kmerLr_learn.go:205-214
  out.clear();
  std::stringstream ss(v); std::string t;
  while (std::getline(ss, t, ',')) out.push_back(atoi(t.c_str()));
  if (out.size() != 1 && (int)out.size() != N - M + 1) die("invalid option `--max-ambiguous'");
}

struct SeqSet { std::string blob; std::vector<int64_t> off{0}; std::vector<uint8_t> labels; };

// import_fasta (kmerLr_data.go:101-123): sequences only; counting happens on the device
static void read_fasta(const std::string& fn, uint8_t label, SeqSet& q) {
  say(1, "Reading sequences from `%s'... ", fn.c_str());
  std::ifstream f(fn);
  if (!f) die("failed\ncannot open `%s'", fn.c_str());
  std::string line; bool have = false;
  while (std::getline(f, line)) {
    if (!line.empty() && line[0] == '>') {
      if (have) { q.off.push_back((int64_t)q.blob.size()); q.labels.push_back(label); }
      have = true; continue;
    }
    if (have) for (char c : line) if (!isspace((unsigned char)c)) q.blob.push_back(c);
  }
  if (have) { q.off.push_back((int64_t)q.blob.size()); q.labels.push_back(label); }
  say(1, "done\n");
}

// scan_sequences + convert_counts on the device (klr_mat_from_sequences)
static klr_mat* count_kmers(klr_ctx* ctx, const SeqSet& q, const KmerSpec& ks) {
  if (q.labels.empty()) die("Error: no training data given");
  say(1, "Counting k-mers on the device... ");
  klr_mat* mat = nullptr;
  const std::string& blob = q.blob.empty() ? std::string("N") : q.blob;
  if (klr_mat_from_sequences_classes(ctx, blob.data(), q.off.data(), (int64_t)q.labels.size(), ks.kc, ks.binarize ? 1 : 0,
                                     q.labels.data(), 0, &mat) != KLR_OK)
    die("failed\n%s", klr_last_error(ctx));
  say(1, "done\n");
  return mat;
}

// ---- TransformFull.Fit (kmerLr_transform.go:38-57): column statistics on the device (klr_transform_fit) -----
static void fit_transform(klr_ctx* ctx, const klr_mat* mat, int64_t dim, const std::string& kind_, std::vector<double>& scale,
                          std::vector<double>& offset) {
  std::string kind = kind_;
  for (auto& c : kind) c = (char)tolower((unsigned char)c);
  scale.clear(); offset.clear();
  if (kind.empty() || kind == "none") return;
  int k = 0;
  if (kind == "standardizer") { k = KLR_TR_STANDARDIZER; say(1, "WARNING: Standardizing data will result in dense data and slow down computations!\n"); }
  else if (kind == "variance-scaler") k = KLR_TR_VARIANCE;
  else if (kind == "max-abs-scaler") k = KLR_TR_MAXABS;
  else if (kind == "mean-scaler") k = KLR_TR_MEAN;
  else die("invalid data transform");
  say(1, "Fitting data transform (%s)... ", kind.c_str());
  scale.assign((size_t)dim, 0.0); offset.assign((size_t)dim, 0.0);
  int has_offset = 0;
  if (klr_transform_fit(ctx, mat, k, scale.data(), offset.data(), &has_offset) != KLR_OK) die("failed\n%s", klr_last_error(ctx));
  if (!has_offset) offset.clear();
  say(1, "done\n");
}

// ---- tiny reader for the model JSON written by klr_estimator_save_model ---------------------------------
static std::vector<double> json_numbers_after(const std::string& s, const std::string& key, size_t from, size_t* endpos) {
  std::vector<double> out;
  size_t p = s.find("\"" + key + "\"", from);
  if (p == std::string::npos) die("invalid config file: missing %s", key.c_str());
  p = s.find('[', p);
  size_t e = p;
  int depth = 0;
  for (; e < s.size(); e++) { if (s[e] == '[') depth++; else if (s[e] == ']') { if (--depth == 0) break; } }
  std::string body = s.substr(p + 1, e - p - 1);
  for (auto& c : body) if (c == '[' || c == ']' || c == ',') c = ' ';
  std::stringstream ss(body); double v;
  while (ss >> v) out.push_back(v);
  if (endpos) *endpos = e;
  return out;
}

// ---- the reference's KmerLrFeatures parameters in a model file (kmerLr_features.go:164-220) --------------------
static size_t json_key(const std::string& s, const std::string& key, size_t from = 0) {
  const size_t p = s.find("\"" + key + "\"", from);
  return p == std::string::npos ? p : s.find(':', p) + 1;
}
static bool json_has(const std::string& s, const std::string& key) { return json_key(s, key) != std::string::npos; }
static long json_int(const std::string& s, const std::string& key) {
  const size_t p = json_key(s, key);
  if (p == std::string::npos) die("invalid config file: missing %s", key.c_str());
  return strtol(s.c_str() + p, nullptr, 10);
}
static bool json_bool(const std::string& s, const std::string& key, bool def, bool required) {
  const size_t p = json_key(s, key);
  if (p == std::string::npos) { if (required) die("invalid config file: missing %s", key.c_str()); return def; }
  size_t q = p; while (q < s.size() && isspace((unsigned char)s[q])) q++;
  return s.compare(q, 4, "true") == 0;
}
static std::string json_string(const std::string& s, const std::string& key) {
  const size_t p = json_key(s, key);
  if (p == std::string::npos) die("invalid config file: missing %s", key.c_str());
  const size_t q0 = s.find('"', p), q1 = s.find('"', q0 + 1);
  return s.substr(q0 + 1, q1 - q0 - 1);
}
static std::vector<std::string> json_strings(const std::string& s, const std::string& key) {
  std::vector<std::string> out;
  size_t p = json_key(s, key);
  if (p == std::string::npos) die("invalid config file: missing %s", key.c_str());
  p = s.find('[', p);
  const size_t e = s.find(']', p);
  for (size_t q = s.find('"', p); q != std::string::npos && q < e; ) {
    const size_t q1 = s.find('"', q + 1);
    out.push_back(s.substr(q + 1, q1 - q - 1));
    q = s.find('"', q1 + 1);
  }
  return out;
}
static std::vector<int> json_ints_or_null(const std::string& s, const std::string& key) {
  std::vector<int> out;
  size_t p = json_key(s, key);
  if (p == std::string::npos) die("invalid config file: missing %s", key.c_str());
  while (p < s.size() && isspace((unsigned char)s[p])) p++;
  if (s.compare(p, 4, "null") == 0) return out;
  const size_t e = s.find(']', p);
  std::string body = s.substr(p + 1, e - p - 1);
  for (auto& c : body) if (c == ',') c = ' ';
  std::stringstream ss(body); int v;
  while (ss >> v) out.push_back(v);
  return out;
}

// KmerLrFeatures.ImportConfig: the counter of a kmerLr model and the classes of its features
static void kmer_spec_from_model(const std::string& js, KmerSpec& ks, std::vector<double>& index) {
  ks.M = (int)json_int(js, "M"); ks.N = (int)json_int(js, "N");
  ks.binarize = json_bool(js, "Binarize", false, true);
  ks.complement = json_bool(js, "Complement", false, true);
  ks.reverse = json_bool(js, "Reverse", false, true);
  ks.revcomp = json_bool(js, "Revcomp", false, true);
  ks.max_ambiguous = json_ints_or_null(js, "MaxAmbiguous");
  if (ks.max_ambiguous.empty()) ks.max_ambiguous.push_back(-1);
  ks.alphabet = json_string(js, "Alphabet");
  kmer_prepare(ks, nullptr);
  index.clear();
  for (const std::string& k : json_strings(js, "Kmers")) {
    const int64_t c = klr_kmer_class_find(ks.kc, k.c_str());         // rel.EquivalenceClass(strings.Split(str, "|")[0])
    if (c < 0) die("invalid config file: `%s' is not a k-mer of the model's alphabet and lengths", k.c_str());
    index.push_back((double)c);
  }
}

// The data restricted to a model's features, transform applied (SelectData, kmerLr_classifier.go:125-147): plain
// models select the Index columns; co-occurrence models (Cooccurrence = true) select them, write out the pair matrix
// of that restriction and pick the columns of the Features pairs.  theta is permuted along if the columns need sorting.
// Summarize ({Kmer,Scores}LrEnsemble.Summarize, kmerLr_classifier_ensemble.go:64-104)
static double summarize(const std::string& summary, const std::vector<double>& x) {
  if (x.empty()) return NAN;
  if (x.size() == 1 && summary.empty()) return x[0];
  double r;
  if (summary == "product") { r = 1.0; for (double v : x) r *= v; }
  else if (summary == "min") { r = x[0]; for (double v : x) if (r > v) r = v; }
  else if (summary == "max") { r = x[0]; for (double v : x) if (r < v) r = v; }
  else if (summary == "mean") { r = 0.0; for (double v : x) r += v; r /= (double)x.size(); }
  else die("no summary given for ensemble classifier");
  return r;
}

// members: one coefficient vector per ensemble member (a single one for plain models); summary from "Name [summary]"
static klr_mat* model_data(klr_ctx* ctx, const klr_mat* mat, const std::string& js, std::vector<std::vector<double>>& members,
                           std::string& summary, const std::vector<double>* kmer_index = nullptr) {
  size_t pos = 0;
  // scoresLr models name their features by "Index"; kmerLr models by the class strings in "Kmers" (kmer_index: their
  // columns in the counter rebuilt from the model)
  std::vector<double> index = kmer_index ? *kmer_index : json_numbers_after(js, "Index", 0, &pos);
  const size_t dpos = js.find("\"Distributions\"");
  members.clear();
  summary.clear();
  {
    // top-level Name: "kmerLr" or "kmerLr [summary]" (kmerLr_classifier_ensemble.go:380,426-430)
    const size_t np = js.find("\"Name\"");
    const size_t q0 = js.find('"', js.find(':', np)), q1 = js.find('"', q0 + 1);
    const std::string name = js.substr(q0 + 1, q1 - q0 - 1);
    const size_t lb = name.find('['), rb = name.find(']');
    if (lb != std::string::npos && rb != std::string::npos && rb > lb) summary = name.substr(lb + 1, rb - lb - 1);
  }
  const size_t tpos = js.find("\"transform\"", dpos);
  for (size_t p = js.find("vector:logistic regression", dpos); p != std::string::npos && p < tpos; p = js.find("vector:logistic regression", pos)) {
    members.push_back(json_numbers_after(js, "Parameters", p, &pos));
  }
  if (members.empty()) die("invalid config file");
  std::vector<double>& theta = members[0];
  std::vector<double> offset = json_numbers_after(js, "Offset", tpos, nullptr);
  std::vector<double> scale = json_numbers_after(js, "Scale", tpos, nullptr);
  const size_t cp = js.find("\"Cooccurrence\"");
  const bool co = cp != std::string::npos && js.compare(js.find(':', cp) + 1, 5, " true") == 0;
  std::vector<int64_t> cols{0};
  for (double v : index) cols.push_back((int64_t)v + 1);
  klr_mat* red = nullptr;
  if (!co) {
    if (theta.size() != index.size() + 1) die("invalid config file");
    if (klr_mat_select_columns(ctx, mat, cols.data(), (int64_t)cols.size(), scale.empty() ? nullptr : scale.data(),
                               offset.empty() ? nullptr : offset.data(), &red) != KLR_OK)
      die("%s", klr_last_error(ctx));
    return red;
  }
  std::vector<double> feat = json_numbers_after(js, "Features", 0, nullptr);
  const size_t nf = feat.size() / 2;
  if (theta.size() != nf + 1) die("invalid config file");
  const int64_t q = (int64_t)index.size();
  klr_mat *base = nullptr, *ex = nullptr;
  if (klr_mat_select_columns(ctx, mat, cols.data(), (int64_t)cols.size(), nullptr, nullptr, &base) != KLR_OK) die("%s", klr_last_error(ctx));
  if (klr_mat_expand_cooccurrence(ctx, base, &ex) != KLR_OK) die("%s", klr_last_error(ctx));
  std::vector<std::pair<int64_t, size_t>> order;                       // expanded column, position in the model
  for (size_t k = 0; k < nf; k++) {
    const int64_t a = (int64_t)std::min(feat[2 * k], feat[2 * k + 1]), b = (int64_t)std::max(feat[2 * k], feat[2 * k + 1]);
    if (a < 0 || b >= q) die("invalid config file");
    order.emplace_back(klr_coeff_ind2sub(q, a, b), k);
  }
  std::sort(order.begin(), order.end());
  std::vector<int64_t> ecols{0};
  std::vector<double> th{theta[0]}, sc, of;
  if (!scale.empty()) sc.push_back(scale[0]);
  if (!offset.empty()) of.push_back(offset[0]);
  for (auto& pr : order) {
    ecols.push_back(pr.first); th.push_back(theta[pr.second + 1]);
    if (!scale.empty()) sc.push_back(scale[pr.second + 1]);
    if (!offset.empty()) of.push_back(offset[pr.second + 1]);
  }
  if (klr_mat_select_columns(ctx, ex, ecols.data(), (int64_t)ecols.size(), sc.empty() ? nullptr : sc.data(), of.empty() ? nullptr : of.data(), &red) != KLR_OK)
    die("%s", klr_last_error(ctx));
  klr_mat_destroy(ex); klr_mat_destroy(base);
  if (members.size() != 1) die("co-occurrence ensembles are not part of this build");
  theta = th;
  return red;
}

struct Options { std::map<std::string, std::string> kv; std::vector<std::string> pos; };
static Options parse(int argc, char** argv, int from) {
  Options o;
  for (int i = from; i < argc; i++) {
    std::string a = argv[i];
    if (a.rfind("--", 0) == 0) {
      size_t eq = a.find('=');
      if (eq == std::string::npos) o.kv[a.substr(2)] = "true"; else o.kv[a.substr(2, eq - 2)] = a.substr(eq + 1);
    } else if (a == "-v") g_verbose = std::max(g_verbose, 1);
    else if (a == "-vv") g_verbose = 2;
    else o.pos.push_back(a);
  }
  return o;
}
static std::string opt(const Options& o, const char* k, const char* def) { auto it = o.kv.find(k); return it == o.kv.end() ? def : it->second; }
static bool flag(const Options& o, const char* k) { return o.kv.count(k) != 0; }

int main(int argc, char** argv) {
  std::string type = "kmerLr";
  int device = 0;
  int ci = 1;
  for (; ci < argc; ci++) {
    std::string a = argv[ci];
    if (a.rfind("--type=", 0) == 0) type = a.substr(7);
    else if (a.rfind("--device=", 0) == 0) device = atoi(a.c_str() + 9);
    else if (a == "-v") g_verbose = std::max(g_verbose, 1);
    else if (a == "-vv") g_verbose = 2;
    else if (a.rfind("--threads", 0) == 0) continue;                                     // accepted, unused
    else if (a.rfind("--seed=", 0) == 0) g_seed = a.substr(7);
    else break;
  }
  if (ci >= argc) {
    printf("usage: kmerlr [--type=kmerLr|scoresLr] [-v] [--device=N] <learn|loss|predict> ...\n");
    return 0;
  }
  const std::string cmd = argv[ci];
  Options o = parse(argc, argv, ci + 1);
  const bool scores = type == "scoresLr" || type == "scoreslr";
  if (!scores && type != "kmerLr" && type != "kmerlr") die("invalid type `%s'", type.c_str());

  klr_ctx* ctx = nullptr;
  if (klr_ctx_create(device, &ctx) != KLR_OK) die("%s", klr_last_error(nullptr));
  klr_backend be;
  if (klr_hip_backend(ctx, &be) != KLR_OK) die("%s", klr_last_error(ctx));

  if (cmd == "learn") {
    klr_config cfg;
    klr_config_default(&cfg);
    cfg.lambda = strtod(opt(o, "lambda", "NaN").c_str(), nullptr);
    std::vector<int64_t> la;
    { std::stringstream ss(opt(o, "lambda-auto", "0")); std::string t; while (std::getline(ss, t, ',')) la.push_back(atoll(t.c_str())); }
    std::sort(la.begin(), la.end());
    if (!std::isnan(cfg.lambda) && !(la.size() == 1 && la[0] == 0)) die("options --lambda and --lambda-auto are incompatible");
    cfg.max_features = atoll(opt(o, "max-features", "0").c_str());
    cfg.max_epochs = atoll(opt(o, "max-epochs", "0").c_str());
    cfg.max_iterations = atoll(opt(o, "max-iterations", "0").c_str());
    cfg.epsilon = strtod(opt(o, "epsilon", "0").c_str(), nullptr);
    cfg.epsilon_lambda = strtod(opt(o, "epsilon-lambda", "0").c_str(), nullptr);
    cfg.epsilon_loss = strtod(opt(o, "epsilon-loss", "1e-8").c_str(), nullptr);
    cfg.eval_loss = flag(o, "eval-loss") || cfg.epsilon_loss != 0.0;
    cfg.balance = flag(o, "balance");
    cfg.penalty_free = flag(o, "penalty-free");
    cfg.save_trace = flag(o, "save-trace");
    cfg.save_path = flag(o, "save-path");
    cfg.verbose = g_verbose;
    const int kfold = atoi(opt(o, "k-fold-cv", "1").c_str());
    const long long seed = atoll(opt(o, "seed", g_seed.c_str()).c_str());
    if (kfold < 1) die("invalid --k-fold-cv");
    const int ensemble = atoi(opt(o, "ensemble-size", "1").c_str());
    const std::string summary = opt(o, "ensemble-summary", "mean");
    if (ensemble < 1) die("invalid --ensemble-size");
    if (summary != "mean" && summary != "product" && summary != "min" && summary != "max") die("invalid --ensemble-summary");
    if (ensemble > 1 && kfold > 1) die("--ensemble-size together with --k-fold-cv is not part of this build");
    if (ensemble > 1 && !std::isnan(cfg.lambda)) die("--ensemble-size needs --lambda-auto in this build");
    cfg.cooccurrence = flag(o, "co-occurrence") ? 1 : 0;
    cfg.copreselection = atoll(opt(o, "co-preselection", "0").c_str());
    if (cfg.copreselection != 0 && !cfg.cooccurrence) die("--co-preselection needs --co-occurrence");
    if (cfg.cooccurrence && cfg.copreselection > 0 && kfold > 1)
      die("--co-preselection together with --k-fold-cv is not part of this build (the pre-selection would have to run per fold)");

    DataSet d;
    SeqSet sq;
    std::string base;
    KmerSpec ks;
    if (scores) {
      if (o.pos.size() != 3) die("usage: kmerlr --type=scoresLr learn [options] FG.table BG.table BASENAME");
      read_scores(o.pos[0], flag(o, "header"), 1, d);
      read_scores(o.pos[1], flag(o, "header"), 0, d);
      base = o.pos[2];
    } else {
      if (o.pos.size() != 5) die("usage: kmerlr learn [options] M N FG.fa BG.fa BASENAME");
      ks.M = atoi(o.pos[0].c_str()); ks.N = atoi(o.pos[1].c_str());
      if (ks.M < 1 || ks.N < ks.M) die("invalid k-mer lengths");
      ks.revcomp = flag(o, "revcomp"); ks.binarize = flag(o, "binarize");
      ks.reverse = flag(o, "reverse"); ks.complement = flag(o, "complement");
      ks.alphabet = opt(o, "alphabet", "nucleotide");
      parse_max_ambiguous(opt(o, "max-ambiguous", "-1"), ks.M, ks.N, ks.max_ambiguous);
      kmer_prepare(ks, &d.names);
      read_fasta(o.pos[2], 1, sq);
      read_fasta(o.pos[3], 0, sq);
      d.dim = ks.m + 1; d.labels = sq.labels;
      base = o.pos[4];
    }
    const int64_t n = (int64_t)d.labels.size();
    if (n == 0) die("Error: no training data given");
    klr_mat* mat = nullptr;
    if (scores) {
      if (klr_mat_from_csr(ctx, n, d.dim, d.rowptr.data(), d.col.data(), d.val.data(), d.labels.data(), 0, &mat) != KLR_OK)
        die("%s", klr_last_error(ctx));
    } else {
      mat = count_kmers(ctx, sq, ks);
    }
    // co-occurrence mode ({Kmer,Scores}LrEstimatorEnsemble.Estimate, kmerLr_estimator_ensemble.go:98-117): optional
    // pre-selection with the transform fitted on the base data, then the pair matrix takes the place of the data
    klr_mat* base_mat = nullptr;
    std::vector<int64_t> base_cols;
    int64_t nb1 = 0;
    if (cfg.cooccurrence) {
      std::vector<double> ps, po;
      if (cfg.copreselection > 0) fit_transform(ctx, mat, d.dim, opt(o, "data-transform", ""), ps, po);
      base_cols.assign((size_t)d.dim, 0);
      klr_mat* ex = nullptr;
      char eb[512] = "";
      say(1, "Compiling co-occurrence features... ");
      if (klr_cooccurrence_prepare(&be, &cfg, mat, ps.empty() ? nullptr : ps.data(), po.empty() ? nullptr : po.data(), &ex, base_cols.data(), &nb1,
                                   eb, sizeof(eb)) != KLR_OK)
        die("failed\n%s", eb);
      say(1, "done\n");
      base_mat = mat;
      mat = ex;
      d.dim = klr_coeff_dim(nb1 - 1);
    }
    // 0-based original features of expanded column j >= 1
    auto pair_of = [&](int64_t j, int64_t* f1, int64_t* f2) {
      int64_t k1, k2;
      klr_coeff_sub2ind(nb1 - 1, j - 1, &k1, &k2);
      *f1 = base_cols[(size_t)k1 + 1] - 1; *f2 = base_cols[(size_t)k2 + 1] - 1;
    };
    auto save_model = [&](klr_estimator* e, int64_t i, const char* fn, const std::vector<const char*>& nm) {
      const char* tn = scores ? "scoresLr" : "kmerLr";
      if (!cfg.cooccurrence) return klr_estimator_save_model(e, i, fn, tn, nm.empty() ? nullptr : nm.data());
      int64_t c1 = 0; const int64_t* cc = nullptr;
      klr_estimator_classifier(e, i, &c1, &cc, nullptr, nullptr, nullptr, nullptr);
      std::vector<int64_t> pairs;
      for (int64_t k = 1; k < c1; k++) { int64_t f1, f2; pair_of(cc[k], &f1, &f2); pairs.push_back(f1); pairs.push_back(f2); }
      if (pairs.empty()) pairs.push_back(0);
      return klr_estimator_save_model_pairs(e, i, fn, tn, nm.empty() ? nullptr : nm.data(), pairs.data());
    };
    std::vector<double> scale, offset;
    fit_transform(ctx, mat, d.dim, opt(o, "data-transform", ""), scale, offset);
    klr_estimator* est = nullptr;
    if (klr_estimator_create(&be, &cfg, &est) != KLR_OK) die("cannot create estimator");
    if (!scores) {
      // kmerLr models carry their counter (KmerLrFeatures.ExportConfig, kmerLr_features.go:222-250)
      klr_kmer_features kf;
      memset(&kf, 0, sizeof(kf));
      kf.M = ks.M; kf.N = ks.N; kf.binarize = ks.binarize; kf.complement = ks.complement; kf.reverse = ks.reverse; kf.revcomp = ks.revcomp;
      kf.n_max_ambiguous = (int)ks.max_ambiguous.size(); kf.max_ambiguous = ks.max_ambiguous.data();
      kf.alphabet = ks.alphabet.c_str();
      klr_estimator_set_kmer_features(est, &kf);
    }
    std::vector<const char*> names;
    for (auto& s : d.names) names.push_back(s.c_str());
    if (kfold > 1) {
      // learn_cv (kmerLr_learn.go:71-87): k folds, per-fold models, joined prediction / loss tables
      if (!std::isnan(cfg.lambda)) die("--k-fold-cv needs --lambda-auto in this build");
      const int64_t nla = (int64_t)la.size();
      std::vector<double> pred((size_t)(nla * n)), ltr((size_t)(nla * kfold)), lte((size_t)(nla * kfold));
      std::vector<int32_t> groups((size_t)n);
      if (klr_estimator_crossvalidate(est, mat, 0, n, kfold, seed, la.data(), nla, scale.empty() ? nullptr : scale.data(),
                                      offset.empty() ? nullptr : offset.data(), pred.data(), groups.data(), ltr.data(), lte.data()) != KLR_OK)
        die("%s", klr_estimator_error(est));
      char fn[4096];
      for (int f = 0; f < kfold; f++) for (int64_t j = 0; j < nla; j++) {                  // learn_parameters :42-66
        if (nla == 1) snprintf(fn, sizeof(fn), "%s_cv%d.json", base.c_str(), f + 1);
        else snprintf(fn, sizeof(fn), "%s_cv%d_%lld.json", base.c_str(), f + 1, (long long)la[(size_t)j]);
        say(1, "Exporting model to `%s'... ", fn);
        if (save_model(est, f * nla + j, fn, names) != KLR_OK) die("failed");
        say(1, "done\n");
      }
      for (int64_t j = 0; j < nla; j++) {                                                    // learn_cv :78-86
        if (nla == 1) snprintf(fn, sizeof(fn), "%s.table", base.c_str()); else snprintf(fn, sizeof(fn), "%s_%lld.table", base.c_str(), (long long)la[(size_t)j]);
        say(1, "Exporting cross-validation results to `%s'... ", fn);
        if (klr_save_crossvalidation(fn, pred.data() + j * n, d.labels.data(), groups.data(), n, kfold) != KLR_OK) die("failed");
        say(1, "done\n");
        if (nla == 1) snprintf(fn, sizeof(fn), "%s_loss.table", base.c_str()); else snprintf(fn, sizeof(fn), "%s_%lld_loss.table", base.c_str(), (long long)la[(size_t)j]);
        say(1, "Exporting cross-validation loss to `%s'... ", fn);
        if (klr_save_crossvalidation_loss(fn, ltr.data() + j * kfold, lte.data() + j * kfold, kfold) != KLR_OK) die("failed");
        say(1, "done\n");
      }
      klr_estimator_destroy(est);
      klr_mat_destroy(mat);
      if (base_mat) klr_mat_destroy(base_mat);
      klr_ctx_destroy(ctx);
      return 0;
    }
    if (ensemble > 1) {
      if (cfg.cooccurrence) die("--ensemble-size together with --co-occurrence is not part of this build");
      if (klr_estimator_estimate_ensemble(est, mat, 0, n, ensemble, seed, la.data(), (int64_t)la.size(), scale.empty() ? nullptr : scale.data(),
                                          offset.empty() ? nullptr : offset.data()) != KLR_OK)
        die("%s", klr_estimator_error(est));
    } else if (klr_estimator_estimate(est, mat, la.data(), (int64_t)la.size(), scale.empty() ? nullptr : scale.data(),
                                      offset.empty() ? nullptr : offset.data()) != KLR_OK)
      die("%s", klr_estimator_error(est));
    if (cfg.save_trace) {
      const std::string fn = base + ".trace";
      say(1, "Exporting trace to `%s'... ", fn.c_str());
      if (klr_estimator_save_trace(est, fn.c_str()) != KLR_OK) die("failed");
      say(1, "done\n");
    }
    if (cfg.save_path) {
      const std::string fn = base + ".path";
      say(1, "Exporting regularization path to `%s'... ", fn.c_str());
      if (klr_estimator_save_path(est, fn.c_str()) != KLR_OK) die("failed");
      say(1, "done\n");
    }
    const int64_t nc = ensemble > 1 ? (int64_t)la.size() : klr_estimator_n_classifiers(est);
    for (int64_t i = 0; i < nc; i++) {                                    // kmerLr_learn.go:60-66
      char fn[4096];
      if (nc == 1) snprintf(fn, sizeof(fn), "%s.json", base.c_str());
      else snprintf(fn, sizeof(fn), "%s_%lld.json", base.c_str(), (long long)la[(size_t)i]);
      say(1, "Exporting model to `%s'... ", fn);
      if (ensemble > 1) {
        if (klr_estimator_save_ensemble(est, i, ensemble, nc, fn, scores ? "scoresLr" : "kmerLr", names.empty() ? nullptr : names.data(),
                                        summary.c_str()) != KLR_OK) die("failed");
      } else if (save_model(est, i, fn, names) != KLR_OK) die("failed");
      say(1, "done\n");
    }
    klr_estimator_destroy(est);
    klr_mat_destroy(mat);
    if (base_mat) klr_mat_destroy(base_mat);
  } else if (cmd == "loss") {
    if (o.pos.size() != 3) die("usage: kmerlr loss [--lambda=F] MODEL.json FG BG");
    std::ifstream jf(o.pos[0]);
    if (!jf) die("cannot open `%s'", o.pos[0].c_str());
    std::stringstream buf; buf << jf.rdbuf();
    const std::string js = buf.str();
    DataSet d;
    klr_mat *mat = nullptr, *red = nullptr;
    std::vector<double> kindex;
    if (scores) {
      read_scores(o.pos[1], flag(o, "header"), 1, d); read_scores(o.pos[2], flag(o, "header"), 0, d);
      if (klr_mat_from_csr(ctx, (int64_t)d.labels.size(), d.dim, d.rowptr.data(), d.col.data(), d.val.data(), d.labels.data(), 0, &mat) != KLR_OK)
        die("%s", klr_last_error(ctx));
    } else {
      // the counter comes from the model (kmerLr_loss.go:53-57)
      KmerSpec ks; SeqSet sq;
      kmer_spec_from_model(js, ks, kindex);
      read_fasta(o.pos[1], 1, sq); read_fasta(o.pos[2], 0, sq);
      mat = count_kmers(ctx, sq, ks);
    }
    std::vector<std::vector<double>> members; std::string summary;
    red = model_data(ctx, mat, js, members, summary, scores ? nullptr : &kindex);
    const double cw[2] = {1.0, 1.0};
    std::vector<double> lv;                                             // {Kmer,Scores}LrEnsemble.Loss: summary of the members' losses
    for (auto& th : members) {
      double out = 0.0;
      if (klr_loss(ctx, red, th.data(), cw, strtod(opt(o, "lambda", "NaN").c_str(), nullptr), &out) != KLR_OK) die("%s", klr_last_error(ctx));
      lv.push_back(out);
    }
    printf("%15e\n", summarize(summary, lv));                            // saveLoss, kmerLr_loss.go:48
    klr_mat_destroy(red); klr_mat_destroy(mat);
  } else if (cmd == "predict") {
    // predict_ / predict_scores_ (kmerLr_predict.go:125-139, scoresLr_predict.go:28-37): log sigma(x . theta) per row
    if (o.pos.size() != 2 && o.pos.size() != 3) die("usage: kmerlr predict MODEL.json DATA [RESULT.table]");
    std::ifstream jf(o.pos[0]);
    if (!jf) die("cannot open `%s'", o.pos[0].c_str());
    std::stringstream buf; buf << jf.rdbuf();
    const std::string js = buf.str();
    DataSet d;
    klr_mat *mat = nullptr, *red = nullptr;
    std::vector<double> kindex;
    int64_t n = 0;
    if (scores) {
      read_scores(o.pos[1], flag(o, "header"), 0, d);
      n = (int64_t)d.labels.size();
      if (klr_mat_from_csr(ctx, n, d.dim, d.rowptr.data(), d.col.data(), d.val.data(), d.labels.data(), 0, &mat) != KLR_OK)
        die("%s", klr_last_error(ctx));
    } else {
      KmerSpec ks; SeqSet sq;
      kmer_spec_from_model(js, ks, kindex);
      read_fasta(o.pos[1], 0, sq);
      n = (int64_t)sq.labels.size();
      mat = count_kmers(ctx, sq, ks);
    }
    std::vector<std::vector<double>> members; std::string summary;
    red = model_data(ctx, mat, js, members, summary, scores ? nullptr : &kindex);
    std::vector<double> out((size_t)std::max<int64_t>(n, 1));
    {
      // {Kmer,Scores}LrEnsemble.Predict (kmerLr_classifier_ensemble.go:124-137): summary of the members' LogPdf per row
      std::vector<std::vector<double>> lp(members.size(), std::vector<double>((size_t)std::max<int64_t>(n, 1)));
      for (size_t k = 0; k < members.size(); k++)
        if (klr_log_pdf(ctx, red, members[k].data(), lp[k].data()) != KLR_OK) die("%s", klr_last_error(ctx));
      std::vector<double> t(members.size());
      for (int64_t i = 0; i < n; i++) { for (size_t k = 0; k < members.size(); k++) t[k] = lp[k][(size_t)i]; out[(size_t)i] = summarize(summary, t); }
    }
    FILE* f = o.pos.size() == 3 ? fopen(o.pos[2].c_str(), "w") : stdout;
    if (!f) die("cannot write `%s'", o.pos[2].c_str());
    fprintf(f, "%15s\n", "prediction");                                  // savePredictions, kmerLr_predict.go:81-84
    for (int64_t i = 0; i < n; i++) fprintf(f, "%15e\n", out[(size_t)i]);
    if (f != stdout) fclose(f);
    klr_mat_destroy(red); klr_mat_destroy(mat);
  } else {
    die("unknown command `%s' (this build provides learn, loss and predict)", cmd.c_str());
  }
  klr_ctx_destroy(ctx);
  return 0;
}
