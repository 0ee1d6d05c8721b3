// kmerlr -- command line of the MI355X build: `learn`, `loss` and `predict` for --type=kmerLr|scoresLr, mirroring the
// reference CLI for the hot path (kmerLr.go:79-175, kmerLr_learn.go:112-338, scoresLr_learn.go:106-286,
// kmerLr_loss.go:32-106).  Host-side ingest (CSV tables, FASTA records) feeds the C ABI of libkmerlr_hip.so; k-mer
// counting, transform fitting and all training arithmetic run on the GPU.  There is no CPU fallback.
//
//   kmerlr [--type=kmerLr|scoresLr] [-v|-vv] [--seed=S] [--device=N] learn [options] <<M> <N>|<MODEL.json>> FG BG BASENAME
//   kmerlr [--type=...] loss [--lambda=F] [--balance] MODEL.json FG BG [RESULT.table]
//   kmerlr [--type=...] predict MODEL.json DATA [RESULT.table]
//
// Several GPUs of one node (north_star: "samples shard naturally across the 8 GPUs of one node"):
//   kmerlr --ranks=N [--transport=rccl|shm] [--devices=a,b,..] learn ...
// The process parses its arguments and then, BEFORE anything touches HIP, forks N ranks.  Rank r reads only its
// contiguous block of the rows of FG followed by BG, builds its shard on its own GPU (rank r -> device r; with
// --transport=shm every rank on --device unless --devices says otherwise) with n_total = all rows, joins the
// collectives -- RCCL over xGMI (the ncclUniqueId travels through a shared page) or the library's host-staged shared
// memory hooks -- and runs the same replicated driver; rank 0 alone writes .json / .trace / .path / .table.
// Options are the reference's (getopt: `--opt=value` and `--opt value`); an unknown option ends the run.
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kmerlr_driver.h"

struct DataSet {                       // ScoresDataSet / KmerDataSet: fg rows first (labels true); rows of THIS rank
  std::vector<int64_t> rowptr{0}, col;
  std::vector<double> val;
  std::vector<uint8_t> labels;
  int64_t dim = 0;
  std::vector<std::string> names;
};

static int g_verbose = 0;
static int g_rank = 0, g_nranks = 1;
static std::string g_seed = "1";

[[noreturn]] static void die(const char* fmt, ...) {
  if (g_nranks > 1) fprintf(stderr, "[rank %d] ", g_rank);
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
  fputc('\n', stderr);
  fflush(stderr);
  _exit(1);                              // a forked rank must not run the parent's exit handlers
}

static const std::chrono::steady_clock::time_point g_t0 = std::chrono::steady_clock::now();
static void say(int level, const char* fmt, ...);
// -vv: seconds since the start of the process at the ends of the phases
static void stamp(const char* what) {
  say(2, "[%.3f s] %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - g_t0).count(), what);
}
static void say(int level, const char* fmt, ...) {   // PrintStderr (kmerLr_io.go:30-34); one voice: rank 0
  if (g_verbose < level || g_rank != 0) return;
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap);
}

// rows [r0, r1) of `n` rows for rank `rank` of `world` (kmerlr_amd/sharding.py: contiguous, balanced)
static void shard_rows(int64_t n, int rank, int world, int64_t* r0, int64_t* r1) {
  *r0 = n * rank / world; *r1 = n * (rank + 1) / world;
}

// reduce_samples (kmerLr_data.go:288-303, scoresLr_data.go:230-245): the first m1 fg and m2 bg samples
static void reduce_samples(int64_t max_samples, int64_t* n1, int64_t* n2) {
  if (max_samples == 0 || *n1 + *n2 <= max_samples) return;
  const int64_t m1 = (int64_t)((double)*n1 / (double)(*n1 + *n2) * (double)max_samples);
  const int64_t m2 = max_samples - m1;
  if (m1 <= 0 || m2 <= 0) die("cannot reduce samples");
  say(1, "Reduced training data from (%lld,%lld) to (%lld,%lld) samples\n", (long long)*n1, (long long)*n2, (long long)m1, (long long)m2);
  *n1 = m1; *n2 = m2;
}

// ---- scores tables: read_scores_table + convert_scores (scoresLr_data.go:56-145) ---------------------
// rows [first, last) of the table go into d (a rank's block); every row is parsed for the checks the reference makes.
// Returns the number of data rows in the file.
// the whole file in memory (tables and FASTA files are read once, in one piece)
static void read_whole(const std::string& fn, std::string& b, const char* failed_prefix) {
  FILE* f = fopen(fn.c_str(), "rb");
  if (!f) die("%scannot open `%s'", failed_prefix, fn.c_str());
  b.clear();
  if (fseek(f, 0, SEEK_END) == 0) { const long sz = ftell(f); if (sz > 0) b.reserve((size_t)sz); }
  rewind(f);
  std::vector<char> chunk((size_t)4 << 20);
  size_t got;
  while ((got = fread(chunk.data(), 1, chunk.size(), f)) > 0) b.append(chunk.data(), got);
  fclose(f);
}

static int64_t read_scores(const std::string& fn, bool header, uint8_t label, DataSet& d, int64_t first = 0, int64_t last = INT64_MAX) {
  say(1, "Reading scores from `%s'... ", fn.c_str());
  std::string buf;
  read_whole(fn, buf, "failed\n");
  bool need_header = header;
  int64_t lineno = 0, row = 0;
  const char* p = buf.data(); const char* const e = p + buf.size();
  std::vector<double> vals;
  while (p < e) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
    const char* le = nl ? nl : e;
    const char* a = p;
    p = nl ? nl + 1 : e;
    lineno++;
    if (le == a) continue;                                             // empty line
    if (need_header) {
      if (d.names.empty()) {                                           // fields as they are; empty ones dropped
        for (const char* q = a; q <= le;) {
          const char* c = (const char*)memchr(q, ',', (size_t)(le - q));
          const char* fe = c ? c : le;
          if (fe > q) d.names.emplace_back(q, (size_t)(fe - q));
          q = fe + 1;
        }
      }
      need_header = false;
      continue;
    }
    // fields between commas, empty ones dropped, the others trimmed and parsed (read_scores_table, scoresLr_data.go:56-100)
    vals.clear();
    for (const char* q = a; q <= le;) {
      const char* c = (const char*)memchr(q, ',', (size_t)(le - q));
      const char* fe = c ? c : le;
      if (fe > q) {
        const char* t0 = q; const char* t1 = fe;
        while (t0 < t1 && (*t0 == ' ' || *t0 == '\t' || *t0 == '\r')) t0++;
        while (t1 > t0 && (t1[-1] == ' ' || t1[-1] == '\t' || t1[-1] == '\r')) t1--;
        char tok[128];
        const size_t len = (size_t)(t1 - t0);
        if (len == 0 || len >= sizeof(tok)) die("failed\nparsing scores failed at line `%lld'", (long long)lineno);
        memcpy(tok, t0, len); tok[len] = 0;
        char* end = nullptr;
        const double v = strtod(tok, &end);
        if (*end) die("failed\nparsing scores failed at line `%lld'", (long long)lineno);
        vals.push_back(v);
      }
      q = fe + 1;
    }
    if (d.dim == 0) d.dim = (int64_t)vals.size() + 1;
    if ((int64_t)vals.size() + 1 != d.dim) die("failed\nError: data has variable number of features");
    if (row >= first && row < last) {
      d.col.push_back(0); d.val.push_back(1.0);                        // bias
      for (size_t j = 0; j < vals.size(); j++) { d.col.push_back((int64_t)j + 1); d.val.push_back(vals[j]); }   // every column stored, zeros included
      d.rowptr.push_back((int64_t)d.col.size()); d.labels.push_back(label);
    }
    row++;
  }
  say(1, "done\n");
  return row;
}

static int64_t count_scores_rows(const std::string& fn, bool header) {
  std::string buf;
  read_whole(fn, buf, "");
  int64_t n = 0; bool need_header = header;
  const char* p = buf.data(); const char* const e = p + buf.size();
  while (p < e) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
    const char* le = nl ? nl : e;
    if (le > p) { if (need_header) need_header = false; else n++; }
    p = nl ? nl + 1 : e;
  }
  return n;
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

// import_fasta (kmerLr_data.go:101-123): sequences only; counting happens on the device.  A file is read ONCE, whole, and
// indexed by its header lines (a 100 MB file of 100,000 sequences: 0.1 s; the line-by-line reader with a second pass for the
// record count took most of a second of a one-second `learn`); a rank then takes its block of records out of it.
struct FastaFile {
  std::string fn, buf;
  std::vector<size_t> rec;            // offset of every header line's '>'
};

static void load_fasta(const std::string& fn, FastaFile& F) {
  F.fn = fn; F.buf.clear(); F.rec.clear();
  std::string& b = F.buf;
  read_whole(fn, b, "");
  // header lines: '>' at the start of a line
  const char* p = b.data(); const char* const e = p + b.size();
  while (p < e) {
    if (*p == '>') F.rec.push_back((size_t)(p - b.data()));
    const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
    if (!nl) break;
    p = nl + 1;
  }
}

// records [first, last) of the file -> q (whitespace inside the sequence lines dropped)
static void take_fasta(const FastaFile& F, uint8_t label, SeqSet& q, int64_t first = 0, int64_t last = INT64_MAX) {
  const std::string& b = F.buf;
  const int64_t nrec = (int64_t)F.rec.size();
  for (int64_t r = std::max<int64_t>(first, 0); r < std::min(last, nrec); r++) {
    const char* p = b.data() + F.rec[(size_t)r];
    const char* const e = b.data() + (r + 1 < nrec ? F.rec[(size_t)r + 1] : b.size());
    const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));       // the header line
    p = nl ? nl + 1 : e;
    while (p < e) {
      nl = (const char*)memchr(p, '\n', (size_t)(e - p));
      const char* le = nl ? nl : e;
      const char* a = p;
      while (le > a && isspace((unsigned char)le[-1])) le--;               // "\r\n", trailing blanks
      // no blank or control byte inside the line?  eight bytes per step: haszero-style test for a byte below 0x21
      bool clean = true;
      {
        const char* c = a;
        for (; c + 8 <= le; c += 8) {
          uint64_t w; memcpy(&w, c, 8);
          if ((w - 0x2121212121212121ull) & ~w & 0x8080808080808080ull) { clean = false; break; }
        }
        if (clean) for (; c < le; c++) if ((unsigned char)*c <= ' ') { clean = false; break; }
      }
      if (clean) q.blob.append(a, (size_t)(le - a));
      else for (const char* c = a; c < le; c++) if (!isspace((unsigned char)*c)) q.blob.push_back(*c);
      p = nl ? nl + 1 : e;
    }
    q.off.push_back((int64_t)q.blob.size()); q.labels.push_back(label);
  }
}

// `learn` starts reading its two FASTA files while the device context is being created (0.2 s of runtime start-up)
struct FastaPrefetch { FastaFile fg, bg; std::thread t1, t2; bool on = false; };
static FastaPrefetch g_prefetch;

static int64_t read_fasta(const std::string& fn, uint8_t label, SeqSet& q, int64_t first = 0, int64_t last = INT64_MAX) {
  say(1, "Reading sequences from `%s'... ", fn.c_str());
  FastaFile F;
  load_fasta(fn, F);
  take_fasta(F, label, q, first, last);
  say(1, "done\n");
  return (int64_t)F.rec.size();
}

// scan_sequences + convert_counts on the device (klr_mat_from_sequences)
static klr_mat* count_kmers(klr_ctx* ctx, const SeqSet& q, const KmerSpec& ks, int64_t n_total = 0) {
  if (q.labels.empty()) die(g_nranks > 1 ? "Error: this rank holds no rows (fewer samples than ranks?)" : "Error: no training data given");
  say(1, "Counting k-mers on the device... ");
  klr_mat* mat = nullptr;
  const std::string& blob = q.blob.empty() ? std::string("N") : q.blob;
  if (klr_mat_from_sequences_classes(ctx, blob.data(), q.off.data(), (int64_t)q.labels.size(), ks.kc, ks.binarize ? 1 : 0,
                                     q.labels.data(), n_total, &mat) != KLR_OK)
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

// -v with several ranks: the full-width sweeps by the collective they posted
static void comm_report(klr_ctx* ctx) {
  if (g_nranks <= 1 || g_rank != 0) return;
  int64_t ar = 0, rs = 0, fb = 0, rg = 0;
  klr_ctx_comm_stats(ctx, &ar, &rs, &fb, nullptr);
  klr_ctx_comm_ranged(ctx, &rg);
  say(1, "Sweep collectives: %lld all-reduce, %lld reduce-scatter + all-gather (%lld repeated as all-reduce, %lld in strip ranges)\n",
      (long long)ar, (long long)rs, (long long)fb, (long long)rg);
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
  std::vector<double> sc, of;
  if (!scale.empty()) sc.push_back(scale[0]);
  if (!offset.empty()) of.push_back(offset[0]);
  for (auto& pr : order) {
    ecols.push_back(pr.first);
    if (!scale.empty()) sc.push_back(scale[pr.second + 1]);
    if (!offset.empty()) of.push_back(offset[pr.second + 1]);
  }
  // every member's coefficients in the order of the selected pair columns
  for (auto& mt : members) {
    if (mt.size() != nf + 1) die("invalid config file");
    std::vector<double> th{mt[0]};
    for (auto& pr : order) th.push_back(mt[pr.second + 1]);
    mt = th;
  }
  if (klr_mat_select_columns(ctx, ex, ecols.data(), (int64_t)ecols.size(), sc.empty() ? nullptr : sc.data(), of.empty() ? nullptr : of.data(), &red) != KLR_OK)
    die("%s", klr_last_error(ctx));
  klr_mat_destroy(ex); klr_mat_destroy(base);
  return red;
}

// ---- options: the reference's getopt tables (kmerLr.go:85-90, kmerLr_learn.go:116-149, scoresLr_learn.go:109-136,
// kmerLr_loss.go:71-73, scoresLr_loss.go:47-50, scoresLr_predict.go:48-49) plus this build's launcher options ----------
struct OptSpec { const char* name; bool takes_value; };
static const OptSpec GLOBAL_OPTS[] = {
  {"type", true}, {"threads", true}, {"seed", true}, {"help", false}, {"verbose", false}, {"version", false},
  {"device", true}, {"devices", true}, {"ranks", true}, {"transport", true}, {nullptr, false}};
static const OptSpec LEARN_OPTS[] = {
  {"lambda", true}, {"lambda-auto", true}, {"max-features", true}, {"balance", false}, {"co-occurrence", false},
  {"co-preselection", true}, {"ensemble-size", true}, {"ensemble-summary", true}, {"max-epochs", true},
  {"max-iterations", true}, {"max-samples", true}, {"epsilon", true}, {"epsilon-lambda", true}, {"epsilon-loss", true},
  {"save-trace", false}, {"save-path", false}, {"eval-loss", false}, {"data-transform", true}, {"k-fold-cv", true},
  {"validation-size", true}, {"scale-step-size", true}, {"penalty-free", false}, {"adaptive-step-size", false},
  {"threads-cv", true}, {"threads-saga", true}, {"threads-lr", true}, {"help", false}, {"seed", true},
  // kmerLr only
  {"alphabet", true}, {"binarize", false}, {"complement", false}, {"max-ambiguous", true}, {"reverse", false}, {"revcomp", false},
  // scoresLr only
  {"header", false}, {nullptr, false}};
static const OptSpec LOSS_OPTS[] = {{"balance", false}, {"lambda", true}, {"header", false}, {"help", false}, {nullptr, false}};
static const OptSpec PREDICT_OPTS[] = {{"header", false}, {"help", false}, {nullptr, false}};
static const char* KMER_ONLY[] = {"alphabet", "binarize", "complement", "max-ambiguous", "reverse", "revcomp", nullptr};
static const char* SCORES_ONLY[] = {"header", nullptr};

// <command> --help: the options of the reference's tables with what they do HERE
static void print_help(const std::string& cmd, bool scores) {
  if (cmd == "learn") {
    printf("usage: kmerlr %slearn [options] %s\n\n", scores ? "--type=scoresLr " : "",
           scores ? "[MODEL.json] FG.table BG.table BASENAME" : "<<M> <N>|<MODEL.json>> FG.fa BG.fa BASENAME");
    if (!scores) printf(
      "  --alphabet=NAME            nucleotide (default), gapped-nucleotide or iupac-nucleotide\n"
      "  --binarize                 count a k-mer at most once per sequence\n"
      "  --complement, --reverse, --revcomp\n"
      "                             count a k-mer together with its complement / reverse / reverse complement\n"
      "  --max-ambiguous=N[,N..]    most ambiguous positions of a k-mer: one number, or one per k-mer length M..N\n");
    else printf("  --header                   the tables start with a line of feature names\n");
    printf(
      "  --lambda-auto=N[,N..]      features to select; one classifier per value, each starting from the one before\n"
      "  --lambda=F                 fixed regularisation strength instead (--max-features=N caps the selection)\n"
      "  --balance                  class weights n / (2 n_class)\n"
      "  --co-occurrence            pair features x_i x_j; --co-preselection=Q builds them on Q pre-selected features\n"
      "  --ensemble-size=E          E members, member k trained without the k-th part of the rows;\n"
      "  --ensemble-summary=S       their predictions summarised by mean (default), product, min or max\n"
      "  --k-fold-cv=K              K-fold cross-validation: BASENAME_cv<f>*.json, BASENAME*.table, BASENAME*_loss.table\n"
      "  --validation-size=F        fraction of the rows set aside to choose among the --lambda-auto classifiers\n"
      "  --data-transform=T         none (default), standardizer, variance-scaler, max-abs-scaler, mean-scaler\n"
      "  --max-samples=N            use the first N samples (split evenly over the two classes)\n"
      "  --max-epochs=N, --max-iterations=N, --epsilon=F, --epsilon-lambda=F, --epsilon-loss=F (1e-8), --eval-loss\n"
      "                             stopping rules of the leapfrog loop and of the solver\n"
      "  --penalty-free             re-estimate the selected coefficients without the penalty\n"
      "  --save-trace, --save-path  write BASENAME.trace / BASENAME.path\n"
      "  --threads-cv=T             folds that run at once on the device (one rank; default: up to 4)\n"
      "  --threads-saga, --threads-lr, --scale-step-size, --adaptive-step-size\n"
      "                             accepted for the reference's command lines; its SAGA solver is not used here\n"
      "global: --type=kmerLr|scoresLr, -v, --seed=S, --threads=T (ensemble members at once), --device=N,\n"
      "        --ranks=N [--transport=rccl|shm] [--devices=a,b,..]: one process per GPU over row shards\n");
  } else if (cmd == "loss") {
    printf("usage: kmerlr %sloss [--lambda=F] [--balance]%s MODEL.json FG BG [RESULT.table]\n"
           "  the model's loss on the data: mean negative log-likelihood (+ lambda |theta|_1), class-weighted with --balance\n",
           scores ? "--type=scoresLr " : "", scores ? " [--header]" : "");
  } else {
    printf("usage: kmerlr %spredict%s MODEL.json DATA [RESULT.table]\n"
           "  log sigma(x . theta) per row (ensembles: the summary of the members)\n", scores ? "--type=scoresLr " : "", scores ? " [--header]" : "");
  }
}

struct Options { std::map<std::string, std::string> kv; std::vector<std::string> pos; };
static const OptSpec* find_opt(const OptSpec* table, const std::string& name) {
  for (const OptSpec* o = table; o->name; o++) if (name == o->name) return o;
  return nullptr;
}
// getopt as the reference uses it: `--name=value`, `--name value`, `-v` / `-vv` / `-h`, `--` ends the options; an
// option the table does not hold ends the run (the reference prints its usage and exits).  stop_at_positional: the
// global options end at the command word.
static Options parse(int argc, char** argv, int from, const OptSpec* table, bool stop_at_positional, int* next) {
  Options o;
  int i = from;
  bool only_pos = false;
  for (; i < argc; i++) {
    std::string a = argv[i];
    if (only_pos) { o.pos.push_back(a); continue; }
    if (a == "--") { only_pos = true; continue; }
    if (a.rfind("--", 0) == 0) {
      const size_t eq = a.find('=');
      const std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
      const OptSpec* sp = find_opt(table, name);
      if (!sp) die("unknown option `--%s'", name.c_str());
      if (!sp->takes_value) {
        if (eq != std::string::npos) die("option `--%s' takes no value", name.c_str());
        if (name == "verbose") g_verbose++;
        o.kv[name] = "true";
      } else if (eq != std::string::npos) o.kv[name] = a.substr(eq + 1);
      else {
        if (i + 1 >= argc) die("option `--%s' needs a value", name.c_str());
        o.kv[name] = argv[++i];
      }
    } else if (a.size() > 1 && a[0] == '-' && !isdigit((unsigned char)a[1]) && a[1] != '.') {
      for (size_t k = 1; k < a.size(); k++) {
        if (a[k] == 'v') g_verbose++;
        else if (a[k] == 'h') o.kv["help"] = "true";
        else die("unknown option `-%c'", a[k]);
      }
    } else {
      if (stop_at_positional) break;
      o.pos.push_back(a);
    }
  }
  if (next) *next = i;
  return o;
}
static std::string opt(const Options& o, const char* k, const char* def) { auto it = o.kv.find(k); return it == o.kv.end() ? def : it->second; }
static bool flag(const Options& o, const char* k) { return o.kv.count(k) != 0; }
static double opt_float(const Options& o, const char* k, const char* def) {           // strconv.ParseFloat + log.Fatal
  const std::string v = opt(o, k, def);
  char* end = nullptr;
  const double x = strtod(v.c_str(), &end);
  if (v.empty() || *end) die("option `--%s': invalid number `%s'", k, v.c_str());
  return x;
}
static long long opt_int(const Options& o, const char* k, const char* def) {
  const std::string v = opt(o, k, def);
  char* end = nullptr;
  const long long x = strtoll(v.c_str(), &end, 10);
  if (v.empty() || *end) die("option `--%s': invalid integer `%s'", k, v.c_str());
  return x;
}

static int transform_kind(const std::string& kind_) {
  std::string kind = kind_;
  for (auto& c : kind) c = (char)tolower((unsigned char)c);
  if (kind.empty() || kind == "none") return 0;
  if (kind == "standardizer") return KLR_TR_STANDARDIZER;
  if (kind == "variance-scaler") return KLR_TR_VARIANCE;
  if (kind == "max-abs-scaler") return KLR_TR_MAXABS;
  if (kind == "mean-scaler") return KLR_TR_MEAN;
  die("invalid data transform");
}

static std::string slurp(const std::string& fn) {
  std::ifstream jf(fn);
  if (!jf) die("cannot open `%s'", fn.c_str());
  std::stringstream buf; buf << jf.rdbuf();
  return buf.str();
}

// ---- the ranks of a multi-GPU run -----------------------------------------------------------------------------
// One page shared by the ranks (mapped before fork): rank 0 publishes the ncclUniqueId there, every rank reports a
// failure there so that nobody waits for a rank that died before it joined the communicator.
struct Boot {
  std::atomic<int> id_ready;
  std::atomic<int> failed;
  char id[128];
};
struct Launch {
  int ranks = 1;
  std::string transport = "rccl";
  std::vector<int> devices;
  Boot* boot = nullptr;
  void* shm = nullptr;
  int64_t shm_bytes = 0;
};

// forks ranks 1 .. N-1 ... and rank 0: the caller stays the parent that only waits (it never touches HIP).  Returns in
// the children (g_rank set); the parent exits here with the ranks' worst status.
static void fork_ranks(Launch& L) {
  L.boot = (Boot*)mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (L.boot == MAP_FAILED) die("mmap failed");
  new (&L.boot->id_ready) std::atomic<int>(0);
  new (&L.boot->failed) std::atomic<int>(0);
  if (L.transport == "shm") {
    L.shm_bytes = klr_shm_region_bytes(L.ranks);
    L.shm = mmap(nullptr, (size_t)L.shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (L.shm == MAP_FAILED) die("mmap of the shared region (%lld bytes) failed", (long long)L.shm_bytes);
    if (klr_shm_region_init(L.shm, L.shm_bytes, L.ranks) != KLR_OK) die("%s", klr_last_error(nullptr));
  }
  fflush(stdout); fflush(stderr);
  std::vector<pid_t> pids;
  for (int r = 0; r < L.ranks; r++) {
    const pid_t pid = fork();
    if (pid < 0) die("fork failed");
    if (pid == 0) { g_rank = r; g_nranks = L.ranks; return; }
    pids.push_back(pid);
  }
  int worst = 0;
  size_t left = pids.size();
  while (left > 0) {
    int st = 0;
    const pid_t p = wait(&st);
    if (p < 0) break;
    left--;
    const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
    if (code != 0) {
      if (worst == 0) worst = code;
      L.boot->failed.store(1);                        // ranks still waiting for the id or at a shm barrier give up
      if (L.shm) klr_shm_region_abort(L.shm);       // ... and so do ranks waiting at a barrier of the shared region
    }
  }
  _exit(worst);
}

static void join_collectives(klr_ctx* ctx, const Launch& L) {
  if (L.ranks <= 1) return;
  if (L.transport == "shm") {
    if (klr_ctx_init_shm(ctx, L.shm, L.shm_bytes, g_rank, L.ranks) != KLR_OK) { L.boot->failed.store(1); die("%s", klr_last_error(ctx)); }
    return;
  }
  char id[128];
  if (g_rank == 0) {
    if (klr_rccl_unique_id(id) != KLR_OK) { L.boot->failed.store(1); die("%s", klr_last_error(nullptr)); }
    memcpy(L.boot->id, id, 128);
    L.boot->id_ready.store(1, std::memory_order_release);
  } else {
    for (long spins = 0; !L.boot->id_ready.load(std::memory_order_acquire); spins++) {
      if (L.boot->failed.load()) die("another rank failed before the communicator was set up");
      if (spins > 600000) die("timed out waiting for rank 0's ncclUniqueId");     // 10 minutes
      usleep(1000);
    }
    memcpy(id, L.boot->id, 128);
  }
  if (klr_ctx_init_rccl(ctx, id, g_rank, L.ranks) != KLR_OK) { L.boot->failed.store(1); die("%s", klr_last_error(ctx)); }
}

int main(int argc, char** argv) {
  int ci = 1;
  Options go = parse(argc, argv, 1, GLOBAL_OPTS, true, &ci);
  const std::string type = opt(go, "type", "kmerLr");
  const int device = (int)opt_int(go, "device", "0");
  g_seed = opt(go, "seed", "1");
  if (opt_int(go, "threads", "1") < 1) die("invalid number of threads `%s'", opt(go, "threads", "1").c_str());
  if (flag(go, "version")) { printf("%s\n", klr_version()); return 0; }
  if (ci >= argc || flag(go, "help")) {
    printf("usage: kmerlr [--type=kmerLr|scoresLr] [-v] [--seed=S] [--device=N] [--ranks=N [--transport=rccl|shm] [--devices=a,b,..]] <learn|loss|predict> ...\n");
    return ci >= argc && !flag(go, "help") ? 1 : 0;
  }
  const std::string cmd = argv[ci];
  const bool scores = type == "scoresLr" || type == "scoreslr";
  if (!scores && type != "kmerLr" && type != "kmerlr") die("invalid classifier type");
  const OptSpec* table = cmd == "learn" ? LEARN_OPTS : cmd == "loss" ? LOSS_OPTS : cmd == "predict" ? PREDICT_OPTS : nullptr;
  if (!table) die("unknown command `%s' (this build provides learn, loss and predict)", cmd.c_str());
  Options o = parse(argc, argv, ci + 1, table, false, nullptr);
  if (flag(o, "help")) { print_help(cmd, scores); return 0; }
  for (const char** k = scores ? KMER_ONLY : SCORES_ONLY; *k; k++)
    if (flag(o, *k)) die("unknown option `--%s' for --type=%s", *k, scores ? "scoresLr" : "kmerLr");
  if (flag(o, "seed")) g_seed = opt(o, "seed", "1");

  Launch L;
  L.ranks = (int)opt_int(go, "ranks", "1");
  L.transport = opt(go, "transport", "rccl");
  if (L.ranks < 1 || L.ranks > 64) die("invalid --ranks");
  if (L.transport != "rccl" && L.transport != "shm") die("invalid --transport `%s' (rccl or shm)", L.transport.c_str());
  { std::stringstream ss(opt(go, "devices", "")); std::string t; while (std::getline(ss, t, ',')) if (!t.empty()) L.devices.push_back(atoi(t.c_str())); }
  if (!L.devices.empty() && (int)L.devices.size() != L.ranks) die("--devices needs one device per rank");
  if (L.ranks > 1 && cmd != "learn") die("--ranks applies to `learn'");
  if (L.ranks > 1) fork_ranks(L);                       // returns in the ranks only; nothing has touched HIP so far
  const int my_device = !L.devices.empty() ? L.devices[(size_t)g_rank] : (L.ranks > 1 && L.transport == "rccl") ? g_rank : device;

  if (cmd == "learn" && !scores && (o.pos.size() == 4 || o.pos.size() == 5)) {
    const size_t k = o.pos.size();
    g_prefetch.on = true;
    g_prefetch.t1 = std::thread([fn = o.pos[k - 3]]() { load_fasta(fn, g_prefetch.fg); });
    g_prefetch.t2 = std::thread([fn = o.pos[k - 2]]() { load_fasta(fn, g_prefetch.bg); });
  }
  klr_ctx* ctx = nullptr;
  if (klr_ctx_create(my_device, &ctx) != KLR_OK) { if (L.boot) L.boot->failed.store(1); die("%s", klr_last_error(nullptr)); }
  stamp("device context ready");
  join_collectives(ctx, L);
  klr_backend be;
  if (klr_hip_backend(ctx, &be) != KLR_OK) die("%s", klr_last_error(ctx));

  if (cmd == "learn") {
    klr_config cfg;
    klr_config_default(&cfg);
    cfg.lambda = opt_float(o, "lambda", "NaN");
    std::vector<int64_t> la;
    { std::stringstream ss(opt(o, "lambda-auto", "0")); std::string t;
      while (std::getline(ss, t, ',')) { char* end = nullptr; la.push_back(strtoll(t.c_str(), &end, 10)); if (t.empty() || *end) die("option `--lambda-auto': invalid integer `%s'", t.c_str()); } }
    if (la.empty()) die("option `--lambda-auto' is empty");
    std::sort(la.begin(), la.end());
    if (!std::isnan(cfg.lambda) && !(la.size() == 1 && la[0] == 0)) die("options --lambda and --lambda-auto are incompatible");
    cfg.max_features = opt_int(o, "max-features", "0");
    cfg.max_epochs = opt_int(o, "max-epochs", "0");
    cfg.max_iterations = opt_int(o, "max-iterations", "0");
    const int64_t max_samples = opt_int(o, "max-samples", "0");
    cfg.epsilon = opt_float(o, "epsilon", "0");
    cfg.epsilon_lambda = opt_float(o, "epsilon-lambda", "0");
    cfg.epsilon_loss = opt_float(o, "epsilon-loss", "1e-8");
    cfg.eval_loss = flag(o, "eval-loss") || cfg.epsilon_loss != 0.0;
    cfg.balance = flag(o, "balance");
    cfg.penalty_free = flag(o, "penalty-free");
    cfg.save_trace = flag(o, "save-trace");
    cfg.save_path = flag(o, "save-path");
    cfg.verbose = g_rank == 0 ? g_verbose : 0;
    cfg.validation_size = opt_float(o, "validation-size", "0.0");
    if (cfg.validation_size < 0.0 || cfg.validation_size > 1.0) die("invalid --validation-size");
    cfg.data_transform = transform_kind(opt(o, "data-transform", ""));
    // SAGA's knobs (step size, its thread pools): accepted like the reference's, without effect on the coordinate-descent solver
    (void)opt_float(o, "scale-step-size", "1.0");
    for (const char* k : {"threads-saga", "threads-lr"}) (void)opt_int(o, k, "1");
    // --threads-cv (kmerLr_learn.go:146, the pool of kmerLr_crossvalidation.go:176): folds that run at once, each in its
    // own host thread on its own context (one rank only; with --ranks > 1 the folds run in turn)
    // Not given: up to four folds at once on one rank (the results do not depend on it; the reference's default of one
    // fold at a time leaves the card idle between the small kernels of a fold's solves: 117 against 58 ms for five folds).
    cfg.threads_cv = o.kv.count("threads-cv") ? (int)opt_int(o, "threads-cv", "1") : 4;
    cfg.threads = go.kv.count("threads") ? (int)opt_int(go, "threads", "1") : 4;   // config.Pool: the members of an ensemble (:81), likewise
    if (cfg.threads_cv < 1) die("invalid --threads-cv");
    if (flag(o, "adaptive-step-size") || flag(o, "scale-step-size")) say(1, "note: the step-size options belong to the reference's SAGA solver and have no effect here\n");
    const int kfold = (int)opt_int(o, "k-fold-cv", "1");
    const long long seed = atoll(g_seed.c_str());
    if (kfold < 1) die("invalid --k-fold-cv");
    const int ensemble = (int)opt_int(o, "ensemble-size", "1");
    const std::string summary = opt(o, "ensemble-summary", "mean");
    if (ensemble < 1) die("invalid --ensemble-size");
    if (summary != "mean" && summary != "product" && summary != "min" && summary != "max" && !summary.empty()) die("invalid --ensemble-summary");
    const bool cv_path = kfold > 1 || cfg.validation_size > 0.0;
    cfg.ensemble_size = ensemble;                                        // read by the cross-validation (members inside every fold)
    cfg.ensemble_summary = summary == "product" ? KLR_SUMMARY_PRODUCT : summary == "min" ? KLR_SUMMARY_MIN : summary == "max" ? KLR_SUMMARY_MAX : KLR_SUMMARY_MEAN;
    cfg.cooccurrence = flag(o, "co-occurrence") ? 1 : 0;
    cfg.copreselection = opt_int(o, "co-preselection", "0");
    if (cfg.copreselection != 0 && !cfg.cooccurrence) die("--co-preselection needs --co-occurrence");

    // positionals: <M> <N> | <MODEL.json>, FG, BG, BASENAME (kmerLr_learn.go:153-189; scoresLr: [MODEL.json] FG BG BASENAME)
    DataSet d;
    SeqSet sq;
    std::string base, model_fn, fn_fg, fn_bg;
    KmerSpec ks;
    if (scores) {
      if (o.pos.size() != 3 && o.pos.size() != 4) die("usage: kmerlr --type=scoresLr learn [options] [MODEL.json] FG.table BG.table BASENAME");
      const size_t k = o.pos.size() - 3;
      if (k) model_fn = o.pos[0];
      fn_fg = o.pos[k]; fn_bg = o.pos[k + 1]; base = o.pos[k + 2];
    } else {
      if (o.pos.size() != 4 && o.pos.size() != 5) die("usage: kmerlr learn [options] <<M> <N>|<MODEL.json>> FG.fa BG.fa BASENAME");
      if (o.pos.size() == 5) {
        char *e1 = nullptr, *e2 = nullptr;
        ks.M = (int)strtol(o.pos[0].c_str(), &e1, 10); ks.N = (int)strtol(o.pos[1].c_str(), &e2, 10);
        if (*e1 || *e2 || ks.M < 1 || ks.N < ks.M) die("invalid k-mer lengths");
        fn_fg = o.pos[2]; fn_bg = o.pos[3]; base = o.pos[4];
      } else { model_fn = o.pos[0]; fn_fg = o.pos[1]; fn_bg = o.pos[2]; base = o.pos[3]; }
    }
    // warm start (learn, kmerLr_learn.go:89-92; NewKmerLrEstimator :45-67): the model's counter / feature names and the
    // coefficients of its first member; the k-mer set is NOT fixed and the transform is fitted anew on the data
    std::vector<int64_t> warm_cols;
    std::vector<double> warm_theta;
    if (!model_fn.empty()) {
      const std::string js = slurp(model_fn);
      if (json_bool(js, "Cooccurrence", false, false)) die("warm start from a co-occurrence model is not part of this build");
      const size_t dpos = js.find("\"Distributions\"");
      if (dpos == std::string::npos) die("invalid config file");
      warm_theta = json_numbers_after(js, "Parameters", js.find("vector:logistic regression", dpos), nullptr);
      std::vector<double> index;
      if (scores) index = json_numbers_after(js, "Index", 0, nullptr);
      else kmer_spec_from_model(js, ks, index);                   // M, N, flags, alphabet, Binarize: the model's, not the command line's
      if (warm_theta.size() != index.size() + 1) die("invalid config file");
      std::vector<std::pair<int64_t, double>> ct;
      for (size_t k = 0; k < index.size(); k++) ct.emplace_back((int64_t)index[k] + 1, warm_theta[k + 1]);
      std::sort(ct.begin(), ct.end());
      warm_cols.push_back(0);
      const double t0 = warm_theta[0];
      warm_theta.assign(1, t0);
      for (auto& pr : ct) { if (pr.first == warm_cols.back()) die("invalid config file: a feature is listed twice"); warm_cols.push_back(pr.first); warm_theta.push_back(pr.second); }
    }
    if (!scores) {
      if (model_fn.empty()) {
        ks.revcomp = flag(o, "revcomp"); ks.binarize = flag(o, "binarize");
        ks.reverse = flag(o, "reverse"); ks.complement = flag(o, "complement");
        ks.alphabet = opt(o, "alphabet", "nucleotide");
        parse_max_ambiguous(opt(o, "max-ambiguous", "-1"), ks.M, ks.N, ks.max_ambiguous);
        kmer_prepare(ks, &d.names);
      } else {
        d.names.clear();
        char buf[128];
        for (int64_t c = 0; c < ks.m; c++) { klr_kmer_class_name(ks.kc, c, buf, sizeof(buf)); d.names.push_back(buf); }
      }
    }
    // the rows: all fg samples, then all bg samples (compile_training_data :305-335); --max-samples keeps the first
    // m1 / m2 of them; this rank reads its contiguous block of that order
    FastaFile ffg, fbg;
    if (!scores && g_prefetch.on && g_prefetch.fg.fn == fn_fg && g_prefetch.bg.fn == fn_bg) {
      g_prefetch.t1.join(); g_prefetch.t2.join();
      ffg = std::move(g_prefetch.fg); fbg = std::move(g_prefetch.bg);
      g_prefetch.on = false;
    } else if (!scores) {
      if (g_prefetch.on) { g_prefetch.t1.join(); g_prefetch.t2.join(); g_prefetch.on = false; }
      std::thread t2([&]() { load_fasta(fn_bg, fbg); });
      load_fasta(fn_fg, ffg);
      t2.join();
    }
    int64_t nfg = scores ? count_scores_rows(fn_fg, flag(o, "header")) : (int64_t)ffg.rec.size();
    int64_t nbg = scores ? count_scores_rows(fn_bg, flag(o, "header")) : (int64_t)fbg.rec.size();
    reduce_samples(max_samples, &nfg, &nbg);
    const int64_t n_all = nfg + nbg;
    if (n_all == 0) die("Error: no training data given");
    if (n_all < L.ranks) die("fewer samples (%lld) than ranks (%d)", (long long)n_all, L.ranks);
    int64_t row0 = 0, row1 = n_all;
    shard_rows(n_all, g_rank, g_nranks, &row0, &row1);
    const int64_t f0 = std::min(row0, nfg), f1 = std::min(row1, nfg);            // fg records [f0, f1)
    const int64_t b0 = std::max(row0, nfg) - nfg, b1 = std::max(row1, nfg) - nfg;  // bg records [b0, b1)
    if (scores) {
      read_scores(fn_fg, flag(o, "header"), 1, d, f0, f1);
      read_scores(fn_bg, flag(o, "header"), 0, d, b0, b1);
    } else {
      say(1, "Reading sequences from `%s'... ", fn_fg.c_str());
      take_fasta(ffg, 1, sq, f0, f1);
      say(1, "done\n");
      say(1, "Reading sequences from `%s'... ", fn_bg.c_str());
      take_fasta(fbg, 0, sq, b0, b1);
      say(1, "done\n");
      ffg = FastaFile(); fbg = FastaFile();
      stamp("sequences read");
      d.dim = ks.m + 1; d.labels = sq.labels;
    }
    const int64_t n = (int64_t)d.labels.size();
    if (n != row1 - row0) die("internal error: read %lld rows for the block [%lld, %lld)", (long long)n, (long long)row0, (long long)row1);
    std::vector<uint8_t> labels_all((size_t)n_all, 0);
    for (int64_t i = 0; i < nfg; i++) labels_all[(size_t)i] = 1;
    if (g_nranks > 1) say(1, "Rows [%lld, %lld) of %lld on rank 0; %d ranks over %s\n", (long long)row0, (long long)row1, (long long)n_all, g_nranks, L.transport.c_str());
    for (int64_t c : warm_cols) if (c >= d.dim) die("invalid config file: feature %lld outside the data", (long long)(c - 1));
    klr_mat* mat = nullptr;
    if (scores) {
      if (klr_mat_from_csr(ctx, n, d.dim, d.rowptr.data(), d.col.data(), d.val.data(), d.labels.data(), n_all, &mat) != KLR_OK)
        die("%s", klr_last_error(ctx));
    } else {
      mat = count_kmers(ctx, sq, ks, n_all);
      stamp("k-mer counts resident");
    }
    // co-occurrence mode ({Kmer,Scores}LrEstimatorEnsemble.Estimate, kmerLr_estimator_ensemble.go:98-117): optional
    // pre-selection with the transform fitted on the base data, then the pair matrix takes the place of the data
    klr_mat* base_mat = nullptr;
    std::vector<int64_t> base_cols;
    int64_t nb1 = 0;
    // with folds / a validation set the pre-selection belongs to the fold (its training rows only): the cross-validation
    // gets the base data and builds every fold's pair matrix itself (klr_estimator_crossvalidate)
    const bool fold_copre = cfg.cooccurrence && cv_path && cfg.copreselection > 0 && cfg.copreselection < d.dim - 1;
    if (cfg.cooccurrence && !warm_cols.empty()) die("warm start together with --co-occurrence is not part of this build");
    if (cfg.cooccurrence && !fold_copre) {
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
    auto pair_of = [&](int64_t j, int64_t* f1_, int64_t* f2_) {
      int64_t k1, k2;
      klr_coeff_sub2ind(nb1 - 1, j - 1, &k1, &k2);
      *f1_ = base_cols[(size_t)k1 + 1] - 1; *f2_ = base_cols[(size_t)k2 + 1] - 1;
    };
    auto save_model = [&](klr_estimator* e, int64_t i, const char* fn, const std::vector<const char*>& nm) {
      const char* tn = scores ? "scoresLr" : "kmerLr";
      if (!cfg.cooccurrence) return klr_estimator_save_model(e, i, fn, tn, nm.empty() ? nullptr : nm.data());
      int64_t c1 = 0; const int64_t* cc = nullptr;
      klr_estimator_classifier(e, i, &c1, &cc, nullptr, nullptr, nullptr, nullptr);
      std::vector<int64_t> pairs;
      for (int64_t k = 1; k < c1; k++) { int64_t f1_, f2_; pair_of(cc[k], &f1_, &f2_); pairs.push_back(f1_); pairs.push_back(f2_); }
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
    if (!warm_cols.empty() && klr_estimator_set_model(est, warm_cols.data(), warm_theta.data(), (int64_t)warm_cols.size()) != KLR_OK)
      die("%s", klr_estimator_error(est));
    std::vector<const char*> names;
    for (auto& s : d.names) names.push_back(s.c_str());
    const bool writer = g_rank == 0;                                   // one rank writes the result files
    auto export_trace_path = [&](klr_estimator* e, const std::string& stem) {   // learn_parameters :45-55
      if (!writer) return;
      if (cfg.save_trace) {
        const std::string fn = stem + ".trace";
        say(1, "Exporting trace to `%s'... ", fn.c_str());
        if (klr_estimator_save_trace(e, fn.c_str()) != KLR_OK) die("failed");
        say(1, "done\n");
      }
      if (cfg.save_path) {
        const std::string fn = stem + ".path";
        say(1, "Exporting regularization path to `%s'... ", fn.c_str());
        if (klr_estimator_save_path(e, fn.c_str()) != KLR_OK) die("failed");
        say(1, "done\n");
      }
    };
    if (cv_path) {
      // learn_cv (kmerLr_learn.go:71-87): k folds (one pass without folds when only a validation set is split off),
      // per-fold models, joined prediction / loss tables
      const int64_t nla = (int64_t)la.size();
      const int nfold = kfold > 1 ? kfold : 1;
      const int64_t ncol = cfg.validation_size > 0.0 ? 1 : nla;          // classifiers kept per fold
      std::vector<double> pred((size_t)(nla * std::max<int64_t>(n, 1))), ltr((size_t)(nla * nfold)), lte((size_t)(nla * nfold));
      std::vector<int32_t> groups((size_t)std::max<int64_t>(n, 1));
      if (klr_estimator_crossvalidate(est, mat, row0, n_all, kfold, seed, la.data(), nla, scale.empty() ? nullptr : scale.data(),
                                      offset.empty() ? nullptr : offset.data(), pred.data(), groups.data(), ltr.data(), lte.data()) != KLR_OK)
        die("%s", klr_estimator_error(est));
      char fn[4096];
      for (int f = 0; f < nfold; f++) {                                                     // learn_parameters :42-66
        const std::string stem = kfold > 1 ? base + "_cv" + std::to_string(f + 1) : base;
        klr_estimator* fe = klr_estimator_fold(est, f);
        if (fe) export_trace_path(fe, stem);
        for (int64_t j = 0; j < ncol && writer; j++) {
          const std::string mfn = ncol == 1 ? stem + ".json" : stem + "_" + std::to_string((long long)la[(size_t)j]) + ".json";
          say(1, "Exporting model to `%s'... ", mfn.c_str());
          if (ensemble > 1) {
            // the fold's estimator holds its members' classifiers member-major ({Kmer,Scores}LrEnsemble of the fold)
            if (!fe) die("failed");
            if (cfg.cooccurrence) {
              if (fold_copre) {
                const int64_t* bc = nullptr; int64_t nbc = 0;
                if (klr_estimator_base_cols(fe, &bc, &nbc) != KLR_OK || nbc < 1) die("failed");
                base_cols.assign(bc, bc + nbc); nb1 = nbc;
              }
              if (klr_estimator_save_ensemble_pairs(fe, j, ensemble, ncol, mfn.c_str(), scores ? "scoresLr" : "kmerLr", names.empty() ? nullptr : names.data(),
                                                    summary.c_str(), base_cols.data(), nb1) != KLR_OK) die("failed");
            } else if (klr_estimator_save_ensemble(fe, j, ensemble, ncol, mfn.c_str(), scores ? "scoresLr" : "kmerLr", names.empty() ? nullptr : names.data(),
                                                   summary.c_str()) != KLR_OK) die("failed");
          } else {
            if (fold_copre) {                                            // the pairs are those of the fold's own pre-selection
              const int64_t* bc = nullptr; int64_t nbc = 0;
              if (!fe || klr_estimator_base_cols(fe, &bc, &nbc) != KLR_OK || nbc < 1) die("failed");
              base_cols.assign(bc, bc + nbc); nb1 = nbc;
            }
            if (save_model(est, f * ncol + j, mfn.c_str(), names) != KLR_OK) die("failed");
          }
          say(1, "done\n");
        }
      }
      if (kfold > 1) {
        // every rank holds the predictions of its own rows: summed over the ranks into the whole table (zeros elsewhere)
        std::vector<int32_t> groups_all((size_t)n_all);
        klr_cv_groups_val(n_all, kfold, cfg.validation_size, seed, 0, n_all, groups_all.data(), nullptr);
        for (int64_t j = 0; j < ncol; j++) {                                                  // learn_cv :78-86
          std::vector<double> pa((size_t)n_all, 0.0);
          for (int64_t i = 0; i < n; i++) pa[(size_t)(row0 + i)] = pred[(size_t)(j * n + i)];
          if (klr_allreduce_host(ctx, pa.data(), n_all, KLR_F64) != KLR_OK) die("%s", klr_last_error(ctx));
          if (!writer) continue;
          if (ncol == 1) snprintf(fn, sizeof(fn), "%s.table", base.c_str()); else snprintf(fn, sizeof(fn), "%s_%lld.table", base.c_str(), (long long)la[(size_t)j]);
          say(1, "Exporting cross-validation results to `%s'... ", fn);
          if (klr_save_crossvalidation(fn, pa.data(), labels_all.data(), groups_all.data(), n_all, kfold) != KLR_OK) die("failed");
          say(1, "done\n");
          if (ncol == 1) snprintf(fn, sizeof(fn), "%s_loss.table", base.c_str()); else snprintf(fn, sizeof(fn), "%s_%lld_loss.table", base.c_str(), (long long)la[(size_t)j]);
          say(1, "Exporting cross-validation loss to `%s'... ", fn);
          if (klr_save_crossvalidation_loss(fn, ltr.data() + j * kfold, lte.data() + j * kfold, kfold) != KLR_OK) die("failed");
          say(1, "done\n");
        }
      }
      klr_estimator_destroy(est);
      klr_mat_destroy(mat);
      if (base_mat) klr_mat_destroy(base_mat);
      comm_report(ctx);
      klr_ctx_destroy(ctx);
      fflush(stdout); fflush(stderr);
      _exit(0);
    }
    if (ensemble > 1) {
      if (klr_estimator_estimate_ensemble(est, mat, row0, n_all, ensemble, seed, la.data(), (int64_t)la.size(), scale.empty() ? nullptr : scale.data(),
                                          offset.empty() ? nullptr : offset.data()) != KLR_OK)
        die("%s", klr_estimator_error(est));
    } else if (klr_estimator_estimate(est, mat, la.data(), (int64_t)la.size(), scale.empty() ? nullptr : scale.data(),
                                      offset.empty() ? nullptr : offset.data()) != KLR_OK)
      die("%s", klr_estimator_error(est));
    stamp("estimated");
    export_trace_path(est, base);
    const int64_t nc = ensemble > 1 ? (int64_t)la.size() : klr_estimator_n_classifiers(est);
    for (int64_t i = 0; i < nc && writer; i++) {                          // kmerLr_learn.go:60-66
      char fn[4096];
      if (nc == 1) snprintf(fn, sizeof(fn), "%s.json", base.c_str());
      else snprintf(fn, sizeof(fn), "%s_%lld.json", base.c_str(), (long long)la[(size_t)i]);
      say(1, "Exporting model to `%s'... ", fn);
      if (ensemble > 1) {
        if (cfg.cooccurrence) {
          if (klr_estimator_save_ensemble_pairs(est, i, ensemble, nc, fn, scores ? "scoresLr" : "kmerLr", names.empty() ? nullptr : names.data(),
                                                summary.c_str(), base_cols.data(), nb1) != KLR_OK) die("failed");
        } else if (klr_estimator_save_ensemble(est, i, ensemble, nc, fn, scores ? "scoresLr" : "kmerLr", names.empty() ? nullptr : names.data(),
                                               summary.c_str()) != KLR_OK) die("failed");
      } else if (save_model(est, i, fn, names) != KLR_OK) die("failed");
      say(1, "done\n");
    }
    klr_estimator_destroy(est);
    klr_mat_destroy(mat);
    if (base_mat) klr_mat_destroy(base_mat);
  } else if (cmd == "loss") {
    if (o.pos.size() != 3 && o.pos.size() != 4) die("usage: kmerlr loss [--lambda=F] [--balance] MODEL.json FG BG [RESULT.table]");
    const std::string js = slurp(o.pos[0]);
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
    double cw[2] = {1.0, 1.0};                                          // {Kmer,Scores}LrEnsemble.Loss :108-116
    if (flag(o, "balance") && klr_class_weights(ctx, red, cw) != KLR_OK) die("%s", klr_last_error(ctx));
    std::vector<double> lv;                                             // summary of the members' losses
    const double lambda = opt_float(o, "lambda", "0.0");
    for (auto& th : members) {
      double out = 0.0;
      if (klr_loss(ctx, red, th.data(), cw, lambda, &out) != KLR_OK) die("%s", klr_last_error(ctx));
      lv.push_back(out);
    }
    FILE* f = o.pos.size() == 4 ? fopen(o.pos[3].c_str(), "w") : stdout;
    if (!f) die("cannot write `%s'", o.pos[3].c_str());
    fprintf(f, "%15e\n", summarize(summary, lv));                       // saveLoss, kmerLr_loss.go:48
    if (f != stdout) fclose(f);
    klr_mat_destroy(red); klr_mat_destroy(mat);
  } else if (cmd == "predict") {
    // predict_ / predict_scores_ (kmerLr_predict.go:125-139, scoresLr_predict.go:28-37): log sigma(x . theta) per row
    if (o.pos.size() != 2 && o.pos.size() != 3) die("usage: kmerlr predict MODEL.json DATA [RESULT.table]");
    const std::string js = slurp(o.pos[0]);
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
  }
  comm_report(ctx);
  klr_ctx_destroy(ctx);
  stamp("done");
  fflush(stdout); fflush(stderr);
  _exit(0);          // every file is written and closed; the runtime's own teardown is 0.1 s of nothing
}
