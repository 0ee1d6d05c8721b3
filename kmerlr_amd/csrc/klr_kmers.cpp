// klr_kmers.cpp -- k-mer equivalence classes on the host: the counterpart of gonetics' KmerEquivalenceRelation /
// KmerClassList as the reference uses them (NewKmerCounter(M, N, complement, reverse, revcomp, maxAmbiguous, alphabet),
// kmerLr_learn.go:94; KmerLrFeatures.ImportConfig / ExportConfig, kmerLr_features.go:164-250).  gonetics itself
// (github.com/pbenner/gonetics v0.0.0-20200513132454-40fc6f7ffc3c) is not in the reference tree; the semantics here
// are pinned by the only known answers the reference holds, TestKmers1 (kmerLr_test.go:30-66: M = 4, N = 8, revcomp,
// gapped alphabet; classes gntanc|gntanc = 3, gntcaa|ttganc = 0, aaagaaa|tttcttt = 1, aagannt|anntctt = 7,
// aacgcgna|tncgcgtt = 1, tgaatgca|tgcattca = 1 on sequence 0 of kmerLr_test.fa):
//
//   * a class is the orbit of a k-mer under the enabled operations (complement, reverse, reverse complement); its
//     count in a sequence is the number of (window, member) matches: a window counts once for every DISTINCT member
//     it matches (a k-mer equal to its own reverse complement is one member and counts once per window);
//   * gapped alphabet: a k-mer may hold the gap letter 'n' at inner positions (never first or last); 'n' matches
//     any nucleotide; max-ambiguous bounds the number of gaps per k-mer (-1: no bound; one value for all lengths or
//     one per length M..N);
//   * iupac alphabet (no fixture in the reference: unpinned): the 15 codes a c g t r y s w k m b d h v n, a code matching
//     its nucleotide set; every code beyond a c g t is "ambiguous" in the sense of max-ambiguous and, like the gap
//     letter, never first or last; complements r<->y, k<->m, b<->v, d<->h, s, w, n fixed; k-mers up to 6 letters;
//   * class name: the members joined by '|', the canonical member first -- "kmer|revcomp" with --revcomp, both
//     printed even when equal.  Canonical member and the class order inside a length are by code (a < c < g < t < n,
//     first letter most significant); lengths ascend.  gonetics' own enumeration order cannot be checked offline:
//     column ORDER parity is unpinned, names and counts are pinned by the vectors above.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "klr_kmers.h"
#include <cstdio>

static const char LETTERS5[] = "acgtn";
// iupac: a c g t | r = ag, y = ct, s = cg, w = at, k = gt, m = ac | b = cgt, d = agt, h = act, v = acg | n
static const char LETTERS15[] = "acgtryswkmbdhvn";
static const int COMP15[15] = {3, 2, 1, 0, 5, 4, 6, 7, 9, 8, 13, 12, 11, 10, 14};
static inline const char* letters_of(int base) { return base == 15 ? LETTERS15 : LETTERS5; }

struct klr_kmer_classes {
  klr_kmer_spec spec;
  std::vector<int> max_amb;                 // per length M..N (-1: no bound)
  int base = 4;                             // letters per position: 4, or 5 with the gap letter
  std::vector<uint64_t> map_off;            // per length: offset into map
  std::vector<uint32_t> map;                // code -> 0-based class id over all lengths, KLR_KMER_INVALID if not a k-mer of the alphabet
  std::vector<uint32_t> rep_code;           // class -> canonical code
  std::vector<uint8_t> rep_len;             // class -> length
  std::string err;
};

static uint64_t ipow(uint64_t b, int e) { uint64_t r = 1; while (e-- > 0) r *= b; return r; }

static void decode(uint64_t code, int k, int base, int* d) {       // d[0] = first letter
  for (int i = k - 1; i >= 0; i--) { d[i] = (int)(code % (uint64_t)base); code /= (uint64_t)base; }
}
static uint64_t encode(const int* d, int k, int base) {
  uint64_t c = 0;
  for (int i = 0; i < k; i++) c = c * (uint64_t)base + (uint64_t)d[i];
  return c;
}
static inline int comp_digit(int x, int base) { return base == 15 ? COMP15[x] : (x == 4 ? 4 : 3 - x); }

// members of the orbit of d under the enabled operations, in the order the name prints them:
// the k-mer, its complement, its reverse, its reverse complement
static int orbit(const klr_kmer_spec& s, const int* d, int k, int base, uint64_t* out, bool* enabled) {
  bool c = s.complement != 0, r = s.reverse != 0, rc = s.revcomp != 0;
  if ((c && r) || (c && rc) || (r && rc)) c = r = rc = true;        // any two generate the third
  enabled[0] = true; enabled[1] = c; enabled[2] = r; enabled[3] = rc;
  int t[KLR_KMER_MAX_LEN];
  out[0] = encode(d, k, base);
  for (int i = 0; i < k; i++) t[i] = comp_digit(d[i], base);
  out[1] = encode(t, k, base);
  for (int i = 0; i < k; i++) t[i] = d[k - 1 - i];
  out[2] = encode(t, k, base);
  for (int i = 0; i < k; i++) t[i] = comp_digit(d[k - 1 - i], base);
  out[3] = encode(t, k, base);
  return 4;
}

extern "C" int klr_kmer_classes_create(const klr_kmer_spec* spec, klr_kmer_classes** out, char* errbuf, int64_t errcap) {
  auto fail = [&](const char* msg) { if (errbuf && errcap > 0) snprintf(errbuf, (size_t)errcap, "%s", msg); return 1; };
  if (!spec || !out) return fail("klr_kmer_classes_create: NULL argument");
  *out = nullptr;
  if (spec->M < 1 || spec->N < spec->M) return fail("invalid k-mer lengths");
  if (spec->alphabet != KLR_ALPHABET_NUCLEOTIDE && spec->alphabet != KLR_ALPHABET_GAPPED && spec->alphabet != KLR_ALPHABET_IUPAC)
    return fail("invalid alphabet");
  const int base = spec->alphabet == KLR_ALPHABET_IUPAC ? 15 : (spec->alphabet == KLR_ALPHABET_GAPPED ? 5 : 4);
  if (base == 15 && spec->N > 6) return fail("iupac k-mers longer than 6 are not part of this build (15^k codes per length)");
  if (spec->N > (base == 5 ? 10 : KLR_KMER_MAX_LEN)) return fail(base == 5 ? "gapped k-mers longer than 10 are not part of this build" : "k-mers longer than 12 are not part of this build");
  if (spec->n_max_ambiguous != 0 && spec->n_max_ambiguous != 1 && spec->n_max_ambiguous != spec->N - spec->M + 1)
    return fail("max-ambiguous: give one value or one per k-mer length");
  klr_kmer_classes* kc = new klr_kmer_classes();
  kc->spec = *spec;
  kc->spec.max_ambiguous = nullptr;
  kc->base = base;
  for (int k = spec->M; k <= spec->N; k++) {
    int a = -1;
    if (spec->n_max_ambiguous == 1) a = spec->max_ambiguous[0];
    else if (spec->n_max_ambiguous > 1) a = spec->max_ambiguous[k - spec->M];
    kc->max_amb.push_back(a);
  }
  uint64_t tot = 0;
  for (int k = spec->M; k <= spec->N; k++) { kc->map_off.push_back(tot); tot += ipow((uint64_t)base, k); }
  kc->map_off.push_back(tot);
  kc->map.assign((size_t)tot, KLR_KMER_INVALID);
  for (int k = spec->M; k <= spec->N; k++) {
    const uint64_t nc = ipow((uint64_t)base, k);
    uint32_t* map = kc->map.data() + kc->map_off[(size_t)(k - spec->M)];
    const int amax = kc->max_amb[(size_t)(k - spec->M)];
    int d[KLR_KMER_MAX_LEN];
    for (uint64_t c = 0; c < nc; c++) {
      decode(c, k, base, d);
      int gaps = 0;                                                                  // ambiguous letters: the gap letter, or any iupac code
      for (int i = 0; i < k; i++) gaps += d[i] >= 4;
      if (d[0] >= 4 || d[k - 1] >= 4 || (amax >= 0 && gaps > amax)) continue;      // not a k-mer of this alphabet
      uint64_t o[4]; bool en[4];
      orbit(kc->spec, d, k, base, o, en);
      uint64_t rep = c;
      for (int j = 1; j < 4; j++) if (en[j]) rep = std::min(rep, o[j]);
      if (rep == c) {                                                                // codes ascend: the canonical member comes first
        map[c] = (uint32_t)kc->rep_code.size();
        kc->rep_code.push_back((uint32_t)c);
        kc->rep_len.push_back((uint8_t)k);
      } else map[c] = map[rep];
    }
  }
  *out = kc;
  return 0;
}

extern "C" void klr_kmer_classes_destroy(klr_kmer_classes* kc) { delete kc; }

extern "C" int64_t klr_kmer_classes_count(const klr_kmer_classes* kc) { return kc ? (int64_t)kc->rep_code.size() : 0; }

extern "C" const klr_kmer_spec* klr_kmer_classes_spec(const klr_kmer_classes* kc) { return kc ? &kc->spec : nullptr; }

extern "C" int klr_kmer_classes_max_ambiguous(const klr_kmer_classes* kc, int k) {
  if (!kc || k < kc->spec.M || k > kc->spec.N) return -1;
  return kc->max_amb[(size_t)(k - kc->spec.M)];
}

extern "C" int klr_kmer_classes_tables(const klr_kmer_classes* kc, const uint32_t** map, const uint64_t** map_off, int* base) {
  if (!kc) return 1;
  if (map) *map = kc->map.data();
  if (map_off) *map_off = kc->map_off.data();
  if (base) *base = kc->base;
  return 0;
}

extern "C" int klr_kmer_class_name(const klr_kmer_classes* kc, int64_t cls, char* buf, int64_t cap) {
  if (!kc || !buf || cap < 1 || cls < 0 || cls >= (int64_t)kc->rep_code.size()) return 1;
  const int k = kc->rep_len[(size_t)cls];
  int d[KLR_KMER_MAX_LEN];
  decode(kc->rep_code[(size_t)cls], k, kc->base, d);
  uint64_t o[4]; bool en[4];
  orbit(kc->spec, d, k, kc->base, o, en);
  std::string s;
  for (int j = 0; j < 4; j++) {
    if (!en[j]) continue;
    int t[KLR_KMER_MAX_LEN];
    decode(o[j], k, kc->base, t);
    if (!s.empty()) s += '|';
    for (int i = 0; i < k; i++) s += letters_of(kc->base)[t[i]];
  }
  if ((int64_t)s.size() + 1 > cap) return 1;
  memcpy(buf, s.c_str(), s.size() + 1);
  return 0;
}

// the class of a k-mer given as text (any member; upper or lower case; text after the first '|' is ignored, as
// ImportConfig does: rel.EquivalenceClass(strings.Split(str, "|")[0]), kmerLr_features.go:207-209)
extern "C" int64_t klr_kmer_class_find(const klr_kmer_classes* kc, const char* kmer) {
  if (!kc || !kmer) return -1;
  int d[KLR_KMER_MAX_LEN];
  int k = 0;
  for (const char* p = kmer; *p && *p != '|'; p++) {
    if (k >= KLR_KMER_MAX_LEN) return -1;
    const char* L = letters_of(kc->base);
    const char* q = strchr(L, *p | 0x20);
    if (!q || !*q) return -1;
    const int v = (int)(q - L);
    if (v >= kc->base) return -1;
    d[k++] = v;
  }
  if (k < kc->spec.M || k > kc->spec.N) return -1;
  const uint32_t c = kc->map[(size_t)(kc->map_off[(size_t)(k - kc->spec.M)] + encode(d, k, kc->base))];
  return c == KLR_KMER_INVALID ? -1 : (int64_t)c;
}
