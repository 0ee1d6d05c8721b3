/*
 * klr_synth.c -- CPU restatement of the deterministic synthetic k-mer count generator
 * (DESIGN.md "synthetic workload").  TEST INFRASTRUCTURE ONLY.
 *
 * The product has its own implementation (kmerlr_amd/csrc/klr_synth.cpp); the tests
 * require the two to agree bit for bit.
 *
 * Spec (all arithmetic in uint64, wrap-around):
 *   sm(x)      = splitmix64 finaliser of x + 0x9E3779B97F4A7C15
 *   motif ids  = first S distinct values of sm(seed + 1 + k) mod m, k = 0,1,2,...
 *   motif k    : class fg if k even, bg if k odd; integer weight W_k = S + k
 *   row i      : label fg iff i < n_total/2; for t = 0..draws-1
 *                  h1 = sm(seed ^ (i * 0x9E3779B97F4A7C15) ^ t);  u = h1 mod m
 *                  h2 = sm(h1 ^ 0xA5A5A5A5A5A5A5A5);  replaced iff (h2 >> 32) < 85899345  (p = 0.02)
 *                  if replaced and S > 0: h3 = sm(h2); r = h3 mod T_class;
 *                     u = id of the first same-class motif whose cumulative weight exceeds r
 *                stored columns = u + 1 ascending, value = multiplicity; column 0 = bias = 1.0
 */
#include "klr_oracle.h"

#include <stdlib.h>
#include <string.h>

static int cmp_u64(const void* a, const void* b) {
  const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return (x > y) - (x < y);
}

/* motif_ids: out, length S */
void klo_synth_motifs(uint64_t seed, int64_t m, int64_t S, int64_t* motif_ids) {
  int64_t found = 0;
  for (uint64_t k = 0; found < S; k++) {
    const int64_t id = (int64_t)(klo_splitmix64(seed + 1 + k) % (uint64_t)m);
    int dup = 0;
    for (int64_t q = 0; q < found; q++) if (motif_ids[q] == id) { dup = 1; break; }
    if (!dup) motif_ids[found++] = id;
  }
}

/* one row; cols/vals need room for draws+1 entries; returns the number written */
int64_t klo_synth_row(uint64_t seed, int64_t row, int64_t n_total, int64_t m, int64_t draws,
                      int64_t S, const int64_t* motif_ids, int64_t* cols, double* vals) {
  uint64_t* u = (uint64_t*)malloc((size_t)(draws > 0 ? draws : 1) * sizeof(uint64_t));
  const int fg = row < n_total / 2;
  const int par = fg ? 0 : 1;
  /* total weight of the row's class */
  uint64_t T = 0;
  for (int64_t k = par; k < S; k += 2) T += (uint64_t)(S + k);
  for (int64_t t = 0; t < draws; t++) {
    const uint64_t h1 = klo_splitmix64(seed ^ ((uint64_t)row * 0x9E3779B97F4A7C15ULL) ^ (uint64_t)t);
    uint64_t c = h1 % (uint64_t)m;
    const uint64_t h2 = klo_splitmix64(h1 ^ 0xA5A5A5A5A5A5A5A5ULL);
    if ((h2 >> 32) < 85899345ULL && T > 0) {
      const uint64_t h3 = klo_splitmix64(h2);
      const uint64_t r = h3 % T;
      uint64_t cum = 0;
      for (int64_t k = par; k < S; k += 2) {
        cum += (uint64_t)(S + k);
        if (cum > r) { c = (uint64_t)motif_ids[k]; break; }
      }
    }
    u[t] = c;
  }
  qsort(u, (size_t)draws, sizeof(uint64_t), cmp_u64);
  int64_t q = 0;
  cols[q] = 0; vals[q] = 1.0; q++;
  for (int64_t t = 0; t < draws;) {
    int64_t e = t + 1;
    while (e < draws && u[e] == u[t]) e++;
    cols[q] = (int64_t)u[t] + 1; vals[q] = (double)(e - t); q++;
    t = e;
  }
  free(u);
  return q;
}

/* rows [row0, row0+n_local) as CSR in the reference layout.  rowptr has n_local+1 entries;
 * col/val need n_local*(draws+1) entries of room.  labels[i] = 1 for fg. */
void klo_synth_csr(uint64_t seed, int64_t row0, int64_t n_local, int64_t n_total, int64_t m,
                   int64_t draws, int64_t S, int64_t* rowptr, int64_t* col, double* val,
                   uint8_t* labels) {
  int64_t* motifs = (int64_t*)malloc((size_t)(S > 0 ? S : 1) * sizeof(int64_t));
  klo_synth_motifs(seed, m, S, motifs);
  rowptr[0] = 0;
  for (int64_t i = 0; i < n_local; i++) {
    const int64_t q = klo_synth_row(seed, row0 + i, n_total, m, draws, S, motifs,
                                    col + rowptr[i], val + rowptr[i]);
    rowptr[i + 1] = rowptr[i] + q;
    labels[i] = (row0 + i) < n_total / 2;
  }
  free(motifs);
}
