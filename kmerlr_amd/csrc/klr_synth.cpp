// klr_synth.cpp -- deterministic synthetic k-mer count matrix (bench / tests), host side, threaded.
// Specification: DESIGN.md "synthetic workload" (SURVEY.md section 8(d)).  The oracle holds an
// independent restatement (oracle/klr_synth.c); tests require bit-identical output.
//
//   h(x)       = splitmix64 output function applied to x + 0x9E3779B97F4A7C15
//   motif ids  = first S distinct h(seed + 1 + k) mod m
//   row i      : fg iff i < n_total/2; draw t: c = h(seed ^ i*GOLD ^ t) mod m; with probability
//                0.02 (second hash) replaced by a motif of the row's class, chosen with integer
//                weight S + k (third hash)
//   row layout : bias (0, 1.0), then column+1 ascending with multiplicities
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

#include "klr_internal.h"

static inline uint64_t sm64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

extern "C" int klr_synth_csr(uint64_t seed, int64_t row0, int64_t n_local, int64_t n_total, int64_t m,
                             int64_t draws, int64_t S, int64_t* rowptr, int64_t* col, double* val,
                             uint8_t* labels, int threads) {
  if (n_local < 0 || m < 1 || draws < 0 || S < 0 || !rowptr || !labels || (n_local > 0 && (!col || !val)))
    return klr_fail(nullptr, KLR_EINVAL, "klr_synth_csr: invalid argument");
  // motif table + per-class cumulative integer weights
  std::vector<int64_t> motif;
  for (uint64_t k = 0; (int64_t)motif.size() < S; k++) {
    const int64_t id = (int64_t)(sm64(seed + 1 + k) % (uint64_t)m);
    if (std::find(motif.begin(), motif.end(), id) == motif.end()) motif.push_back(id);
  }
  std::vector<uint64_t> cum[2];
  std::vector<int64_t> mid[2];
  for (int64_t k = 0; k < S; k++) {
    const int cls = (int)(k & 1);     // 0: fg motifs (even k), 1: bg motifs
    cum[cls].push_back((cum[cls].empty() ? 0 : cum[cls].back()) + (uint64_t)(S + k));
    mid[cls].push_back(motif[(size_t)k]);
  }
  const int64_t stride = draws + 1;
  int T = threads > 0 ? threads : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
  if ((int64_t)T > n_local) T = (int)std::max<int64_t>(1, n_local);
  std::vector<int64_t> rowlen((size_t)std::max<int64_t>(n_local, 1));
  // pass 1: every row generated into its own slot of stride draws+1 at the END region of the
  // caller's buffers is not possible without aliasing, so rows are built in thread-local scratch and
  // lengths recorded; pass 2 regenerates and writes at the final offsets.
  auto gen_row = [&](int64_t i, std::vector<uint64_t>& u) {
    const int64_t row = row0 + i;
    const int cls = row < n_total / 2 ? 0 : 1;
    const uint64_t Tw = cum[cls].empty() ? 0 : cum[cls].back();
    u.resize((size_t)draws);
    for (int64_t t = 0; t < draws; t++) {
      const uint64_t h1 = sm64(seed ^ ((uint64_t)row * 0x9E3779B97F4A7C15ull) ^ (uint64_t)t);
      uint64_t c = h1 % (uint64_t)m;
      const uint64_t h2 = sm64(h1 ^ 0xA5A5A5A5A5A5A5A5ull);
      if ((h2 >> 32) < 85899345ull && Tw > 0) {
        const uint64_t r = sm64(h2) % Tw;
        const size_t q = (size_t)(std::upper_bound(cum[cls].begin(), cum[cls].end(), r) - cum[cls].begin());
        c = (uint64_t)mid[cls][q];
      }
      u[(size_t)t] = c;
    }
    std::sort(u.begin(), u.end());
  };
  auto run = [&](int pass) {
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
      std::vector<uint64_t> u;
      const int64_t r0 = n_local * t / T, r1 = n_local * (t + 1) / T;
      for (int64_t i = r0; i < r1; i++) {
        gen_row(i, u);
        if (pass == 0) {
          int64_t q = 1;
          for (int64_t a = 0; a < draws; a++) if (a == 0 || u[(size_t)a] != u[(size_t)a - 1]) q++;
          rowlen[(size_t)i] = q;
          labels[i] = (row0 + i) < n_total / 2;
        } else {
          int64_t o = rowptr[i];
          col[o] = 0; val[o] = 1.0; o++;
          for (int64_t a = 0; a < draws;) {
            int64_t e = a + 1;
            while (e < draws && u[(size_t)e] == u[(size_t)a]) e++;
            col[o] = (int64_t)u[(size_t)a] + 1; val[o] = (double)(e - a); o++;
            a = e;
          }
        }
      }
    });
    for (auto& x : th) x.join();
  };
  (void)stride;
  run(0);
  rowptr[0] = 0;
  for (int64_t i = 0; i < n_local; i++) rowptr[i + 1] = rowptr[i] + rowlen[(size_t)i];
  run(1);
  return KLR_OK;
}
