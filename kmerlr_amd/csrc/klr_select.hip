// klr_select.hip -- featureSelector.Select on the device (kmerLr_feature_selection.go:78-134,
// computeLambda :179-192, NLargestAbsFloat64 kmerLr_sort.go:120-131).
//
// The reference sorts all m gradient entries by |g| (O(m log m) on one core) to read the first
// 2N.  Here: keys = bit pattern of |g| (monotone for non-negative doubles); an 8-pass radix
// SELECT finds the 2N-th largest key; an ordered compaction gathers the candidates (ties in
// ascending column order); one workgroup sorts the <= 2N candidates by (|g| desc, column asc);
// one max-reduction finds the largest |g| strictly below the N-th.  The host then runs the two
// short pick loops of :95-116 over the 2N candidates.  Everything is integer/compare work, so the
// result is bit-exact with the oracle for identical g.
#include "klr_internal.h"

#include <algorithm>
#include <cmath>

struct SelState {
  u64 prefix;      // bits decided so far (high to low)
  u32 kremain;     // rank still to resolve inside the current prefix bucket
  u32 pad;
  u64 thresh;      // final: key of the K-th largest
  u32 n_gt;        // keys  > thresh
  u32 n_eq_take;   // keys == thresh to take (lowest column ids)
  u64 vkey;        // key of the N-th largest (set by the sort kernel)
  u64 wkey;        // largest key < vkey
};

// abs_key (klr_internal.h): NaN is treated as a zero gradient (never preferred); see DESIGN.md

__global__ __launch_bounds__(256) void k_sel_keys(const double* __restrict__ g, i64 m, u64* __restrict__ keys) {
  const i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
  if (j < m) keys[j] = abs_key(g[j + 1]);      // g[0] is the bias
}

__global__ __launch_bounds__(256) void k_sel_hist(const u64* __restrict__ keys, i64 m, const SelState* __restrict__ st,
                                                  int shift, u32* __restrict__ hist) {
  __shared__ u32 sh[256];
  sh[threadIdx.x] = 0;
  __syncthreads();
  const u64 prefix = st->prefix;
  const u64 himask = shift >= 56 ? 0ull : (~0ull << (shift + 8));
  for (i64 j = (i64)blockIdx.x * 256 + threadIdx.x; j < m; j += (i64)gridDim.x * 256) {
    const u64 k = keys[j];
    if ((k & himask) == (prefix & himask)) atomicAdd(&sh[(k >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}

__global__ void k_sel_pick(SelState* st, int shift, u32* hist) {
  // one thread: walk the 256 bins from the top
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  u32 k = st->kremain, cum = 0;
  int d = 255;
  for (; d > 0; d--) {
    if (cum + hist[d] >= k) break;
    cum += hist[d];
  }
  st->prefix |= ((u64)d) << shift;
  st->kremain = k - cum;
  for (int i = 0; i < 256; i++) hist[i] = 0;
  if (shift == 0) {
    st->thresh = st->prefix;
    st->n_eq_take = st->kremain;
  }
}

// per-block counts of (key > T) and (key == T)
#define CMP_BLOCK 1024
__global__ __launch_bounds__(CMP_BLOCK) void k_sel_count(const u64* __restrict__ keys, i64 m, const SelState* __restrict__ st,
                                                        u32* __restrict__ bgt, u32* __restrict__ beq) {
  __shared__ u32 sg[CMP_BLOCK / 64], se[CMP_BLOCK / 64];
  const i64 j = (i64)blockIdx.x * CMP_BLOCK + threadIdx.x;
  const u64 T = st->thresh;
  const u64 k = j < m ? keys[j] : 0ull;
  const bool gt = j < m && k > T, eq = j < m && k == T;
  const u32 cg = (u32)__popcll(__ballot(gt)), ce = (u32)__popcll(__ballot(eq));
  if ((threadIdx.x & 63) == 0) { sg[threadIdx.x >> 6] = cg; se[threadIdx.x >> 6] = ce; }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 a = 0, b = 0;
    for (int w = 0; w < CMP_BLOCK / 64; w++) { a += sg[w]; b += se[w]; }
    bgt[blockIdx.x] = a; beq[blockIdx.x] = b;
  }
}

// single block: exclusive scans of the per-block counts (in place) + totals
__global__ __launch_bounds__(1024) void k_sel_scan(u32* bgt, u32* beq, i64 nb, SelState* st) {
  __shared__ u32 sa[1024], sb[1024];
  __shared__ u32 ca, cb;
  if (threadIdx.x == 0) { ca = 0; cb = 0; }
  __syncthreads();
  for (i64 base = 0; base < nb; base += 1024) {
    const i64 i = base + threadIdx.x;
    const u32 va = i < nb ? bgt[i] : 0u, vb = i < nb ? beq[i] : 0u;
    sa[threadIdx.x] = va; sb[threadIdx.x] = vb;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const u32 ta = (int)threadIdx.x >= o ? sa[threadIdx.x - o] : 0u;
      const u32 tb = (int)threadIdx.x >= o ? sb[threadIdx.x - o] : 0u;
      __syncthreads();
      sa[threadIdx.x] += ta; sb[threadIdx.x] += tb;
      __syncthreads();
    }
    if (i < nb) { bgt[i] = ca + sa[threadIdx.x] - va; beq[i] = cb + sb[threadIdx.x] - vb; }
    __syncthreads();
    if (threadIdx.x == 1023) { ca += sa[1023]; cb += sb[1023]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) st->n_gt = ca;
}


__global__ __launch_bounds__(CMP_BLOCK) void k_sel_compact(const u64* __restrict__ keys, i64 m, const SelState* __restrict__ st,
                                                          const u32* __restrict__ bgt, const u32* __restrict__ beq,
                                                          Cand* __restrict__ cand) {
  __shared__ u32 sg[CMP_BLOCK / 64], se[CMP_BLOCK / 64];
  const i64 j = (i64)blockIdx.x * CMP_BLOCK + threadIdx.x;
  const u64 T = st->thresh;
  const u64 k = j < m ? keys[j] : 0ull;
  const bool gt = j < m && k > T, eq = j < m && k == T;
  const unsigned long long mg = __ballot(gt), me = __ballot(eq);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sg[w] = (u32)__popcll(mg); se[w] = (u32)__popcll(me); }
  __syncthreads();
  u32 og = bgt[blockIdx.x], oe = beq[blockIdx.x];
  for (int q = 0; q < w; q++) { og += sg[q]; oe += se[q]; }
  const unsigned long long below = (1ull << lane) - 1ull;
  if (gt) {
    const u32 pos = og + (u32)__popcll(mg & below);
    cand[pos].key = k; cand[pos].idx = (u32)j; cand[pos].pad = 0;
  } else if (eq) {
    const u32 r = oe + (u32)__popcll(me & below);
    if (r < st->n_eq_take) {
      const u32 pos = st->n_gt + r;
      cand[pos].key = k; cand[pos].idx = (u32)j; cand[pos].pad = 0;
    }
  }
}

__device__ __forceinline__ bool cand_before(const Cand& a, const Cand& b) {
  // larger |g| first, then lower column id
  return a.key > b.key || (a.key == b.key && a.idx < b.idx);
}

// one workgroup bitonic sort of P = 2^k candidates held in global memory (L2-resident, <= 2N)
__global__ __launch_bounds__(1024) void k_sel_sort(Cand* __restrict__ c, u32 K, u32 P, u32 Nth, SelState* st) {
  for (u32 i = K + threadIdx.x; i < P; i += 1024) { c[i].key = 0ull; c[i].idx = 0xFFFFFFFFu; c[i].pad = 0; }
  __syncthreads();
  for (u32 size = 2; size <= P; size <<= 1) {
    for (u32 stride = size >> 1; stride > 0; stride >>= 1) {
      for (u32 t = threadIdx.x; t < P / 2; t += 1024) {
        const u32 lo = 2 * t - (t & (stride - 1));
        const u32 hi = lo + stride;
        const bool up = (lo & size) == 0;     // ascending block in "before" order
        Cand a = c[lo], b = c[hi];
        const bool swap = up ? cand_before(b, a) : cand_before(a, b);
        if (swap) { c[lo] = b; c[hi] = a; }
      }
      __threadfence_block();
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) { st->vkey = (Nth >= 1 && Nth <= K) ? c[Nth - 1].key : 0ull; st->wkey = 0ull; }
}

__global__ __launch_bounds__(256) void k_sel_wmax(const u64* __restrict__ keys, i64 m, SelState* st) {
  // w = max{ |g_k| : |g_k| < v }   (computeLambda :186-190); integer max -> order independent
  const u64 v = st->vkey;
  u64 best = 0ull;
  for (i64 j = (i64)blockIdx.x * 256 + threadIdx.x; j < m; j += (i64)gridDim.x * 256) {
    const u64 k = keys[j];
    if (k < v && k > best) best = k;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const u64 t = __shfl_xor(best, o, 64);
    if (t > best) best = t;
  }
  if ((threadIdx.x & 63) == 0 && best) atomicMax((unsigned long long*)&st->wkey, (unsigned long long)best);
}

static inline double key_to_double(u64 k) { double d; memcpy(&d, &k, 8); return d; }
static inline u64 abs_key_host(double v) { v = std::fabs(v); u64 k; memcpy(&k, &v, 8); return (v != v) ? 0ull : k; }

// ---------------------------------------------------------------------------------------------------
// fast path: ONE coarse histogram over the top 13 key bits (exponent + 2 mantissa bits) brackets the
// K-th largest |g|; a second-level histogram (next 12 bits) inside the bracketing bin trims the candidates to
// K + a few, which one workgroup sorts in LDS.  Two launches (k_fast_compact, k_fast_finish) instead of 25.  Falls back to the
// exact 8-pass radix select when more than FAST_CAP candidates remain (e.g. K reaches into the zeros).
// ---------------------------------------------------------------------------------------------------
#define FAST_CAP 4096

struct FastState {
  u64 lowkey;        // keys >= lowkey are candidates
  u32 ncand;         // their number
  u32 cursor;        // compaction cursor
  u32 fallback;      // 1: too many candidates
  u32 wfound;        // 1: wkey valid
  u64 vkey, wkey;
};

__global__ __launch_bounds__(256) void k_hist_from_keys(const u64* __restrict__ keys, i64 m, u32* __restrict__ hist) {
  __shared__ u32 sh[FAST_BINS];
  for (int i = threadIdx.x; i < FAST_BINS; i += 256) sh[i] = 0;
  __syncthreads();
  for (i64 j = (i64)blockIdx.x * 256 + threadIdx.x; j < m; j += (i64)gridDim.x * 256) atomicAdd(&sh[(u32)(keys[j] >> FAST_SHIFT)], 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < FAST_BINS; i += 256) if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

__global__ __launch_bounds__(256) void k_fast_keys_hist(const double* __restrict__ g, i64 m, u64* __restrict__ keys,
                                                        u32* __restrict__ hist) {
  __shared__ u32 sh[FAST_BINS];
  for (int i = threadIdx.x; i < FAST_BINS; i += 256) sh[i] = 0;
  __syncthreads();
  for (i64 j = (i64)blockIdx.x * 256 + threadIdx.x; j < m; j += (i64)gridDim.x * 256) {
    const u64 k = abs_key(g[j + 1]);
    keys[j] = k;
    atomicAdd(&sh[(u32)(k >> FAST_SHIFT)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < FAST_BINS; i += 256) if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

// inclusive suffix sum over the 1024 lanes of a workgroup (sum of v over lanes >= t): wave shuffles, one LDS
// hop for the 16 wave totals, two barriers
__device__ __forceinline__ u32 block_suffix_sum_1024(u32 v, u32* wtot /* 16 */) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  u32 x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const u32 y = __shfl_down(x, o, 64);
    if (lane + o < 64) x += y;
  }
  if (lane == 0) wtot[w] = x;                      // lane 0 holds the wave's total
  __syncthreads();
  u32 add = 0;
  for (int u = w + 1; u < 16; u++) add += wtot[u];
  __syncthreads();                                 // wtot may be reused by the caller
  return x + add;
}

// rank sort of n <= 1024 candidates in LDS by (|g| desc, column asc): (key, column) pairs are distinct, so the ranks
// are a permutation.  The comparisons of an element are split over 1024/n' lanes (n' = n rounded up to a power of
// two), partial ranks meet in LDS.  src, dst, rk: LDS.  Ends with a barrier.
__device__ __forceinline__ void rank_sort_1024(const Cand* src, Cand* dst, u32* rk, u32 n) {
  const u32 t = threadIdx.x;
  u32 np = 64; while (np < n) np <<= 1;            // 64..1024
  const u32 parts = 1024u / np;                    // 16..1
  const u32 e = t & (np - 1), part = t / np;
  if (t < np) rk[t] = 0u;
  __syncthreads();
  if (e < n) {
    const Cand my = src[e];
    const u32 j0 = (u32)((unsigned long long)n * part / parts), j1 = (u32)((unsigned long long)n * (part + 1) / parts);
    u32 r = 0;
    for (u32 j = j0; j < j1; j++) r += cand_before(src[j], my) ? 1u : 0u;
    if (parts == 1) rk[e] = r; else if (r) atomicAdd(&rk[e], r);
  }
  __syncthreads();
  if (t < n) dst[rk[t]] = src[t];
  __syncthreads();
}

// the same for up to FAST_CAP candidates: a thread owns up to four of them (t, t + 1024, ...) and walks the whole list once
// (every lane reads the same LDS word: a broadcast read), counting for each of its own how many come before it.
// n = 1,300: 7 us (the bitonic network it replaces: 66 barrier stages, 65-75 us at 2,048 slots).  Ends with a barrier.
__device__ __forceinline__ void rank_sort_big(const Cand* src, Cand* dst, u32 n) {
  const u32 t = threadIdx.x;
  Cand my[4]; u32 r[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int u = 0; u < 4; u++) { const u32 e = t + 1024u * u; my[u] = e < n ? src[e] : Cand{0ull, 0xFFFFFFFFu, 0u}; }
  const int cnt = (int)((n - 1u) / 1024u) + 1;          // owned slots in use (wave-uniform upper bound)
  for (u32 j = 0; j < n; j++) {
    const Cand c = src[j];
#pragma unroll
    for (int u = 0; u < 4; u++) if (u < cnt) r[u] += cand_before(c, my[u]) ? 1u : 0u;
  }
#pragma unroll
  for (int u = 0; u < 4; u++) { const u32 e = t + 1024u * u; if (e < n) dst[r[u]] = my[u]; }
  __syncthreads();
}

// control block + histograms of the fast path (all zero between calls: k_fast_finish re-zeroes what was used)
#define FINE_BINS 4096
#define FINE_SHIFT 38

// Fast path, first launch (many workgroups).  Every workgroup brackets the K-th largest key in the coarse
// histogram (top 13 key bits), then compacts its slice of the keys at or above the bracketing bin into a global
// candidate list and counts the keys inside the bracketing bin in a second-level histogram (next 12 bits).
// A single workgroup doing this scan is latency bound (m/1024 dependent trips to L2 per lane: 22 us at 65k
// columns); a device-scope fence + ticket to finish in the same launch costs about as much (L2 write-back and
// invalidate), so the sort is a second launch.
__global__ __launch_bounds__(1024) void k_fast_compact(const u32* __restrict__ hist, u32* __restrict__ fine, FastCtl* __restrict__ ctl,
                                                       const u64* __restrict__ keys, i64 m, u32 K, Cand* __restrict__ gcand) {
  __shared__ u32 ssum[1024], wtot[16];
  __shared__ u32 fh[FINE_BINS];
  __shared__ unsigned long long s_low;
  __shared__ u32 s_n, s_above;
  const int t = threadIdx.x;
  // keys of this workgroup's slice first: their latency overlaps the bracket search
  const i64 per = (m + gridDim.x - 1) / gridDim.x;
  const i64 j0 = per * blockIdx.x, j1 = min(m, j0 + per);
  u64 kk[4];
#pragma unroll
  for (int u = 0; u < 4; u++) kk[u] = (j0 + t + u * 1024 < j1) ? keys[j0 + t + u * 1024] : 0ull;
  // thread t owns bins [8t, 8t+8); inclusive suffix scan over the thread sums
  u32 h[8], loc = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { h[i] = hist[t * 8 + i]; loc += h[i]; }
  for (int i = t; i < FINE_BINS; i += 1024) fh[i] = 0u;
  if (t == 0) { s_low = 0ull; s_n = 0; s_above = 0; }
  {
    const u32 suf = block_suffix_sum_1024(loc, wtot);
    ssum[t] = suf;
  }
  __syncthreads();
  {
    const u32 above = (t + 1 < 1024) ? ssum[t + 1] : 0u;
    if (above < K && ssum[t] >= K) {
      u32 cum = above;
      int b = 7;
      for (; b > 0; b--) { if (cum + h[b] >= K) break; cum += h[b]; }
      s_low = ((u64)(t * 8 + b)) << FAST_SHIFT;
      s_above = cum;                               // keys in bins above the bracketing bin (< K)
      s_n = cum + h[b];                            // keys at or above it (>= K)
    }
  }
  __syncthreads();
  const u32 n = s_n;
  const u64 low = s_low;
  if (blockIdx.x == 0 && t == 0) { ctl->n = n; ctl->above = s_above; ctl->low = low; }
  if (n > GCAND_CAP || n == 0) return;             // k_fast_finish reports the fallback
  for (i64 jb = j0; jb < j1; jb += 4096) {
    if (jb > j0) {
#pragma unroll
      for (int u = 0; u < 4; u++) kk[u] = (jb + t + u * 1024 < j1) ? keys[jb + t + u * 1024] : 0ull;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const i64 j = jb + t + u * 1024;
      if (j < j1 && kk[u] >= low) {
        const u32 pos = atomicAdd(&ctl->cursor, 1u);
        gcand[pos].key = kk[u]; gcand[pos].idx = (u32)j; gcand[pos].pad = 0;
        if ((kk[u] >> FAST_SHIFT) == (low >> FAST_SHIFT)) atomicAdd(&fh[(u32)(kk[u] >> FINE_SHIFT) & (FINE_BINS - 1)], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = t; i < FINE_BINS; i += 1024) if (fh[i]) atomicAdd(&fine[i], fh[i]);
}

// Fast path, second launch (one workgroup): trims the candidates with the second-level histogram to K + a few,
// sorts them in LDS by (|g| desc, column asc) -- rank sort up to 1024, bitonic above --, writes the top K,
// v = the N-th key and w = the largest key below v, and re-zeroes the histograms and the control block.
__global__ __launch_bounds__(1024) void k_fast_finish(u32* __restrict__ hist, u32* __restrict__ fine, FastCtl* __restrict__ ctl,
                                                      u32 K, u32 Nth, const Cand* __restrict__ gcand, FastState* st,
                                                      Cand* __restrict__ out, const double* __restrict__ extra_src,
                                                      double* __restrict__ extra_dst) {
  extern __shared__ Cand sc[];                    // FAST_CAP candidates + FAST_CAP sorted
  __shared__ u32 ssum[1024], wtot[16];
  __shared__ unsigned long long wbest;
  __shared__ u32 s_cursor, s_fbin, s_nt;
  const int t = threadIdx.x;
  if (t == 0 && extra_src) *extra_dst = *extra_src;     // a caller scalar (the sweep's loss) rides in the same block
  const u32 n = ctl->n, above_bin = ctl->above;
  const u64 low = ctl->low;
  const bool too_many = n > GCAND_CAP || n == 0;
  // second-level threshold: the largest fine bin f with above + #(fine bins >= f) >= K
  u32 fv[4], floc = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) { fv[i] = fine[t * 4 + i]; floc += fv[i]; }
  if (t == 0) { s_fbin = 0; s_nt = n; s_cursor = 0; wbest = 0ull; }
  {
    const u32 suf = block_suffix_sum_1024(floc, wtot);
    ssum[t] = suf;
  }
  __syncthreads();
  if (!too_many) {
    const u32 need = K - above_bin;                // >= 1 keys wanted from the bracketing bin
    const u32 above = (t + 1 < 1024) ? ssum[t + 1] : 0u;
    if (above < need && ssum[t] >= need) {
      u32 cum = above;
      int b = 3;
      for (; b > 0; b--) { if (cum + fv[b] >= need) break; cum += fv[b]; }
      s_fbin = (u32)(t * 4 + b);
      s_nt = above_bin + cum + fv[b];
    }
  }
  __syncthreads();
  const u32 nt = s_nt;
  const u64 lowfine = low | ((u64)s_fbin << FINE_SHIFT);
  // re-zero what this call used
#pragma unroll
  for (int i = 0; i < 8; i++) hist[t * 8 + i] = 0u;
#pragma unroll
  for (int i = 0; i < 4; i++) fine[t * 4 + i] = 0u;
  __syncthreads();                                 // every lane has read ctl
  if (t < 16) ((u32*)ctl)[t] = 0u;
  if (too_many || nt > FAST_CAP) {                 // the host falls back to the exact radix select
    if (t == 0) { st->lowkey = low; st->ncand = too_many ? n : nt; st->cursor = 0; st->fallback = 1; st->wfound = 0; st->vkey = 0; st->wkey = 0; }
    return;
  }
  for (u32 i = t; i < n; i += 1024) {
    const Cand c = gcand[i];
    if (c.key >= lowfine) { const u32 pos = atomicAdd(&s_cursor, 1u); sc[pos] = c; }
  }
  __syncthreads();
  Cand* sorted = sc;
  if (nt <= 1024) {
    sorted = sc + 1024;
    rank_sort_1024(sc, sorted, ssum, nt);
  } else {
    sorted = sc + FAST_CAP;
    rank_sort_big(sc, sorted, nt);
  }
  for (u32 i = t; i < K && i < nt; i += 1024) out[i] = sorted[i];
  const u64 vkey = (Nth >= 1 && Nth <= nt) ? sorted[Nth - 1].key : 0ull;
  for (u32 i = t; i < nt; i += 1024)
    if (sorted[i].key < vkey && (i == 0 || sorted[i - 1].key >= vkey)) atomicMax(&wbest, (unsigned long long)sorted[i].key | (1ull << 63));
  __syncthreads();
  if (t == 0) {
    st->lowkey = lowfine; st->ncand = nt; st->cursor = nt; st->fallback = 0;
    st->vkey = vkey; st->wfound = (wbest >> 63) ? 1u : 0u; st->wkey = wbest & ~(1ull << 63);
  }
}

// Hinted selection: the candidate list was filled while the keys were produced (SelHint).  One workgroup sorts it;
// if the hint was too high (fewer than K candidates) or far too low (more than FAST_CAP) it reports fallback = 2 and
// leaves the histogram for the regular path.
// Ranks of the hinted candidates over MANY workgroups: the n^2 comparisons of a rank sort are what a one-workgroup sort
// spends its time on (n = 1,300 candidates at K = 1,000: 65 us on one CU, whatever the network).  FAST_RANK_WGS workgroups
// each stage the whole list in LDS and rank their share (32 lanes per candidate, partial counts met by shuffles); a
// candidate lands at sorted_dev[rank].  k_fast_direct then only copies.
#define FAST_RANK_WGS 128
__global__ __launch_bounds__(1024) void k_fast_rank(const FastCtl* __restrict__ ctl, u32 K, const Cand* __restrict__ gcand,
                                                    Cand* __restrict__ sorted_dev) {
  extern __shared__ Cand sc[];
  const u32 n = ctl->cursor;
  if (n < K || n > FAST_CAP) return;                 // k_fast_direct reports the fallback
  const u32 t = threadIdx.x;
  for (u32 i = t; i < n; i += 1024) sc[i] = gcand[i];
  __syncthreads();
  const u32 per = (n + FAST_RANK_WGS - 1) / FAST_RANK_WGS;       // <= 32 candidates per workgroup
  const u32 cl = t >> 5, part = t & 31u;
  const u32 ci = blockIdx.x * per + cl;
  const bool valid = cl < per && ci < n;
  const Cand my = valid ? sc[ci] : Cand{0ull, 0xFFFFFFFFu, 0u};
  u32 r = 0;
  if (valid) for (u32 j = part; j < n; j += 32u) r += cand_before(sc[j], my) ? 1u : 0u;
  r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 4, 64); r += __shfl_xor(r, 8, 64); r += __shfl_xor(r, 16, 64);
  if (valid && part == 0) sorted_dev[r] = my;
}

// k_fast_rank that also delivers: a candidate lands at its rank in PINNED HOST memory (all n of them: the host reads v
// and w off the sorted list), the first workgroup writes the state word and the caller's scalar.  No second launch; the
// cursor is cleared by the row pass of the next sweep (RowPassArgs.zero_u32), the first kernel of every step.
__global__ __launch_bounds__(1024) void k_fast_rank_out(const FastCtl* __restrict__ ctl, u64 tau, u32 K, const Cand* __restrict__ gcand,
                                                        FastState* st, Cand* __restrict__ out, const double* __restrict__ extra_src,
                                                        double* __restrict__ extra_dst) {
  extern __shared__ Cand sc[];
  const u32 n = ctl->cursor;
  const u32 t = threadIdx.x;
  const bool usable = n >= K && n <= FAST_CAP;
  if (blockIdx.x == 0 && t == 0) {
    if (extra_src) *extra_dst = *extra_src;
    st->lowkey = tau; st->ncand = n; st->cursor = usable ? n : 0u; st->fallback = usable ? 0u : 2u; st->wfound = 0; st->vkey = 0; st->wkey = 0;
  }
  if (!usable) return;
  for (u32 i = t; i < n; i += 1024) sc[i] = gcand[i];
  __syncthreads();
  const u32 per = (n + FAST_RANK_WGS - 1) / FAST_RANK_WGS;       // <= 32 candidates per workgroup
  const u32 cl = t >> 5, part = t & 31u;
  const u32 ci = blockIdx.x * per + cl;
  const bool valid = cl < per && ci < n;
  const Cand my = valid ? sc[ci] : Cand{0ull, 0xFFFFFFFFu, 0u};
  u32 r = 0;
  if (valid) for (u32 j = part; j < n; j += 32u) r += cand_before(sc[j], my) ? 1u : 0u;
  r += __shfl_xor(r, 1, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 4, 64); r += __shfl_xor(r, 8, 64); r += __shfl_xor(r, 16, 64);
  if (valid && part == 0) out[r] = my;
}

__global__ __launch_bounds__(1024) void k_fast_direct(u32* __restrict__ /*hist: untouched (not built when hinted)*/, FastCtl* __restrict__ ctl, u64 tau, u32 K, u32 Nth,
                                                      const Cand* __restrict__ sorted_dev, FastState* st, Cand* __restrict__ out,
                                                      const double* __restrict__ extra_src, double* __restrict__ extra_dst) {
  extern __shared__ Cand sc[];                    // the candidates, sorted by k_fast_rank
  __shared__ unsigned long long wbest;
  const int t = threadIdx.x;
  if (t == 0 && extra_src) *extra_dst = *extra_src;
  const u32 n = ctl->cursor;
  if (t == 0) wbest = 0ull;
  __syncthreads();                                 // every lane has read the cursor
  if (t == 0) ctl->cursor = 0u;
  if (n < K || n > FAST_CAP) {
    if (t == 0) { st->lowkey = tau; st->ncand = n; st->cursor = 0; st->fallback = 2; st->wfound = 0; st->vkey = 0; st->wkey = 0; }
    return;
  }
  for (u32 i = t; i < n; i += 1024) sc[i] = sorted_dev[i];
  __syncthreads();
  const Cand* sorted = sc;
  for (u32 i = t; i < K; i += 1024) out[i] = sorted[i];
  const u64 vkey = (Nth >= 1 && Nth <= n) ? sorted[Nth - 1].key : 0ull;
  for (u32 i = t; i < n; i += 1024)
    if (sorted[i].key < vkey && (i == 0 || sorted[i - 1].key >= vkey)) atomicMax(&wbest, (unsigned long long)sorted[i].key | (1ull << 63));
  __syncthreads();
  if (t == 0) {
    st->lowkey = tau; st->ncand = n; st->cursor = n; st->fallback = 0;
    st->vkey = vkey; st->wfound = (wbest >> 63) ? 1u : 0u; st->wkey = wbest & ~(1ull << 63);
  }
}

static int select_exact_device(klr_ctx* ctx, const double* g_dev, u64* keys, bool keys_ready, i64 m, i64 K, i64 N,
                               std::vector<Cand>& hc, u64* vkey, u64* wkey) {
  hipStream_t st = ctx->stream;
  char* misc; Cand* cand;
  const i64 nb = ceil_div(m, CMP_BLOCK);
  u32 P = 1; while ((i64)P < K) P <<= 1; if (P < 2) P = 2;
  const size_t misc_bytes = sizeof(SelState) + 256 * sizeof(u32) + (size_t)nb * 2 * sizeof(u32) + 64;
  KLR_TRY(ws_get(ctx, WS_SEL_MISC, misc_bytes, (void**)&misc));
  KLR_TRY(ws_get(ctx, WS_SEL_CAND, (size_t)P * sizeof(Cand), (void**)&cand));
  SelState* dst = (SelState*)misc;
  u32* hist = (u32*)(misc + sizeof(SelState));
  u32* bgt = hist + 256;
  u32* beq = bgt + nb;
  SelState init; memset(&init, 0, sizeof(init));
  init.kremain = (u32)K;
  KLR_HIP(ctx, hipMemsetAsync(misc, 0, misc_bytes, st));
  KLR_TRY(h2d(ctx, dst, &init, sizeof(init)));
  if (!keys_ready) hipLaunchKernelGGL(k_sel_keys, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, st, g_dev, m, keys);
  const unsigned hgrid = (unsigned)std::min<i64>(ceil_div(m, 256 * 8), 1024);
  for (int shift = 56; shift >= 0; shift -= 8) {
    hipLaunchKernelGGL(k_sel_hist, dim3(std::max(1u, hgrid)), dim3(256), 0, st, keys, m, dst, shift, hist);
    hipLaunchKernelGGL(k_sel_pick, dim3(1), dim3(64), 0, st, dst, shift, hist);
  }
  hipLaunchKernelGGL(k_sel_count, dim3((unsigned)nb), dim3(CMP_BLOCK), 0, st, keys, m, dst, bgt, beq);
  hipLaunchKernelGGL(k_sel_scan, dim3(1), dim3(1024), 0, st, bgt, beq, nb, dst);
  hipLaunchKernelGGL(k_sel_compact, dim3((unsigned)nb), dim3(CMP_BLOCK), 0, st, keys, m, dst, bgt, beq, cand);
  hipLaunchKernelGGL(k_sel_sort, dim3(1), dim3(1024), 0, st, cand, (u32)K, P, (u32)N, dst);
  hipLaunchKernelGGL(k_sel_wmax, dim3(std::max(1u, hgrid)), dim3(256), 0, st, keys, m, dst);
  KLR_HIP(ctx, hipGetLastError());
  SelState hs;
  KLR_HIP(ctx, hipMemcpyAsync(hc.data(), cand, (size_t)K * sizeof(Cand), hipMemcpyDeviceToHost, st));
  KLR_TRY(d2h_sync(ctx, &hs, dst, sizeof(hs)));
  *vkey = hs.vkey; *wkey = hs.wkey;
  return KLR_OK;
}

// buffers of the fast path; the histogram is kept all-zero between uses
int sel_prepare(klr_ctx* ctx, i64 m, u64** keys, u32** hist) {
  char* fm;
  KLR_TRY(ws_get(ctx, WS_SEL_KEYS, (size_t)std::max<i64>(m, 1) * sizeof(u64), (void**)keys));
  // [FastCtl 64 B | coarse histogram | second-level histogram]
  const size_t fm_bytes = sizeof(FastCtl) + (FAST_BINS + FINE_BINS) * sizeof(u32);
  KLR_TRY(ws_get(ctx, WS_MASK, fm_bytes, (void**)&fm));
  *hist = (u32*)(fm + sizeof(FastCtl));
  if (ctx->sel_hist_zeroed != (void*)*hist) {          // k_fast_finish re-zeroes what was used
    KLR_HIP(ctx, hipMemsetAsync(fm, 0, fm_bytes, ctx->stream));
    ctx->sel_hist_zeroed = (void*)*hist;
  }
  return KLR_OK;
}

// Select :95-119 and computeLambda :179-192 on the K = min(m, 2N) candidates sorted by (|g| desc, column asc);
// vkey = key of the N-th, wkey = the largest key below it
static void select_finish_host(const std::vector<Cand>& hc, i64 K, u64 vkey, u64 wkey, uint8_t* b, i64 c_prev, i64 N,
                               double lambda_prev, double eps_lambda, double* lambda_out, i64* c_out, int* ok_out, i64* picked_out) {
  i64 c = c_prev;
  int ok = 0;
  for (i64 k = 0; k < K; k++) {                                        // :95-105
    if (c >= N) break;
    const u32 j = hc[k].idx;
    if (!b[j + 1] && hc[k].key != 0ull) { ok = 1; b[j + 1] = 1; c++; }
  }
  for (i64 k = 0; k < K; k++) {                                        // :107-116
    if (c >= N) break;
    const u32 j = hc[k].idx;
    if (!b[j + 1]) { b[j + 1] = 1; c++; }
  }
  if (c > N) ok = 1;                                                   // :117-119
  // computeLambda :179-192
  double l = 0.0;
  if (!(N > K) && N >= 1) {
    const double v = key_to_double(vkey), w = key_to_double(wkey);
    l = (v + w) / 2.0;
  }
  if (picked_out) for (i64 k = 0; k < K; k++) picked_out[k] = (i64)hc[k].idx;
  *lambda_out = l;
  *c_out = c;
  *ok_out = (ok || (eps_lambda > 0.0 && std::fabs(lambda_prev - l) >= eps_lambda)) ? 1 : 0;   // :133
}

int select_from_device(klr_ctx* ctx, const double* g_dev, i64 m, uint8_t* b, i64 c_prev, i64 N,
                       double lambda_prev, double eps_lambda, double* lambda_out, i64* c_out,
                       int* ok_out, i64* picked_out, bool keys_hist_ready, const double* extra_src, double* extra_out,
                       u64 hint_tau, const std::function<int()>* refill) {
  KLR_ARG(ctx, m >= 0 && N >= 0 && b != nullptr);
  if (extra_out) *extra_out = NAN;
  const i64 K = std::min<i64>(m, 2 * N);             // NLargestAbsFloat64(g_, 2N)
  std::vector<Cand> hc((size_t)std::max<i64>(K, 1));
  u64 vkey = 0, wkey = 0;
  if (K > 0) {
    hipStream_t st = ctx->stream;
    u64* keys; u32* hist; Cand* cand;
    KLR_TRY(sel_prepare(ctx, m, &keys, &hist));
    // WS_TMP0: GCAND_CAP compacted candidates.  The result -- [FastState (40 B) + caller scalar | K sorted candidates],
    // 3 KB -- is written by k_fast_finish straight into pinned host memory (device-visible, coherent): no copy kernel
    KLR_TRY(ws_get(ctx, WS_TMP0, (size_t)(GCAND_CAP + FAST_CAP) * sizeof(Cand), (void**)&cand));
    char* pin;
    KLR_TRY(pin_get(ctx, 64 + (size_t)std::max<i64>(K, FAST_CAP) * sizeof(Cand), (void**)&pin));
    FastState* fs = (FastState*)pin;
    double* extra_dst = (double*)(pin + 48);
    Cand* out = (Cand*)(pin + 64);
    if (!keys_hist_ready) {
      const unsigned grid = (unsigned)std::max<i64>(1, std::min<i64>(ceil_div(m, 256 * 4), 256));
      hipLaunchKernelGGL(k_fast_keys_hist, dim3(grid), dim3(256), 0, st, g_dev, m, keys, hist);
    }
    KLR_TRY(ctx_func_lds(ctx, (const void*)k_fast_finish, (int)(2 * FAST_CAP * sizeof(Cand))));
    KLR_TRY(ctx_func_lds(ctx, (const void*)k_fast_direct, (int)(2 * FAST_CAP * sizeof(Cand))));
    KLR_TRY(ctx_func_lds(ctx, (const void*)k_fast_rank, (int)(FAST_CAP * sizeof(Cand))));
    KLR_TRY(ctx_func_lds(ctx, (const void*)k_fast_rank_out, (int)(FAST_CAP * sizeof(Cand))));
    bool direct_ok = false, ranked_out = false;
    bool lean_unfilled = hint_tau != 0 && keys_hist_ready && refill;   // g and the keys are not on the device yet
    if (hint_tau != 0 && keys_hist_ready) {
      // the candidates are already in `cand` (SelHint): one sort kernel, no pass over the keys
      FastCtl* ctl = (FastCtl*)((char*)hist - sizeof(FastCtl));
      Cand* sorted_dev = cand + GCAND_CAP;             // WS_TMP0 holds the candidates and, behind them, their sorted copy
      if (ctx_opt(ctx, "KLR_SELECT_TWO_LAUNCHES")) {    // A/B: rank, then a one-workgroup kernel that copies out and finds v and w
        hipLaunchKernelGGL(k_fast_rank, dim3(FAST_RANK_WGS), dim3(1024), FAST_CAP * sizeof(Cand), st, (const FastCtl*)ctl, (u32)K, (const Cand*)cand, sorted_dev);
        hipLaunchKernelGGL(k_fast_direct, dim3(1), dim3(1024), FAST_CAP * sizeof(Cand), st, hist, ctl, hint_tau, (u32)K, (u32)N, (const Cand*)sorted_dev, fs, out,
                           extra_src, extra_dst);
      } else {
        hipLaunchKernelGGL(k_fast_rank_out, dim3(FAST_RANK_WGS), dim3(1024), FAST_CAP * sizeof(Cand), st, (const FastCtl*)ctl, hint_tau, (u32)K, (const Cand*)cand,
                           fs, out, extra_src, extra_dst);
        ranked_out = true;
        ctx->sel_cursor_dirty = true;
      }
      KLR_HIP(ctx, hipGetLastError());
      KLR_HIP(ctx, hipStreamSynchronize(st));
      direct_ok = fs->fallback != 2;
      if (direct_ok && ranked_out) {
        // v = the N-th key of the sorted list, w = the first key below it (all n candidates are here, not only the first K)
        const u32 n = fs->ncand;
        const u64 vk = (N >= 1 && (u32)N <= n) ? out[(size_t)N - 1].key : 0ull;
        fs->vkey = vk; fs->wfound = 0; fs->wkey = 0;
        for (u32 i = (u32)std::max<i64>(N, 1); i < n; i++) if (out[i].key < vk) { fs->wfound = 1; fs->wkey = out[i].key; break; }
      }
    }
    if (!direct_ok) {
      FastCtl* ctl = (FastCtl*)((char*)hist - sizeof(FastCtl));
      if (ctx->sel_cursor_dirty) {                   // a hinted selection's candidate count (this call's failed attempt, or an earlier call's)
        KLR_HIP(ctx, hipMemsetAsync(ctl, 0, sizeof(FastCtl), st));
        ctx->sel_cursor_dirty = false;
      }
      if (hint_tau != 0 && keys_hist_ready && refill) {
        KLR_TRY((*refill)());                        // a lean hinted sweep wrote neither g nor the keys: the conversion again, in full
        lean_unfilled = false;
      } else if (hint_tau != 0 && keys_hist_ready) {   // the hinted sweep did not build the coarse histogram
        const unsigned hgrid = (unsigned)std::max<i64>(1, std::min<i64>(ceil_div(m, 256 * 4), 256));
        hipLaunchKernelGGL(k_hist_from_keys, dim3(hgrid), dim3(256), 0, st, keys, m, hist);
      }
      const unsigned grid = (unsigned)std::max<i64>(1, std::min<i64>(ceil_div(m, 4096), 256));
      hipLaunchKernelGGL(k_fast_compact, dim3(grid), dim3(1024), 0, st, hist, hist + FAST_BINS, ctl, keys, m, (u32)K, cand);
      hipLaunchKernelGGL(k_fast_finish, dim3(1), dim3(1024), 2 * FAST_CAP * sizeof(Cand), st, hist, hist + FAST_BINS, ctl, (u32)K, (u32)N,
                         cand, fs, out, extra_src, extra_dst);
      KLR_HIP(ctx, hipGetLastError());
      KLR_HIP(ctx, hipStreamSynchronize(st));
    }
    FastState hs;
    memcpy(&hs, pin, sizeof(hs));
    if (extra_out) memcpy(extra_out, pin + 48, sizeof(double));
    memcpy(hc.data(), pin + 64, (size_t)K * sizeof(Cand));
    vkey = hs.vkey; wkey = hs.wkey;
    const bool need_w = !(N > K) && N >= 1;
    ctx->sel_hint = 0;
    if (hs.fallback || (need_w && !hs.wfound)) {
      // exact 8-pass radix select (keys are already on the device)
      if (lean_unfilled) {
        KLR_TRY((*refill)());
        ctx->sel_hist_zeroed = nullptr;              // the conversion also built the coarse histogram, which nobody consumes here
      }
      KLR_TRY(select_exact_device(ctx, g_dev, keys, true, m, K, N, hc, &vkey, &wkey));
    } else if (hc[(size_t)K - 1].key != 0ull) {
      // next time: compact everything above 0.9 x this call's K-th |g| while the keys are produced
      ctx->sel_hint = abs_key_host(0.9 * key_to_double(hc[(size_t)K - 1].key));
      ctx->sel_hint_K = K; ctx->sel_hint_m = m;
    }
  }
  if (K == 0 && extra_src && extra_out) KLR_TRY(d2h_sync(ctx, extra_out, extra_src, sizeof(double)));
  select_finish_host(hc, K, vkey, wkey, b, c_prev, N, lambda_prev, eps_lambda, lambda_out, c_out, ok_out, picked_out);
  return KLR_OK;
}

extern "C" int klr_select(klr_ctx* ctx, const double* g, int64_t m, uint8_t* b, int64_t c_prev, int64_t N,
                          double lambda_prev, double eps_lambda, double* lambda_out, int64_t* c_out, int* ok_out) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, (g != nullptr || m == 0) && b != nullptr && lambda_out && c_out && ok_out);
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  double* gd;
  KLR_TRY(ws_get(ctx, WS_G, (size_t)(m + 8) * sizeof(double), (void**)&gd));
  if (m) KLR_TRY(h2d(ctx, gd + 1, g, (size_t)m * sizeof(double)));
  return select_from_device(ctx, gd, m, b, c_prev, N, lambda_prev, eps_lambda, lambda_out, c_out, ok_out, nullptr);
}

// featureSelector.SelectFixed (kmerLr_feature_selection.go:136-159): a sequential scan in column
// order with an early exit -- O(m) compares on an array already on the host.
extern "C" int klr_select_fixed(klr_ctx* ctx, const double* g_full, int64_t m, uint8_t* b, int64_t c_prev,
                                double lambda, int64_t max_features, int64_t* c_out, int* ok_out) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, g_full != nullptr && b != nullptr && c_out && ok_out);
  int ok = 0; i64 c = c_prev;
  for (i64 k = 1; k < m + 1; k++) {
    if (std::fabs(g_full[k]) >= lambda && !b[k]) { ok = 1; b[k] = 1; c++; }
    if (max_features > 0 && c >= max_features && ok) break;
  }
  *c_out = c; *ok_out = ok;
  return KLR_OK;
}

// =================================================================================================
// The sweep's collective as reduce-scatter + local top-2N + all-gather of the candidates.
//
// ibuf = this rank's int64 column sums [0, dim) | bias hi, lo | loss limbs | non-finite count, zero padded to
// nranks * slice.  After the reduce-scatter rank r holds the EXACT sums of columns [r * slice, (r+1) * slice).  It
// converts them to keys, selects its 2N largest |g| with the same device kernels as the whole-vector selection (they
// sort by (|g| desc, column asc)), and all-gathers one block per rank:
//     [FastState 40 B | - | 5 int64 extras it owns at +64 | Cand[K_r] at +128]
// Every rank then merges the nranks sorted lists on the host -- the global 2N largest are among the local 2N largest
// -- and finishes Select / computeLambda exactly as the single-vector path does.  lambda needs w = the largest |g|
// BELOW the N-th: it is among the gathered candidates unless some rank's list is all >= v and its slice holds more
// columns than the list; then (and when a local selection fell back to the radix select) every rank -- they all see
// the same gathered blocks -- takes the all-reduce path on its intact ibuf instead.  Integer sums are the same
// numbers whichever collective adds them: selection and lambda are bitwise those of the all-reduce path.
// =================================================================================================
#define SCAT_HDR 128
// keys of a rank's slice of one range.  tau != 0 (hinted, SelHint): nothing but the candidate list is produced -- every
// key at or above tau goes into `cand` with its position in the rank's key list (pos0 + t), claimed per workgroup with
// one atomic on the global cursor; tau == 0: keys + coarse histogram for the regular selection.
__global__ __launch_bounds__(256) void k_slice_keys(const i64* __restrict__ rbuf, i64 lo, i64 feat0, i64 m_s, double inv_scale,
                                                    u64* __restrict__ keys, u32* __restrict__ hist, u64 tau, Cand* __restrict__ cand,
                                                    u32* __restrict__ cursor, u32 pos0) {
  __shared__ u32 sh[FAST_BINS];
  __shared__ Cand wcand[256];
  __shared__ u32 wcnt, wbase;
  if (threadIdx.x == 0) wcnt = 0u;
  if (tau == 0ull) for (int i = threadIdx.x; i < FAST_BINS; i += 256) sh[i] = 0;
  __syncthreads();
  for (i64 t = (i64)blockIdx.x * 256 + threadIdx.x; t < m_s; t += (i64)gridDim.x * 256) {
    const double v = (double)rbuf[feat0 + t - lo] * inv_scale;        // inv_scale is a power of two: exact
    const u64 k = abs_key(v);
    if (tau == 0ull) {
      keys[t] = k;
      atomicAdd(&sh[(u32)(k >> FAST_SHIFT)], 1u);
    } else if (k >= tau) {
      const u32 wp = atomicAdd(&wcnt, 1u);
      if (wp < 256u) { wcand[wp].key = k; wcand[wp].idx = pos0 + (u32)t; wcand[wp].pad = 0; }
      else {
        const u32 pos = atomicAdd(cursor, 1u);
        if (pos < GCAND_CAP) { cand[pos].key = k; cand[pos].idx = pos0 + (u32)t; cand[pos].pad = 0; }
      }
    }
  }
  __syncthreads();
  if (tau != 0ull) {
    if (threadIdx.x == 0) { const u32 nw = min(wcnt, 256u); wbase = nw ? atomicAdd(cursor, nw) : 0u; }
    __syncthreads();
    const u32 nw = min(wcnt, 256u);
    if (threadIdx.x < nw && wbase + threadIdx.x < GCAND_CAP) cand[wbase + threadIdx.x] = wcand[threadIdx.x];
  } else
    for (int i = threadIdx.x; i < FAST_BINS; i += 256) if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

// the five scalars behind the column sums (elements dim .. dim+4 of the vector whose slice this is): the rank that owns
// one writes it, the others 0
__global__ void k_slice_extras(const i64* __restrict__ rbuf, i64 lo, i64 slice, i64 dim, i64* __restrict__ dst, u32 m_s, u32 K_s) {
  const int q = threadIdx.x;
  if (q < IB_EXTRA) { const i64 j = dim + q; dst[q] = (j >= lo && j < lo + slice) ? rbuf[j - lo] : 0; }
  if (q == 5) dst[5] = ((i64)m_s << 32) | (i64)K_s;
  if (q == 6) dst[6] = 0;                                  // the list is a top-K_s list, not "every key at or above lowkey"
}

// Hinted slice selection, last launch (one workgroup): the rank's block of the all-gather in one go --
// [FastState | five scalars it owns, (m_s, listed), complete | the candidates sorted by k_fast_rank, at most kmax of them] --
// and the control block re-armed.  complete = 1: the list holds EVERY key of the slice at or above lowkey = tau.
__global__ __launch_bounds__(1024) void k_slice_report(FastCtl* __restrict__ ctl, u64 tau, u32 kmax, const Cand* __restrict__ sorted_dev,
                                                       const i64* __restrict__ rbuf, i64 lo, i64 slice, i64 dim, u32 m_s, char* __restrict__ blk) {
  const u32 t = threadIdx.x;
  const u32 n = ctl->cursor;
  __syncthreads();                                         // every lane has read the cursor
  if (t == 0) ctl->cursor = 0u;
  const bool over = n > FAST_CAP;                          // k_fast_rank left the list alone: the caller falls back
  const u32 listed = over ? 0u : min(n, kmax);
  FastState* st = (FastState*)blk;
  i64* ex = (i64*)(blk + 64);
  Cand* out = (Cand*)(blk + SCAT_HDR);
  if (t == 0) {
    st->lowkey = tau; st->ncand = n; st->cursor = listed; st->fallback = over ? 2u : 0u; st->wfound = 0; st->vkey = 0; st->wkey = 0;
    ex[5] = ((i64)m_s << 32) | (i64)listed;
    ex[6] = (!over && n <= kmax) ? 1 : 0;
  }
  if (t < IB_EXTRA) { const i64 j = dim + t; ex[t] = (j >= lo && j < lo + slice) ? rbuf[j - lo] : 0; }
  for (u32 i = t; i < listed; i += 1024) out[i] = sorted_dev[i];
}

// How the dim + 5 sums travel: G strip ranges (G = 1: the whole vector), range g a vector of its columns [col0, col1)
// -- the last one with the five scalars behind them -- cut into `nranks` slices of slice[g] elements.  Rank q owns the
// local elements [q slice, (q + 1) slice) of every range; its features are those that are columns >= 1.  A rank's keys are
// the concatenation of its ranges' features in range order, i.e. in ascending column order.
struct ScatterPlan {
  int G = 1, R = 1;
  std::vector<i64> col0, col1, len, slice, ioff, roff;
  i64 send_total = 0, recv_total = 0;
  i64 feat_lo(int q, int g) const { return (g == 0 && q == 0) ? 1 : (i64)q * slice[(size_t)g]; }                 // local element of the rank's first feature
  i64 feat_hi(int q, int g) const { return std::min<i64>((i64)(q + 1) * slice[(size_t)g], col1[(size_t)g] - col0[(size_t)g]); }
  i64 feat_cnt(int q, int g) const { return std::max<i64>(0, feat_hi(q, g) - feat_lo(q, g)); }
  i64 feat_total(int q) const { i64 t = 0; for (int g = 0; g < G; g++) t += feat_cnt(q, g); return t; }
  // 0-based feature (column - 1) of position `pos` of rank q's key list
  i64 feature_of(int q, i64 pos) const {
    for (int g = 0; g < G; g++) { const i64 c = feat_cnt(q, g); if (pos < c) return col0[(size_t)g] + feat_lo(q, g) + pos - 1; pos -= c; }
    return -1;
  }
};

static ScatterPlan scatter_plan(i64 dim, int R, const Q64Ranges* qr) {
  ScatterPlan P;
  P.R = R; P.G = qr ? qr->G : 1;
  for (int g = 0; g < P.G; g++) {
    const i64 c0 = qr ? qr->r[(size_t)g].col0 : 0, c1 = qr ? qr->r[(size_t)g].col1 : dim;
    const i64 len = c1 - c0 + (g == P.G - 1 ? IB_EXTRA : 0);
    const i64 sl = ceil_div(len, (i64)R);
    P.col0.push_back(c0); P.col1.push_back(c1); P.len.push_back(len); P.slice.push_back(sl);
    P.ioff.push_back(P.send_total); P.roff.push_back(P.recv_total);
    P.send_total += sl * R; P.recv_total += sl;
  }
  return P;
}

// Which collective the sweep of a vector of `dim` sums posts on THIS communicator, and in how many strip ranges.
// Both questions depend on the links, the number of ranks and RCCL, not on this code, so the installed hooks are TIMED
// once per vector length (3 calls after a warm-up each, HIP events on the ctx stream; the times are averaged over the
// ranks with one f64 all-reduce, so every rank takes the same decisions):
//   * form: reduce-scatter (dim + 5 int64) + all-gather (one 16 KB block per rank) + the slice work and the host merge
//     that come with it (~20 us more than the all-reduce path's conversion + selection, measured with one rank) against
//     ONE all-reduce of the vector.  The reduce-scatter moves half the bytes over the links but pays a second collective's
//     latency: it is used when t_rs + t_ag + 20 us < t_ar.
//   * ranges: splitting the sweep into G launches costs ~15 us per extra launch (start, tail, a second set of piece
//     vectors) and ~13 us per cross-stream event (MI355X, configs[3] shard: 142.7 -> 171.9 us for G = 2 with a collective
//     that costs nothing); it hides all but the last range's finish + reduce-scatter + key pass: G = 2 when half of t_rs
//     exceeds that overhead.
// KLR_RS=1 / 0 fixes the form, KLR_SWEEP_RANGES=g the ranges (1 = never split).
static int sweep_collective_auto(klr_ctx* ctx, i64 dim, i64 blk_bytes, bool* scatter_out, int* G_out) {
  if (ctx->probe_dim != dim) {
    const i64 R = ctx->nranks, slice = ceil_div(dim + IB_EXTRA, R);
    i64* buf;
    KLR_TRY(ws_get(ctx, WS_TMP1, (size_t)(slice * R + slice) * 8 + (size_t)blk_bytes * (size_t)(R + 1), (void**)&buf));
    KLR_HIP(ctx, hipMemsetAsync(buf, 0, (size_t)(slice * R + slice) * 8 + (size_t)blk_bytes * (size_t)(R + 1), ctx->stream));
    char* agb = (char*)(buf + slice * R + slice);
    int rc = 0;
    hipError_t he = hipSuccess;
    const int iters = 3;
    // warm-up: one call of each
    rc = ctx->rs_fn(ctx->coll_user, buf, buf + slice * R, slice, KLR_I64, (void*)ctx->stream);
    if (rc == 0) rc = ctx->ag_fn(ctx->coll_user, agb, agb + blk_bytes, blk_bytes, (void*)ctx->stream);
    if (rc == 0) rc = ctx->ar_fn(ctx->ar_user, buf, dim + IB_EXTRA, KLR_I64, (void*)ctx->stream);
    if (rc != 0) return klr_fail(ctx, KLR_ECOMM, "a collective hook failed while it was timed");
    // exact timings, one collective at a time
    double t[3] = {0.0, 0.0, 0.0};
    for (int which = 0; which < 3; which++) {
      hipEvent_t a, b;
      KLR_HIP(ctx, hipEventCreate(&a));
      KLR_HIP(ctx, hipEventCreate(&b));
      (void)hipEventRecord(a, ctx->stream);
      for (int i = 0; i < iters && rc == 0; i++) {
        if (which == 0) rc = ctx->rs_fn(ctx->coll_user, buf, buf + slice * R, slice, KLR_I64, (void*)ctx->stream);
        else if (which == 1) rc = ctx->ag_fn(ctx->coll_user, agb, agb + blk_bytes, blk_bytes, (void*)ctx->stream);
        else rc = ctx->ar_fn(ctx->ar_user, buf, dim + IB_EXTRA, KLR_I64, (void*)ctx->stream);
      }
      (void)hipEventRecord(b, ctx->stream);
      he = hipEventSynchronize(b);
      float m1 = 0.f;
      if (he == hipSuccess) he = hipEventElapsedTime(&m1, a, b);
      (void)hipEventDestroy(a); (void)hipEventDestroy(b);
      if (rc != 0) return klr_fail(ctx, KLR_ECOMM, "a collective hook failed while it was timed");
      if (he != hipSuccess) return klr_fail(ctx, KLR_EHIP, "timing the collectives failed: %s", hipGetErrorString(he));
      t[which] = (double)m1 / iters;
    }
    KLR_TRY(klr_allreduce_host(ctx, t, 3, KLR_F64));
    for (int i = 0; i < 3; i++) t[i] /= (double)R;
    ctx->probe_ms = t[0]; ctx->probe_ag_ms = t[1]; ctx->probe_ar_ms = t[2];
    const double overhead_ms = 0.030;                  // one extra launch + one event, measured
    ctx->probe_G = t[0] * 0.5 > overhead_ms ? 2 : 1;
    ctx->probe_scatter = t[0] + t[1] + 0.020 < t[2];
    ctx->probe_dim = dim;
  }
  *scatter_out = ctx->sw.rs == 1 ? true : ctx->probe_scatter;
  *G_out = ctx->sw.overlap >= 0 ? ctx->sw.overlap : ctx->probe_G;
  return KLR_OK;
}

// ibuf: the send vectors of all ranges (P.ioff); ranged = the sweep has NOT run yet: launch it range by range on the ctx
// stream and let a second stream finish, reduce-scatter and key each range as soon as its launch is done.
static int sweep_select_scatter(klr_ctx* ctx, const klr_mat* full, double scale_fx, const ScatterPlan& P, i64* ibuf, bool ranged, i64* part,
                                const unsigned long long* rowblk, i64 nrowblk, i64 N, uint8_t* b, i64 c_prev,
                                double lambda_prev, double eps_lambda, double* lambda_out, i64* c_out, int* ok_out, double* loss_sum,
                                bool* done) {
  *done = false;
  const i64 dim = full->dim, m = dim - 1;
  const int R = ctx->nranks, r = ctx->rank, G = P.G;
  hipStream_t stA = ctx->stream, st = stA;
  if (ranged) {
    if (!ctx->stream2) KLR_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    while ((int)ctx->range_ev.size() < G) { hipEvent_t ev; KLR_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence)); ctx->range_ev.push_back(ev); }   // device-side ordering only: no system-scope flush
    st = ctx->stream2;
  }
  const i64 m_s = P.feat_total(r);
  const i64 Kmax = std::min<i64>(m, 2 * N);                            // candidates per rank at most
  const i64 K_s = std::min<i64>(m_s, 2 * N);
  const size_t blk = SCAT_HDR + (size_t)std::max<i64>(Kmax, 1) * sizeof(Cand);
  i64* rbuf; char *sendb, *recvb;
  KLR_TRY(ws_get(ctx, WS_TMP1, (size_t)P.recv_total * 8 + blk * (size_t)(R + 1), (void**)&rbuf));
  sendb = (char*)(rbuf + P.recv_total); recvb = sendb + blk;
  u64* keys = nullptr; u32* hist = nullptr; Cand* cand = nullptr;
  KLR_TRY(sel_prepare(ctx, std::max<i64>(m_s, 1), &keys, &hist));
  KLR_TRY(ws_get(ctx, WS_TMP0, (size_t)(GCAND_CAP + FAST_CAP) * sizeof(Cand), (void**)&cand));
  KLR_TRY(ctx_func_lds(ctx, (const void*)k_fast_finish, (int)(2 * FAST_CAP * sizeof(Cand))));
  if (ranged) {
    // every range's launch is queued first; `range g is done` is an event the second stream waits for
    for (int g = 0; g < G; g++) {
      KLR_TRY(sweep_range_launch(ctx, full, g, part));
      KLR_HIP(ctx, hipEventRecord(ctx->range_ev[(size_t)g], stA));
    }
  }
  // hinted (every rank holds the same hint: it comes from the merged result of the previous sweep): candidates are
  // gathered while the keys are formed and ranked by many workgroups; else the regular bracket + trim + sort.  A hint that
  // turns out too high (fewer than K keys at or above it over all ranks, or none below the N-th) costs one more round of
  // the LOCAL selection without it -- the reduced slices are still there -- and one more all-gather; only the tie case in
  // which a top-K list cannot prove lambda goes back to the all-reduce.
  u64 tau = (ctx->sel_hint != 0 && ctx->sel_hint_K == Kmax && ctx->sel_hint_m == m) ? ctx->sel_hint : 0ull;
  FastCtl* ctl = (FastCtl*)((char*)hist - sizeof(FastCtl));
  Cand* sorted_dev = cand + GCAND_CAP;
  KLR_TRY(ctx_func_lds(ctx, (const void*)k_fast_rank, (int)(FAST_CAP * sizeof(Cand))));
  auto slice_keys = [&](int g, i64 koff) {
    const i64 cnt = P.feat_cnt(r, g);
    if (cnt > 0 && K_s > 0)
      hipLaunchKernelGGL(k_slice_keys, dim3((unsigned)std::max<i64>(1, std::min<i64>(ceil_div(cnt, 1024), 256))), dim3(256), 0, st,
                         (const i64*)(rbuf + P.roff[(size_t)g]), (i64)r * P.slice[(size_t)g], P.feat_lo(r, g), cnt, 1.0 / scale_fx, keys + koff, hist, tau,
                         cand, &ctl->cursor, (u32)koff);
    return cnt;
  };
  i64 koff = 0;
  for (int g = 0; g < G; g++) {
    i64* send = ibuf + P.ioff[(size_t)g];
    i64* recv = rbuf + P.roff[(size_t)g];
    const i64 len = P.len[(size_t)g], sl = P.slice[(size_t)g];
    if (ranged) {
      KLR_HIP(ctx, hipStreamWaitEvent(st, ctx->range_ev[(size_t)g], 0));
      KLR_TRY(sweep_range_finish(ctx, full, P.col0[(size_t)g], P.col1[(size_t)g], part, send, g == G - 1 ? rowblk : nullptr, nrowblk, st));
    }
    if (sl * R > len) KLR_HIP(ctx, hipMemsetAsync(send + len, 0, (size_t)(sl * R - len) * 8, st));
    if (ctx->rs_fn(ctx->coll_user, send, recv, sl, KLR_I64, (void*)st) != 0) return klr_fail(ctx, KLR_ECOMM, "reduce-scatter hook failed");
    koff += slice_keys(g, koff);
  }
  const i64 K = Kmax;
  const bool need_w = !(N > K) && N >= 1;
  u64 vkey = 0, wkey = 0;
  std::vector<Cand> hc;
  i64 tot[IB_EXTRA] = {0, 0, 0, 0, 0};
  bool fallback = false;
  char* pin;
  KLR_TRY(pin_get(ctx, blk * (size_t)R, (void**)&pin));
  for (int attempt = 0; attempt < 2; attempt++) {
    const int gl = G - 1;                                              // the five scalars ride behind the last range's columns
    if (tau) {
      hipLaunchKernelGGL(k_fast_rank, dim3(FAST_RANK_WGS), dim3(1024), FAST_CAP * sizeof(Cand), st, (const FastCtl*)ctl, 0u, (const Cand*)cand, sorted_dev);
      hipLaunchKernelGGL(k_slice_report, dim3(1), dim3(1024), 0, st, ctl, tau, (u32)Kmax, (const Cand*)sorted_dev, (const i64*)(rbuf + P.roff[(size_t)gl]),
                         (i64)r * P.slice[(size_t)gl], P.slice[(size_t)gl], P.col1[(size_t)gl] - P.col0[(size_t)gl], (u32)m_s, sendb);
    } else {
      KLR_HIP(ctx, hipMemsetAsync(sendb, 0, blk, st));
      if (K_s > 0) {
        const unsigned grid = (unsigned)std::max<i64>(1, std::min<i64>(ceil_div(m_s, 4096), 256));
        hipLaunchKernelGGL(k_fast_compact, dim3(grid), dim3(1024), 0, st, hist, hist + FAST_BINS, ctl, keys, m_s, (u32)K_s, cand);
        hipLaunchKernelGGL(k_fast_finish, dim3(1), dim3(1024), 2 * FAST_CAP * sizeof(Cand), st, hist, hist + FAST_BINS, ctl, (u32)K_s, (u32)std::min<i64>(N, K_s),
                           cand, (FastState*)sendb, (Cand*)(sendb + SCAT_HDR), (const double*)nullptr, (double*)nullptr);
      }
      hipLaunchKernelGGL(k_slice_extras, dim3(1), dim3(64), 0, st, rbuf + P.roff[(size_t)gl], (i64)r * P.slice[(size_t)gl], P.slice[(size_t)gl],
                         P.col1[(size_t)gl] - P.col0[(size_t)gl], (i64*)(sendb + 64), (u32)m_s, (u32)K_s);
    }
    KLR_HIP(ctx, hipGetLastError());
    if (ctx->ag_fn(ctx->coll_user, sendb, recvb, (int64_t)blk, (void*)st) != 0) return klr_fail(ctx, KLR_ECOMM, "all-gather hook failed");
    KLR_HIP(ctx, hipMemcpyAsync(pin, recvb, blk * (size_t)R, hipMemcpyDeviceToHost, st));
    KLR_HIP(ctx, hipStreamSynchronize(st));
    if (ranged && attempt == 0) KLR_HIP(ctx, hipStreamSynchronize(stA));   // (already complete: the second stream waited for its last event)
    // ---- merge on the host (identical on every rank): the R lists are sorted by (|g| desc, column asc); the global order
    // is a K-way merge of them -- the heads are compared, R - 1 comparisons per output, K + a few outputs
    bool retry = false;
    fallback = false;
    for (int u = 0; u < IB_EXTRA; u++) tot[u] = 0;
    std::vector<const Cand*> lst((size_t)R, nullptr);
    std::vector<i64> len_q((size_t)R, 0), mq_q((size_t)R, 0), head((size_t)R, 0);
    std::vector<u64> low_q((size_t)R, 0);          // != 0: the rank's list holds EVERY key of its slice at or above this one
    for (int q = 0; q < R; q++) {
      const char* bq = pin + blk * (size_t)q;
      FastState fs; memcpy(&fs, bq, sizeof(fs));
      i64 ex[7]; memcpy(ex, bq + 64, sizeof(ex));
      for (int u = 0; u < IB_EXTRA; u++) tot[u] += ex[u];
      mq_q[(size_t)q] = (i64)((u64)ex[5] >> 32); len_q[(size_t)q] = (i64)((u64)ex[5] & 0xFFFFFFFFull);
      if (fs.fallback) { if (tau) retry = true; else fallback = true; break; }
      if (ex[6] == 1) low_q[(size_t)q] = fs.lowkey;
      if (len_q[(size_t)q] == 0) continue;
      lst[(size_t)q] = (const Cand*)(bq + SCAT_HDR);
    }
    hc.clear();
    hc.reserve((size_t)K + 8);
    bool found_w = false;
    vkey = 0; wkey = 0;
    if (!fallback && !retry) {
      // global feature index of every list head (positions map piecewise onto columns: P.feature_of)
      auto cand_at = [&](int q, i64 pos) { Cand c = lst[(size_t)q][pos]; c.idx = (u32)P.feature_of(q, (i64)c.idx); c.pad = (u32)q; return c; };
      std::vector<Cand> hd((size_t)R);
      for (int q = 0; q < R; q++) if (len_q[(size_t)q] > 0) hd[(size_t)q] = cand_at(q, 0);
      for (;;) {
        int best = -1;
        for (int q = 0; q < R; q++) {
          if (head[(size_t)q] >= len_q[(size_t)q]) continue;
          const Cand& c = hd[(size_t)q];
          if (best < 0 || c.key > hd[(size_t)best].key || (c.key == hd[(size_t)best].key && c.idx < hd[(size_t)best].idx)) best = q;
        }
        if (best < 0) break;
        const Cand bc = hd[(size_t)best];
        if (++head[(size_t)best] < len_q[(size_t)best]) hd[(size_t)best] = cand_at(best, head[(size_t)best]);
        if ((i64)hc.size() < K) hc.push_back(bc);
        if ((i64)hc.size() >= std::max<i64>(N, 1) && need_w && !found_w) {
          vkey = hc[(size_t)N - 1].key;
          if (bc.key < vkey) { wkey = bc.key; found_w = true; }        // merged order: the first key below v is the largest one
        }
        if ((i64)hc.size() >= K && (!need_w || found_w)) break;
      }
      // what a rank did NOT list must rank behind everything that is used: a list complete down to its lowkey hides only
      // keys below that (the K-th merged key, and w, must be at or above it); a top-K list hides keys up to its last one
      // (which must then lie below v: w is at least as large); a list that is the whole slice hides nothing
      bool short_of = (i64)hc.size() < K || (need_w && !found_w);
      const u64 kth = (i64)hc.size() >= K && K > 0 ? hc[(size_t)K - 1].key : 0ull;
      if (need_w && (i64)hc.size() >= N) vkey = hc[(size_t)N - 1].key;
      for (int q = 0; q < R && !short_of; q++) {
        const i64 mq = mq_q[(size_t)q], Kq = len_q[(size_t)q];
        if (Kq == mq) continue;
        if (low_q[(size_t)q]) { if (kth < low_q[(size_t)q] || (need_w && wkey < low_q[(size_t)q])) short_of = true; }
        else if (Kq == 0) { if (mq > 0) short_of = true; }
        else if (need_w && lst[(size_t)q][Kq - 1].key >= vkey) short_of = true;
      }
      if (short_of) { if (tau) retry = true; else fallback = true; }
    }
    if (!retry) break;
    // once more without the hint: keys + histogram of every range's slice, then the regular selection
    tau = 0ull;
    ctx->sel_hint = 0;
    KLR_HIP(ctx, hipMemsetAsync(ctl, 0, sizeof(FastCtl), st));         // candidates of the failed attempt
    i64 ko = 0;
    for (int g = 0; g < G; g++) ko += slice_keys(g, ko);
  }
  if (fallback) return KLR_OK;                                         // *done stays false: the caller takes the all-reduce path
  if (K > 0 && hc[(size_t)K - 1].key != 0ull) {
    ctx->sel_hint = abs_key_host(0.9 * key_to_double(hc[(size_t)K - 1].key));
    ctx->sel_hint_K = K; ctx->sel_hint_m = m;
  } else ctx->sel_hint = 0;
  hc.resize((size_t)std::max<i64>(K, 1));
  select_finish_host(hc, K, vkey, wkey, b, c_prev, N, lambda_prev, eps_lambda, lambda_out, c_out, ok_out, nullptr);
  *loss_sum = tot[4] ? NAN : ((double)tot[2] + (double)tot[3] * (1.0 / 4398046511104.0)) * (1.0 / 1048576.0);
  *done = true;
  return KLR_OK;
}

// =================================================================================================
// fused leapfrog epoch step: kmerLr_estimator.go:235 (Select) with the gradient of
// kmerLr_feature_selection.go:87 computed in a single pass over the full CSC.
// =================================================================================================
extern "C" int klr_sweep_select(klr_ctx* ctx, const klr_mat* full, const klr_mat* reduced,
                                const int64_t* red_cols, const double* theta_red, int64_t c_plus_1,
                                const double cw[2], int64_t N, double lambda_prev, double eps_lambda,
                                const double* scale, const double* offset,
                                uint8_t* b, double* lambda_out, int64_t* c_out, int* ok_out,
                                double* loss_out, double* g_out) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, full != nullptr && cw != nullptr && b != nullptr && lambda_out && c_out && ok_out);
  KLR_ARG(ctx, theta_red != nullptr && c_plus_1 >= 1);
  KLR_ARG(ctx, reduced == nullptr || (reduced->dim == c_plus_1 && reduced->n == full->n && red_cols != nullptr));
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  const i64 n = full->n, dim = full->dim, m = dim - 1;
  double *s, *gbuf, *thd = nullptr;
  KLR_TRY(ws_get(ctx, WS_S, (size_t)std::max<i64>(n, 1) * sizeof(double), (void**)&s));
  KLR_TRY(ws_get(ctx, WS_G, (size_t)(dim + 8) * sizeof(double), (void**)&gbuf));
  // restoreNonzero (:194-219): b[0] = 1 and b[j] = 1 where theta_j != 0; c = count
  i64 c_prev = 0;
  bool theta_inline = false;
  bool use_int;                                        // integer stream on EVERY rank, or the f64 protocol on every rank
  KLR_TRY(sweep_protocol(ctx, full, &use_int));
  // the mask itself (dim bytes: 1 MB at a million columns, 20 us of the host's time) is written once the kernels are
  // queued -- restore_b below -- so that the device is already busy while the host clears it
  auto restore_b = [&]() {
    memset(b, 0, (size_t)dim);
    b[0] = 1;
    if (reduced) for (i64 p = 1; p < c_plus_1; p++) if (theta_red[p] != 0.0) b[red_cols[p]] = 1;
  };
  if (reduced) {
    for (i64 p = 1; p < c_plus_1; p++) {
      KLR_ARG(ctx, red_cols[p] > 0 && red_cols[p] < dim);
      if (theta_red[p] != 0.0) c_prev++;
    }
    KLR_TRY(ws_get(ctx, WS_THETA, (size_t)c_plus_1 * sizeof(double), (void**)&thd));
    // count data and a small reduced problem: theta rides in the row pass's kernel arguments (a 1 KB upload
    // is a 5 us blit kernel of its own)
    theta_inline = use_int && c_plus_1 <= KLR_THETA_INLINE_MAX;
    if (!theta_inline) KLR_TRY(h2d(ctx, thd, theta_red, (size_t)c_plus_1 * sizeof(double)));
  }
  bool keys_ready = false, restored = false;
  u64 hint_tau = 0;
  std::function<int()> refill;                         // set when the hinted sweep wrote candidates only (SelHint.lean)
  if (use_int) {
    // count data: row pass -> int64 fixed-point weights + exact loss limbs; integer sweep; ONE int64
    // all-reduce of [g | bias | loss]; conversion to f64 emits the selection keys + histogram
    double scale_fx; i64* S; unsigned long long* acc; u64* keys; u32* hist;
    KLR_TRY(sel_prepare(ctx, m, &keys, &hist));
    SelHint hint{0ull, nullptr, nullptr};
    if (ctx->sel_hint != 0 && ctx->sel_hint_K == std::min<i64>(m, 2 * N) && ctx->sel_hint_m == m && !(scale || offset)) {
      KLR_TRY(ws_get(ctx, WS_TMP0, (size_t)(GCAND_CAP + FAST_CAP) * sizeof(Cand), (void**)&hint.cand));
      hint.tau = ctx->sel_hint;
      hint.ctl = (FastCtl*)((char*)hist - sizeof(FastCtl));
      hint.lean = g_out == nullptr && !ctx_opt(ctx, "KLR_SELECT_NO_LEAN");     // nothing reads g or the keys while the hint holds
      hint_tau = hint.tau;
    }
    KLR_TRY(sweep_fix_begin(ctx, full, cw, &scale_fx, &S, &acc));
    unsigned long long* rowblk; i64 nrowblk;
    (void)acc;
    KLR_TRY(launch_row_pass_fix(ctx, reduced ? reduced : full, thd, theta_red[0], reduced == nullptr, cw, scale_fx, S,
                                &rowblk, &nrowblk, theta_inline ? theta_red : nullptr, theta_inline ? c_plus_1 : 0,
                                &((FastCtl*)((char*)hist - sizeof(FastCtl)))->cursor));   // the row pass clears the candidate cursor
    ctx->sel_cursor_dirty = false;
    const bool tr = scale != nullptr || offset != nullptr;
    // reduce-scatter form of the collective: wide matrices on several ranks with both hooks installed, nothing that
    // needs the whole vector on every rank (g_out, a transform), a selection the one-workgroup sort can hold
    const i64 rs_min_dim = ctx->sw.rs_min_dim;
    const bool rs_force = ctx->sw.rs_force;                                // plumbing check: take the path with one rank too
    // ... on a context whose hooks passed klr_ctx_comm_selfcheck (KLR_RS=0 keeps the all-reduce, KLR_RS=1 skips that test)
    const bool scatter = ctx->rs_fn && ctx->ag_fn && ctx->ar_fn && (ctx->nranks > 1 || rs_force) && !tr && !g_out && dim >= rs_min_dim &&
                         2 * N <= FAST_CAP && N >= 1 && ctx->sw.rs != 0 && (ctx->comm_checked || ctx->sw.rs == 1);
    bool scatter_now = scatter;
    int G_want = 1;
    if (scatter) {
      // the form (reduce-scatter + all-gather, or one all-reduce) and the strip ranges from the MEASURED collectives of this
      // communicator (sweep_collective_auto); a forced form (KLR_RS=1, KLR_RS_FORCE) with fixed ranges needs no probe
      if (ctx->sw.rs == 1 && ctx->sw.overlap >= 0) G_want = ctx->sw.overlap;
      else KLR_TRY(sweep_collective_auto(ctx, dim, (i64)(SCAT_HDR + (size_t)std::max<i64>(std::min<i64>(m, 2 * N), 1) * sizeof(Cand)), &scatter_now, &G_want));
      if (rs_force) scatter_now = true;
      if (!(full->fmt == FMT_Q64C12 || full->fmt == FMT_Q64C13)) G_want = 1;
    }
    if (scatter_now) {
      // Strip ranges: the sweep runs as one launch per range and range g's sums are finished, reduce-scattered and keyed on
      // a second stream while range g + 1 streams.
      const Q64Ranges* qr = nullptr;
      i64* part_r = nullptr;
      KLR_TRY(sweep_ranges_prepare(ctx, full, G_want, &qr, &part_r));
      const ScatterPlan P = scatter_plan(dim, ctx->nranks, qr);
      i64* ibuf = nullptr;
      if (qr) KLR_TRY(ws_get(ctx, WS_TMP2, (size_t)(P.send_total + dim + IB_EXTRA) * 8, (void**)&ibuf));
      else KLR_TRY(sweep_fix_finish(ctx, full, scale_fx, gbuf, nullptr, nullptr, rowblk, nrowblk, nullptr, &ibuf, P.send_total));
      bool done = false;
      double loss_sum_s = NAN;
      restore_b(); restored = true;
      KLR_TRY(sweep_select_scatter(ctx, full, scale_fx, P, ibuf, qr != nullptr, part_r, rowblk, nrowblk, N, b, c_prev, lambda_prev, eps_lambda,
                                   lambda_out, c_out, ok_out, &loss_sum_s, &done));
      ctx->n_rs_sweeps++;
      if (qr) ctx->n_ranged_sweeps++;
      if (done && full->rs_verified) {
        if (loss_out) *loss_out = full->n_total ? loss_sum_s / (double)full->n_total : 0.0;
        return KLR_OK;
      }
      if (!done) ctx->n_rs_fallbacks++;
      // the all-reduce form on this rank's intact local sums: as ONE vector (the ranged form sums its pieces once more)
      i64* ibc = ibuf;
      if (qr) {
        ibc = ibuf + P.send_total;
        KLR_TRY(sweep_range_finish(ctx, full, 0, dim, part_r, ibc, rowblk, nrowblk, ctx->stream));
      }
      ctx->sel_hist_zeroed = nullptr;                                // the slice selection used the same buffers: start clean
      KLR_TRY(sel_prepare(ctx, m, &keys, &hist));
      KLR_TRY(sweep_fix_tail(ctx, full, scale_fx, ibc, gbuf, keys, hist, nullptr));
      hint_tau = 0;
      if (done) {
        // the FIRST reduce-scatter sweep on a matrix is repeated as an all-reduce on the intact local sums: the two forms
        // must give the same mask, count, lambda and loss bit for bit (the sums are integers), or the run ends here
        std::vector<uint8_t> b2((size_t)dim, 0);
        b2[0] = 1;
        if (reduced) for (i64 p = 1; p < c_plus_1; p++) if (theta_red[p] != 0.0) b2[(size_t)red_cols[p]] = 1;
        double lam2 = 0.0, loss2 = NAN; i64 c2 = 0; int ok2 = 0;
        const u64 keep_hint = ctx->sel_hint; const i64 keep_K = ctx->sel_hint_K, keep_m = ctx->sel_hint_m;
        KLR_TRY(select_from_device(ctx, gbuf, m, b2.data(), c_prev, N, lambda_prev, eps_lambda, &lam2, &c2, &ok2, nullptr, true, gbuf + dim, &loss2, 0));
        ctx->sel_hint = keep_hint; ctx->sel_hint_K = keep_K; ctx->sel_hint_m = keep_m;
        const bool same = lam2 == *lambda_out && c2 == *c_out && ok2 == *ok_out && memcmp(b2.data(), b, (size_t)dim) == 0 &&
                          (loss2 == loss_sum_s || (loss2 != loss2 && loss_sum_s != loss_sum_s));
        if (!same)
          return klr_fail(ctx, KLR_ECOMM, "reduce-scatter and all-reduce forms of the sweep's collective disagree (lambda %.17g / %.17g, c %lld / %lld): "
                          "the installed collectives do not sum the same data", *lambda_out, lam2, (long long)*c_out, (long long)c2);
        const_cast<klr_mat*>(full)->rs_verified = 1;
        if (loss_out) *loss_out = full->n_total ? loss_sum_s / (double)full->n_total : 0.0;
        return KLR_OK;
      }
    } else {
      if (ctx->ar_fn) ctx->n_ar_sweeps++;
      KLR_TRY(sweep_fix_finish(ctx, full, scale_fx, gbuf, tr ? nullptr : keys, tr ? nullptr : hist, rowblk, nrowblk, hint.tau ? &hint : nullptr));
      if (hint.tau && hint.lean && !tr)
        refill = [=]() { return sweep_fix_emit(ctx, full, scale_fx, gbuf, keys, hist, rowblk, nrowblk); };
    }
    keys_ready = m > 0 && !tr;
  } else {
    // the row pass leaves the local loss partial in gbuf[dim]; [g | loss] in ONE f64 all-reduce
    if (reduced) KLR_TRY(launch_row_pass(ctx, reduced, thd, ROW_SWEEP, cw, s, nullptr, gbuf + dim));
    else         KLR_TRY(launch_row_const(ctx, full, theta_red[0], ROW_SWEEP, cw, s, nullptr, gbuf + dim));
    KLR_TRY(sweep_all(ctx, full, cw, s, nullptr, gbuf));
  }
  if (!restored) restore_b();
  if (scale || offset) KLR_TRY(apply_transform_dev(ctx, gbuf, dim, scale, offset));
  // g_out is the gradient the selection sees (transformed like klr_gradient's), kmerLr_logistic_regression.go:196-222
  if (g_out) KLR_HIP(ctx, hipMemcpyAsync(g_out, gbuf, (size_t)dim * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));

  double loss_sum = NAN;
  KLR_TRY(select_from_device(ctx, gbuf, m, b, c_prev, N, lambda_prev, eps_lambda, lambda_out, c_out, ok_out, nullptr, keys_ready,
                             gbuf + dim, &loss_sum, hint_tau, refill ? &refill : nullptr));
  KLR_HIP(ctx, hipStreamSynchronize(ctx->stream));     // g_out copy (select may not have synchronised)
  if (loss_out) *loss_out = full->n_total ? loss_sum / (double)full->n_total : 0.0;
  return KLR_OK;
}
