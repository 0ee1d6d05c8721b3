// klr_solve.hip -- the reduced-problem solver: IRLS + covariance-update cyclic coordinate descent
// with soft threshold (the algorithm of kmerLr_estimator_coordinate.go:31-139, see oracle
// klo_cd_solve for the exact semantics), eval_stopping (kmerLr_estimator_proximal.go:30-52) and the
// hook's loss rule (kmerLr_estimator_hook.go:46-99).
//
// Per IRLS iteration:
//   row pass (CSR)          r, p, w = cw p(1-p), z                              [n rows]
//   Gram     (CSC + CSR)    G = X'WX, xy = X'Wz  -> ONE all-reduce of d*d+d doubles
//   sweeps   (1 workgroup)  Gauss-Seidel over the d coordinates on the replicated Gram; q = G theta is
//                           kept current with a rank-1 update per changed coordinate, so coordinates
//                           whose value does not move cost no synchronisation
//   loss     (CSR)          log-loss of every sweep's theta in one pass -> ONE all-reduce of K scalars
// Sweeps are issued K at a time ("speculatively"); the host replays the reference's per-sweep
// hook/stop logic over the K recorded iterates and discards those after the stopping one, so the
// iterate sequence is exactly the sweep-by-sweep one.
#include "klr_internal.h"

#include <algorithm>
#include <chrono>
#include <cmath>

#define KSPEC 16         // most sweeps per launch (a batch is 8 or 4 by default, up to this many when the run's history says so)
#define CD_THREADS 1024


// Gram scale: 2^E with max_{j>=1} G_jj * 2^E < 2^61 (|G_jk| <= max diag by Cauchy-Schwarz).  Computed on
// the device from the ALL-REDUCED diagonal, so every rank uses the same power of two.
__global__ __launch_bounds__(256) void k_gram_scale(const double* __restrict__ nm, int d, double* __restrict__ sc) {
  __shared__ double sh[256];
  double v = 0.0;
  for (int j = 1 + threadIdx.x; j < d; j += 256) v = fmax(v, fabs(nm[j]));
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double m = sh[0];
    if (!(m > 0.0) || m != m || m > 1e300) m = 1.0;
    int e;
    (void)frexp(m, &e);                 // m < 2^e
    sc[0] = ldexp(1.0, 61 - e);
    sc[1] = ldexp(1.0, e - 61);
  }
}

// Column sums of a re-linearisation from their chunk partials, three per column: a0 | nm | xy (contiguous, 3d).
// A wavefront per column: lanes stride over the column's chunks, then a fixed shuffle tree (the bias column of a
// reduced matrix is n entries = hundreds of chunks; one thread walking them took 28 us).  SCALE (single rank: the
// sums are already global): the Gram scale in the same launch -- the largest |diagonal| by an integer atomic max on
// the bits of the non-negative doubles, the last block to finish (ticket) turns it into the power of two and
// resets both words.  ctl = {ticket, max bits}, zero before the first launch.
template <bool SCALE>
__global__ __launch_bounds__(256) void k_irls_finish(const i64* __restrict__ colchunk, int d, i64 nchunks,
                                                     const double* __restrict__ part, double* __restrict__ a0nmxy,
                                                     double* __restrict__ sc, unsigned long long* __restrict__ ctl) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j < d) {
    const i64 c0 = colchunk[j], c1 = colchunk[j + 1];
    double v0 = 0.0, v1 = 0.0, v2 = 0.0;
    for (i64 c = c0 + lane; c < c1; c += 64) { v0 += part[c]; v1 += part[nchunks + c]; v2 += part[2 * nchunks + c]; }
    for (int o = 32; o > 0; o >>= 1) {
      v0 += __shfl_xor(v0, o, 64); v1 += __shfl_xor(v1, o, 64); v2 += __shfl_xor(v2, o, 64);
    }
    if (lane == 0) {
      a0nmxy[j] = v0; a0nmxy[d + j] = v1; a0nmxy[2 * d + j] = v2;
      if (SCALE && j >= 1 && v1 == v1) atomicMax(&ctl[1], (unsigned long long)__double_as_longlong(fabs(v1)));
    }
  }
  if (!SCALE) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned long long t = atomicAdd(&ctl[0], 1ull);
    if (t == gridDim.x - 1) {                    // k_gram_scale
      __threadfence();
      double m = __longlong_as_double((long long)atomicMax(&ctl[1], 0ull));
      if (!(m > 0.0) || m != m || m > 1e300) m = 1.0;
      int e;
      (void)frexp(m, &e);
      sc[0] = ldexp(1.0, 61 - e);
      sc[1] = ldexp(1.0, e - 61);
      atomicExch(&ctl[1], 0ull);
      atomicExch(&ctl[0], 0ull);
    }
  }
}

// off-diagonal, non-bias Gram cells from the short reduced rows, upper triangle only:
//   Gi[ja][jb] += round(w_i v_a v_b 2^E)  for a < b   (integer atomics: exact, order free)
// 16 lanes per row: every lane holds one stored entry (coalesced load), partners arrive by shuffle, so a row
// of q entries costs ~q steps instead of q^2/2 dependent load round trips (rows longer than 16 are chunked).
// LDSACC: the d x d integer Gram is privatised in LDS (d*d*8 <= 128 KiB) and written out as one partial per
// workgroup (summed by k_gram_reduce); otherwise global integer atomics.
template <int O>
__device__ __forceinline__ void dpp_pairs(bool ha, int ja, double wa, int jc_own, double vc_own, int d, unsigned long long* acc,
                                          int nrot) {
  if constexpr (O < 16) {
    // row_ror:O rotates right: lane l receives lane (l - O) & 15, i.e. the entry 16 - O places further on.  Inside
    // the same 16 entries a row of at most nrot entries has partners 1 .. nrot - 1 places on: rotations 17 - nrot .. 15
    // (nrot is the same for every active lane; 16 = all rotations, a different chunk of the row)
    if (O == 0 || O > 16 - nrot) {
      int jc; double vc;
      if constexpr (O == 0) { jc = jc_own; vc = vc_own; }
      else {
        constexpr int ctrl = 0x120 + O;
        jc = __builtin_amdgcn_update_dpp(0, jc_own, ctrl, 0xF, 0xF, false);
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(vc_own), ctrl, 0xF, 0xF, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(vc_own), ctrl, 0xF, 0xF, false);
        vc = __hiloint2double(hi, lo);
      }
      // columns ascend inside a row: the pair is taken once, with ja < jc
      if (ha && jc > ja) atomicAdd(&acc[ja * d + jc], (unsigned long long)__double2ll_rn(wa * vc));
    }
    dpp_pairs<O + 1>(ha, ja, wa, jc_own, vc_own, d, acc, nrot);
  }
}

// packed upper triangle without the bias row: cells (j, k), 1 <= j < k < d, row after row
__host__ __device__ __forceinline__ int gram_tri_count(int d) { return d >= 3 ? (d - 1) * (d - 2) / 2 : 0; }
__host__ __device__ __forceinline__ int gram_tri_index(int j, int k, int d) { return (j - 1) * (d - 1) - (j - 1) * j / 2 + (k - j - 1); }

template <bool LDSACC>
__global__ __launch_bounds__(1024) void k_gram_offdiag(const i64* __restrict__ rowptr, const u32* __restrict__ col,
                                                       const double* __restrict__ val, const double* __restrict__ w,
                                                       i64 n, int d, const double* __restrict__ sc,
                                                       unsigned long long* __restrict__ Gi) {
  extern __shared__ unsigned long long gl[];
  const int dd = d * d;
  if (LDSACC) {
    for (int t = threadIdx.x; t < dd; t += 1024) gl[t] = 0ull;
    __syncthreads();
  }
  unsigned long long* acc = LDSACC ? gl : Gi;
  const double s0 = sc[0];
  const int lane = threadIdx.x & 15;
  const i64 grp0 = ((i64)blockIdx.x * 1024 + threadIdx.x) >> 4, ngrp = ((i64)gridDim.x * 1024) >> 4;
  // two rows per trip: their offsets, then their first 16 entries, are requested together (a row is otherwise two
  // dependent round trips to L2 with nothing else in flight)
  constexpr int RB = 2;
  for (i64 i0 = grp0; i0 < n; i0 += RB * ngrp) {
    i64 rb[RB], re[RB]; double rw[RB];
#pragma unroll
    for (int r = 0; r < RB; r++) {
      const i64 i = i0 + r * ngrp;
      const bool ok = i < n;
      rb[r] = ok ? rowptr[i] + 1 : 0;                           // skip the bias entry
      re[r] = ok ? rowptr[i + 1] : 0;
      rw[r] = ok ? w[i] : 0.0;
    }
    int fj[RB]; double fv[RB];
#pragma unroll
    for (int r = 0; r < RB; r++) {
      const bool h = rb[r] + lane < re[r];
      fj[r] = h ? (int)col[rb[r] + lane] : -1;
      fv[r] = h ? val[rb[r] + lane] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < RB; r++) {
      const i64 b = rb[r], e = re[r];
      const double wi = rw[r] * s0;
      for (i64 a0 = b; a0 < e; a0 += 16) {
        int ja_raw; double va_raw;
        if (a0 == b) { ja_raw = fj[r]; va_raw = fv[r]; }
        else { const bool h = a0 + lane < e; ja_raw = h ? (int)col[a0 + lane] : -1; va_raw = h ? val[a0 + lane] : 0.0; }
        const bool ha = ja_raw >= 0;
        const int ja = ha ? ja_raw : 0;
        const double wa = ha ? wi * va_raw : 0.0;
        // the 16 lanes of a row are one DPP row: partners arrive by row rotations (VALU, no LDS round trip);
        // inside the same 16 entries only the rotations that reach 1 .. (longest row of the wavefront) - 1 places on
        const unsigned long long hm = __ballot(ha);
        const unsigned m16 = (unsigned)((hm | (hm >> 16) | (hm >> 32) | (hm >> 48)) & 0xFFFFull);
        dpp_pairs<0>(ha, ja, wa, ja_raw, va_raw, d, acc, 32 - __clz((int)m16));
        for (i64 c0 = a0 + 16; c0 < e; c0 += 16) {
          const bool hc = c0 + lane < e;
          const int jc_own = hc ? (int)col[c0 + lane] : -1;
          const double vc_own = hc ? val[c0 + lane] : 0.0;
          dpp_pairs<0>(ha, ja, wa, jc_own, vc_own, d, acc, 16);
        }
      }
    }
  }
  if (LDSACC) {
    __syncthreads();
    // per-workgroup partial: only the cells that carry integers, j < k with j >= 1, packed row after row
    // (gram_tri_index) -- half the bytes of the d x d square, written here and read back by the reduce
    unsigned long long* out = Gi + (size_t)dd + (size_t)blockIdx.x * gram_tri_count(d);
    for (int t = threadIdx.x; t < dd; t += 1024) {
      const int j = t / d, k = t - j * d;
      if (j >= 1 && j < k) out[gram_tri_index(j, k, d)] = gl[t];
    }
  }
}

// 128 < d <= 1024: the integer Gram no longer fits LDS and global 64-bit atomics run at ~20 G/s at the memory side
// (440 us per re-linearisation for 5.6M pairs at d = 501).  The Gram is cut into GT x GT tiles; workgroup
// (row group g, tile pair A <= B) privatises ONE tile in LDS (128 KiB), walks its rows with the same 16-lane DPP
// scheme and keeps the pairs (ja in A, jb in B); partial tiles are summed by k_gram_tiled_reduce.  The rows are read
// once per tile pair (they are short and L2 resident); the pair work is the same in total.
#define GT 128
__device__ __forceinline__ void gram_tile_pair(int p, int nt, int& A, int& B) {       // p-th pair (A <= B), row major
  A = 0;
  while (p >= nt - A) { p -= nt - A; A++; }
  B = A + p;
}

// toff[i][t], t = 0 .. nt: position (inside row i, bias skipped) of the first entry with column >= t * GT -- the
// column tiles of a row; the rows of a reduced matrix do not change during a solve, so this is built once per call
__global__ __launch_bounds__(256) void k_row_tile_offsets(const i64* __restrict__ rowptr, const u32* __restrict__ col, i64 n,
                                                          int nt, unsigned short* __restrict__ toff) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const i64 b = rowptr[i] + 1, e = rowptr[i + 1];
  unsigned short* o = toff + i * (nt + 1);
  int t = 0;
  for (i64 p = b; p < e; p++) {
    const int ct = (int)(col[p] / GT);
    while (t <= ct) o[t++] = (unsigned short)(p - b);
  }
  while (t <= nt) o[t++] = (unsigned short)(e - b);
}

// A thread per row: with the tile offsets a row knows at once which of its entries fall in tile A and which in
// tile B (usually a few, often none) and takes exactly those pairs -- the 16-lane rotation scheme of the small case
// walked all 256 lane pairs of every row in every one of the (nt + 1) nt / 2 passes (700 us at d = 1001).  Rows longer
// than 65535 entries cannot occur (d <= 1024).  Same addends as k_gram_offdiag: round(w_i v_a 2^E * v_b).  (For the
// single-tile kernel above a thread per row is SLOWER than the rotations: 45 sequential pairs per thread, 24.3 -> 33 ms
// for the configs[2] estimate.)
__global__ __launch_bounds__(1024) void k_gram_offdiag_tiled(const i64* __restrict__ rowptr, const u32* __restrict__ col,
                                                             const double* __restrict__ val, const double* __restrict__ w,
                                                             const unsigned short* __restrict__ toff,
                                                             i64 n, int d, const double* __restrict__ sc,
                                                             unsigned long long* __restrict__ part) {
  extern __shared__ unsigned long long gl[];
  for (int t = threadIdx.x; t < GT * GT; t += 1024) gl[t] = 0ull;
  __syncthreads();
  const int nt = (d + GT - 1) / GT;
  int A, B;
  gram_tile_pair((int)blockIdx.y, nt, A, B);
  const int A0 = A * GT, B0 = B * GT;
  const double s0 = sc[0];
  for (i64 i = (i64)blockIdx.x * 1024 + threadIdx.x; i < n; i += (i64)gridDim.x * 1024) {
    const unsigned short* o = toff + i * (nt + 1);
    const int a0 = o[A], a1 = o[A + 1], b0 = o[B], b1 = o[B + 1];
    if (a1 == a0 || b1 == b0) continue;
    const i64 rb = rowptr[i] + 1;
    const double wi = w[i] * s0;
    for (int pa = a0; pa < a1; pa++) {
      const int ja = (int)col[rb + pa] - A0;
      const double wa = wi * val[rb + pa];
      for (int pb = (b0 > pa + 1 ? b0 : pa + 1); pb < b1; pb++)
        atomicAdd(&gl[ja * GT + ((int)col[rb + pb] - B0)], (unsigned long long)__double2ll_rn(wa * val[rb + pb]));
    }
  }
  __syncthreads();
  unsigned long long* out = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (GT * GT);
  for (int t = threadIdx.x; t < GT * GT; t += 1024) out[t] = gl[t];
}

// Many columns (d > GRAM_ROWS_MIN): ONE ROW OF THE GRAM MATRIX PER WORKGROUP.  Workgroup a walks column a of the
// reduced matrix (CSC: the rows that hold it) and, for each such row i, the entries of row i to the right of a (CSR):
// every pair (a, b > a) of the row, i.e. exactly the cells of Gram row a -- accumulated with LDS atomics in a
// d-entry int64 vector and written out once, complete: no partials, no reduce, no memory-side atomics (24M of them took
// 700 us per re-linearisation at d = 5001), no tile passes.  Same addends as the other builds.  Each 16-lane group
// takes one entry of the column and spreads over that row's entries.
__global__ __launch_bounds__(1024) void k_gram_rows(const i64* __restrict__ rowptr, const u32* __restrict__ col,
                                                    const double* __restrict__ val, const i64* __restrict__ colptr,
                                                    const u32* __restrict__ collen, const u32* __restrict__ crow,
                                                    const double* __restrict__ cval, const double* __restrict__ w,
                                                    int d, const double* __restrict__ sc, i64* __restrict__ Gi) {
  extern __shared__ unsigned long long gl[];            // d accumulators: cells (a, .)
  const int a = blockIdx.x;                             // column / Gram row (0 = bias: from the column sums, zeros here)
  for (int t = threadIdx.x; t < d; t += 1024) gl[t] = 0ull;
  __syncthreads();
  if (a >= 1) {
    const double s0 = sc[0];
    const i64 cb = colptr[a];
    const u32 len = collen[a];
    const int lane = threadIdx.x & 15, grp = threadIdx.x >> 4;
    // two column entries per group and trip: their rows' offsets, then the rows' first 16 entries, in flight together
    for (u32 e0 = grp; e0 < len; e0 += 128) {
      u32 ee[2]; i64 rb[2], re[2]; double wa[2];
#pragma unroll
      for (int r = 0; r < 2; r++) {
        ee[r] = e0 + 64u * r;
        const bool ok = ee[r] < len;
        const u32 i = ok ? crow[cb + ee[r]] : 0u;
        const double va = ok ? cval[cb + ee[r]] : 0.0;
        rb[r] = ok ? rowptr[i] + 1 : 0;                 // skip the bias entry
        re[r] = ok ? rowptr[i + 1] : 0;
        wa[r] = ok ? (w[i] * s0) * va : 0.0;
      }
      int fj[2]; double fv[2];
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const bool h = rb[r] + lane < re[r];
        fj[r] = h ? (int)col[rb[r] + lane] : -1;
        fv[r] = h ? val[rb[r] + lane] : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 2; r++) {
        if (fj[r] > a) atomicAdd(&gl[fj[r]], (unsigned long long)__double2ll_rn(wa[r] * fv[r]));
        for (i64 p = rb[r] + 16 + lane; p < re[r]; p += 16) {
          const int jb = (int)col[p];
          if (jb > a) atomicAdd(&gl[jb], (unsigned long long)__double2ll_rn(wa[r] * val[p]));
        }
      }
    }
  }
  __syncthreads();
  i64* out = Gi + (i64)a * d;
  for (int t = threadIdx.x; t < d; t += 1024) out[t] = (i64)gl[t];
}

// cell (j, k), 1 <= j < k: the sum of its tile's partials in group order.  G != null (single rank): G[j][k] =
// G[k][j] = sum * 2^-E and row / column 0 and the diagonal from the column sums; else Gi[j][k] = sum (all-reduced and
// assembled by the caller).
__global__ __launch_bounds__(256) void k_gram_tiled_reduce(const unsigned long long* __restrict__ part, int d, int ngroups,
                                                           const double* __restrict__ a0, const double* __restrict__ nm,
                                                           const double* __restrict__ sc, double* __restrict__ G,
                                                           i64* __restrict__ Gi) {
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= (i64)d * d) return;
  const int j = (int)(t / d), k = (int)(t % d);
  if (j >= 1 && j < k) {
    const int nt = (d + GT - 1) / GT;
    const int A = j / GT, B = k / GT;
    const int p = A * nt - A * (A - 1) / 2 + (B - A);
    const unsigned long long* src = part + (size_t)p * ngroups * (GT * GT) + (size_t)(j - A * GT) * GT + (k - B * GT);
    unsigned long long v = 0ull;
    for (int g = 0; g < ngroups; g++) v += src[(size_t)g * (GT * GT)];
    if (G) { const double x = (double)(i64)v * sc[1]; G[(i64)j * d + k] = x; G[(i64)k * d + j] = x; }
    else Gi[t] = (i64)v;
  } else if (G) {
    if (j == k) G[t] = nm[j];
    else if (j == 0) G[t] = a0[k];
    else if (k == 0) G[t] = a0[j];
  } else Gi[t] = 0;
}

// 64 cells x 4 partial lanes per block, 8 loads in flight per lane (one lane walking all 256 partials of a cell
// took 66 us for 21 MB)
__global__ __launch_bounds__(256) void k_gram_reduce(unsigned long long* __restrict__ Gi, int d, int nblk) {
  __shared__ unsigned long long sh[4][64];
  const int dd = d * d;
  const int cx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + cx;
  const int j = t < dd ? t / d : 0, k = t < dd ? t % d : 0;
  const bool upper = t < dd && j >= 1 && j < k;
  const size_t ntri = (size_t)gram_tri_count(d);
  const unsigned long long* part = Gi + dd + (upper ? gram_tri_index(j, k, d) : 0);
  unsigned long long v = 0ull;
  if (upper) {
    for (int b = 1 + ly; b <= nblk; b += 32) {
      unsigned long long x[8];
#pragma unroll
      for (int u = 0; u < 8; u++) x[u] = (b + 4 * u <= nblk) ? part[(size_t)(b + 4 * u - 1) * ntri] : 0ull;
      v += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    }
  }
  sh[ly][cx] = v;
  __syncthreads();
  if (ly == 0 && t < dd) Gi[t] = (sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx]);
}

// Single rank: k_gram_reduce and k_gram_assemble in one launch.  Only the upper triangle j < k (j >= 1) carries
// integers; cell (j, k) sums its partials (same order as k_gram_reduce) and writes G[j][k] and G[k][j]; row /
// column 0 and the diagonal come from the column sums.
__global__ __launch_bounds__(256) void k_gram_reduce_assemble(const unsigned long long* __restrict__ Gi, int d, int nblk,
                                                              const double* __restrict__ a0, const double* __restrict__ nm,
                                                              const double* __restrict__ sc, double* __restrict__ G) {
  __shared__ unsigned long long sh[4][64];
  const int dd = d * d;
  const int cx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + cx;
  const int j = t < dd ? t / d : 0, k = t < dd ? t % d : 0;
  const bool upper = t < dd && j >= 1 && j < k;
  const size_t ntri = (size_t)gram_tri_count(d);
  const unsigned long long* part = Gi + dd + (upper ? gram_tri_index(j, k, d) : 0);
  unsigned long long v = 0ull;
  if (upper) {
    for (int b = 1 + ly; b <= nblk; b += 32) {
      unsigned long long x[8];
#pragma unroll
      for (int u = 0; u < 8; u++) x[u] = (b + 4 * u <= nblk) ? part[(size_t)(b + 4 * u - 1) * ntri] : 0ull;
      v += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    }
  }
  sh[ly][cx] = v;
  __syncthreads();
  if (ly != 0 || t >= dd) return;
  if (upper) {
    const double g = (double)(i64)((sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx])) * sc[1];
    G[(i64)j * d + k] = g; G[(i64)k * d + j] = g;
  } else if (j == k) G[t] = nm[j];
  else if (j == 0) G[t] = a0[k];
  else if (k == 0) G[t] = a0[j];
}

// =================================================================================================
// Dense-panel Gram on the matrix cores (north_star: "MFMA used only if the selected-feature block is materialised as a
// dense panel"; the reference's dense inputs are the scoresLr tables, every column stored: scoresLr_data.go:108-117).
// When every row of the reduced matrix holds all d entries, its CSR value array IS the row-major n x d panel X, and
// G = X' diag(w) X is a GEMM with K = n: v_mfma_f64_16x16x4_f64, D[i][j] += sum_k A[i][k] B[k][j] with
// A[i][k] = w_k x[k][16 ta + i], B[k][j] = x[k][16 tb + j] (lane l: A row l & 15, k = l >> 4; B column l & 15, k = l >> 4;
// C/D column l & 15, row (l >> 4) + 4 reg).  A workgroup stages GD_ROWS rows of the panel (and w) in LDS, its 16
// wavefronts share the upper-triangle tiles; row blocks are dealt to the workgroups; every workgroup writes its partial
// tiles and k_gram_dense_reduce adds them in block order (deterministic) and mirrors.  The sparse builds above spend one
// LDS atomic per pair: n d^2 / 2 of them on dense rows (5e8 at n = 100k, d = 101).
// =================================================================================================
#define GD_ROWS 32
#define GD_MAXD 256
typedef double gd_double4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k_gram_dense_mfma(const double* __restrict__ X, const double* __restrict__ w, i64 n, int d,
                                                          double* __restrict__ part) {
  extern __shared__ double gd_lds[];
  const int T = (d + 15) >> 4, dpad = T * 16 + 2;        // + 2 doubles: rows of a fragment fall into different banks
  double* xs = gd_lds;                                   // GD_ROWS x dpad
  double* ws = gd_lds + GD_ROWS * dpad;                  // GD_ROWS
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntile = T * (T + 1) / 2;
  constexpr int MAXOWN = (GD_MAXD / 16) * (GD_MAXD / 16 + 1) / 2 / 16 + 1;     // 9 tiles per wavefront at d = 256
  gd_double4 acc[MAXOWN];
  int ta_[MAXOWN], tb_[MAXOWN];
  int nown = 0;
  {
    int q = 0;
    for (int ta = 0; ta < T; ta++) for (int tb = ta; tb < T; tb++, q++)
      if ((q & 15) == wave && nown < MAXOWN) { ta_[nown] = ta; tb_[nown] = tb; nown++; }
  }
#pragma unroll
  for (int u = 0; u < MAXOWN; u++) acc[u] = gd_double4{0.0, 0.0, 0.0, 0.0};
  const i64 nblk = (n + GD_ROWS - 1) / GD_ROWS;
  for (i64 blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const i64 r0 = blk * GD_ROWS;
    __syncthreads();                                     // the previous chunk has been consumed
    for (int idx = tid; idx < GD_ROWS * dpad; idx += 1024) {
      const int row = idx / dpad, c = idx - row * dpad;
      xs[idx] = (c < d && r0 + row < n) ? X[(r0 + row) * d + c] : 0.0;
    }
    if (tid < GD_ROWS) ws[tid] = (r0 + tid < n) ? w[r0 + tid] : 0.0;
    __syncthreads();
    const int kk = lane >> 4, cc = lane & 15;
#pragma unroll
    for (int u = 0; u < MAXOWN; u++) {
      if (u < nown) {
        const int ca = ta_[u] * 16 + cc, cb = tb_[u] * 16 + cc;
#pragma unroll
        for (int k = 0; k < GD_ROWS; k += 4) {
          const double a = xs[(k + kk) * dpad + ca] * ws[k + kk];
          const double b = xs[(k + kk) * dpad + cb];
          acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
        }
      }
    }
  }
  // partial tiles, raw accumulator layout: [block][tile q][reg][lane]
  {
    int q = 0, u = 0;
    for (int ta = 0; ta < T; ta++) for (int tb = ta; tb < T; tb++, q++)
      if ((q & 15) == wave && u < MAXOWN) {
        double* o = part + ((size_t)blockIdx.x * ntile + q) * 256;
        o[lane] = acc[u].x; o[64 + lane] = acc[u].y; o[128 + lane] = acc[u].z; o[192 + lane] = acc[u].w;
        u++;
      }
  }
}

// cell (a, b), a <= b: sum of its partials in block order; row / column 0 and the diagonal from the column sums like
// k_gram_assemble (a0 = sum w x, nm = sum w x^2: the operands the solver uses elsewhere); G symmetric
__global__ __launch_bounds__(256) void k_gram_dense_reduce(const double* __restrict__ part, int nblk, int d, const double* __restrict__ a0,
                                                           const double* __restrict__ nm, int use_sums, double* __restrict__ G) {
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= (i64)d * d) return;
  const int a = (int)(t / d), b = (int)(t % d);
  if (a > b) return;
  double v;
  if (use_sums && a == b) v = nm[a];
  else if (use_sums && a == 0) v = a0[b];
  else {
    const int T = (d + 15) >> 4, ntile = T * (T + 1) / 2;
    const int ta = a >> 4, tb = b >> 4, i = a & 15, j = b & 15;
    const int q = ta * T - ta * (ta - 1) / 2 + (tb - ta);
    const int lane = j + 16 * (i & 3), reg = i >> 2;                    // C/D: column = lane & 15, row = (lane >> 4) + 4 reg
    v = 0.0;
    for (int k = 0; k < nblk; k++) v += part[((size_t)k * ntile + q) * 256 + reg * 64 + lane];
  }
  G[(i64)a * d + b] = v;
  G[(i64)b * d + a] = v;
}

// several ranks: after the all-reduce of the local partial Grams, row / column 0 and the diagonal from the all-reduced sums
__global__ __launch_bounds__(256) void k_gram_dense_fix(const double* __restrict__ a0, const double* __restrict__ nm, int d, double* __restrict__ G) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= d) return;
  G[(i64)j * d + j] = nm[j];
  if (j > 0) { G[j] = a0[j]; G[(i64)j * d] = a0[j]; }
}

// G from its parts: row/column 0 and the diagonal from the column sweeps, the rest from the integers
__global__ __launch_bounds__(256) void k_gram_assemble(const double* __restrict__ a0, const double* __restrict__ nm,
                                                       const i64* __restrict__ Gi, const double* __restrict__ sc, int d,
                                                       double* __restrict__ G) {
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= (i64)d * d) return;
  const int j = (int)(t / d), k = (int)(t % d);
  double v;
  if (j == k) v = nm[j];
  else if (j == 0) v = a0[k];
  else if (k == 0) v = a0[j];
  else v = (double)(j < k ? Gi[(i64)j * d + k] : Gi[(i64)k * d + j]) * sc[1];
  G[t] = v;
}

struct CdArgs {
  const double* G; const double* xy; int d;
  double* theta;              // in/out (device)
  double l1reg, l2reg, eps;
  int K;
  double* hist;               // K*d iterates
  double* delta;              // K eval_stopping deltas
  int* stopf;                 // K eval_stopping verdicts
  int* ndone;
  int g_in_lds;
  double* q_io;               // k_cd_wave: q = G theta is stored here at exit and, with q_load, taken from here at entry
  int q_load;
};

// One workgroup.  LDS: theta[d], q[d], diag[d], xyv[d] (+ G[d*d] when it fits); up to d = 4900.
__global__ __launch_bounds__(CD_THREADS) void k_cd_sweeps(CdArgs a) {
  extern __shared__ double lds[];
  const int d = a.d, tid = threadIdx.x;
  double* th = lds;
  double* q = th + d;
  double* dg = q + d;
  double* xyv = dg + d;
  double* Gl = xyv + d;
  __shared__ double sh_delta;
  __shared__ int sh_j, sh_flag;
  const double* G = a.G;
  if (a.g_in_lds) {
    for (i64 i = tid; i < (i64)d * d; i += CD_THREADS) Gl[i] = a.G[i];
    G = Gl;
  }
  for (int i = tid; i < d; i += CD_THREADS) { th[i] = a.theta[i]; xyv[i] = a.xy[i]; dg[i] = a.G[(i64)i * d + i]; }
  __syncthreads();
  // q = G theta (skipping zero coefficients)
  for (int i = tid; i < d; i += CD_THREADS) {
    double s = 0.0;
    for (int k = 0; k < d; k++) { const double t = th[k]; if (t != 0.0) s += G[(i64)k * d + i] * t; }
    q[i] = s;
  }
  __syncthreads();
  int done = 0;
  for (int k = 0; k < a.K; k++) {
    double max_x = 0.0, max_delta = 0.0;   // thread 0 only
    int isnan_ = 0;
    int j = 0;
    while (true) {
      // thread 0 walks coordinates until one moves
      if (tid == 0) {
        double dl = 0.0;
        for (; j < d; j++) {
          const double old = th[j];
          double t = xyv[j] + dg[j] * old - q[j];                 // coordinate.go:56-59
          if (j > 0) {                                            // :60-67 soft threshold
            const double m = fmax(fabs(t) - a.l1reg, 0.0);
            t = (t >= 0.0) ? m : -m;
          }
          t /= (dg[j] + a.l2reg);                                 // :69
          if (t != t) isnan_ = 1;
          th[j] = t;                                              // :70-71
          max_x = fmax(max_x, fabs(t));
          dl = t - old;
          max_delta = fmax(max_delta, fabs(dl));
          if (dl != 0.0 || t != t) break;
        }
        sh_j = j; sh_delta = dl;
      }
      __syncthreads();
      const int jj = sh_j;
      const double dl = sh_delta;
      if (jj >= d) break;
      for (int i = tid; i < d; i += CD_THREADS) q[i] += dl * G[(i64)jj * d + i];
      __syncthreads();
      j = jj + 1;
    }
    // record the iterate and eval_stopping (proximal.go:30-52)
    for (int i = tid; i < d; i += CD_THREADS) a.hist[(i64)k * d + i] = th[i];
    if (tid == 0) {
      double delta; int stop;
      if (isnan_) { delta = nan(""); stop = 1; }
      else {
        delta = (max_x != 0.0) ? max_delta / max_x : max_delta;
        stop = ((max_x != 0.0 && max_delta / max_x <= a.eps) || (max_x == 0.0 && max_delta == 0.0)) ? 1 : 0;
      }
      a.delta[k] = delta; a.stopf[k] = stop; sh_flag = stop;
    }
    done = k + 1;
    __syncthreads();
    if (sh_flag) break;
  }
  for (int i = tid; i < d; i += CD_THREADS) a.theta[i] = th[i];
  if (tid == 0) *a.ndone = done;
}

// Wave-synchronous variant for d <= 64*QPL: ONE wavefront, no barriers.  q = G theta, theta, xy and the
// diagonal live in registers (QPL values per lane; coordinate i on lane i & 63, slot i >> 6) and reach all
// lanes through v_readlane (j is wave-uniform); row j of G is requested at the top of the step so its
// latency hides behind the scalar update.  Same arithmetic and order as k_cd_sweeps.
__device__ __forceinline__ double bcast_lane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// Row i of the Gram matrix: from LDS (address space known at compile time -> ds_read) or from global memory
template <bool GLDS>
__device__ __forceinline__ double gram_at(const double* __restrict__ Gg, i64 idx) {
  extern __shared__ double lds[];
  return GLDS ? lds[idx] : Gg[idx];
}

// One coordinate step (coordinate.go:56-69) on slot S0 of q.  G / C / O / I are this step's operands (row j of the
// Gram matrix, c_j, the old theta_j, 1/(norm_j + l2)), GN / CN / ON / IN receive those of step j+1, requested
// first so that their LDS latency hides behind the chain.  The chain q_j -> t -> delta -> q is branch free (7
// dependent f64 operations); a coordinate whose value does not move gives delta = 0 and q += 0 * row.
#define CD_STEP(S0, J, L1, G, C, O, I, GN, CN, ON, IN)                                                        \
  {                                                                                                            \
    const int jn_ = (J) + 1 < d ? (J) + 1 : (J);                                                               \
    _Pragma("unroll") for (int s_ = 0; s_ < QPL; s_++) GN[s_] = gram_at<GLDS>(Gg, (i64)jn_ * d + off[s_]);      \
    const double2 p01_ = *reinterpret_cast<const double2*>(pk + 4 * jn_);                                      \
    CN = p01_.x; ON = p01_.y; IN = pk[4 * jn_ + 2];                                                            \
    const double qj_ = bcast_lane(q[S0], (J) & 63);                                                            \
    const double t0_ = C - qj_;                                   /* coordinate.go:56-59 */                    \
    const double m_ = fmax(fabs(t0_) - (L1), 0.0);                /* :60-67 soft threshold */                  \
    const double tn_ = copysign(m_, t0_) * I;                     /* :69 (reciprocal) */                       \
    const double dl_ = tn_ - O;                                                                                \
    _Pragma("unroll") for (int s_ = 0; s_ < QPL; s_++) q[s_] = fma(dl_, G[s_], q[s_]);                          \
    thl[(J)] = tn_;                                               /* every lane, same value, same address */   \
  }

// The same step with the Gram row requested TWO steps ahead (three register sets in rotation): rows that come from
// L2 (GLDS = false, d > 128) take 500-700 cycles, a step ~250 -- one step of distance left every step waiting.
#define CD_STEP3(S0, J, L1, G, GN2, C, O, I, CN, ON, IN)                                                      \
  {                                                                                                            \
    const int j2_ = (J) + 2 < d ? (J) + 2 : d - 1;                                                             \
    const char* rowb_ = reinterpret_cast<const char*>(Gg) + (i64)j2_ * d * 8;      /* uniform row base */           \
    _Pragma("unroll") for (int s_ = 0; s_ < QPL; s_++)                                                            \
      GN2[s_] = *reinterpret_cast<const double*>(rowb_ + lane8 + s_ * 512);     /* columns >= d: never used */   \
    const int jn_ = (J) + 1 < d ? (J) + 1 : (J);                                                               \
    const double2 p01_ = *reinterpret_cast<const double2*>(pk + 4 * jn_);                                      \
    CN = p01_.x; ON = p01_.y; IN = pk[4 * jn_ + 2];                                                            \
    const double qj_ = bcast_lane(q[S0], (J) & 63);                                                            \
    const double t0_ = C - qj_;                                                                                \
    const double m_ = fmax(fabs(t0_) - (L1), 0.0);                                                             \
    const double tn_ = copysign(m_, t0_) * I;                                                                  \
    const double dl_ = tn_ - O;                                                                                \
    _Pragma("unroll") for (int s_ = 0; s_ < QPL; s_++) q[s_] = fma(dl_, G[s_], q[s_]);                          \
    thl[(J)] = tn_;                                                                                            \
  }

template <int QPL, bool GLDS, bool PF2 = false>
__global__ __launch_bounds__(256) void k_cd_wave(CdArgs a) {
  // 256 threads stage the Gram matrix into LDS (one wave alone would spend ~1 us per dependent load);
  // wavefront 0 then runs the sweeps.  A single wavefront cannot hide anything: every instruction of a step
  // costs its issue slots (the first version spent ~75 instructions = ~360 cycles per coordinate, four times the
  // length of the dependency chain), so the step is kept to the chain plus one prefetch:
  //   * per-coordinate operands {c_j = xy_j + norm_j theta_j, old theta_j, 1/(norm_j + l2)} are packed in LDS at
  //     the start of a sweep (theta_j only changes at its own step) and fetched one step ahead;
  //   * row reads are unguarded (lanes past d read column d-1 again; their q slots are never used);
  //   * two register sets alternate (no copies), the q slot is static (the j loop is split at multiples of 64);
  //   * max|theta|, max|delta| and the NaN check of eval_stopping are lane-parallel work after the sweep.
  // LDS: [G d*d (GLDS)] | pk[4d] | theta[d].  Same arithmetic and order as k_cd_sweeps.
  extern __shared__ double lds[];
  const int d = a.d, lane = threadIdx.x & 63;
  const double* __restrict__ Gg = a.G;
  double* pk = lds + (GLDS ? (size_t)d * d : 0);
  double* thl = pk + 4 * (size_t)d;
  if (GLDS) {
    const i64 dd = (i64)d * d;
#pragma unroll 8
    for (i64 i = threadIdx.x; i < dd; i += 256) lds[i] = a.G[i];
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;                       // the sweeps are one wavefront's work
  double q[QPL], xy[QPL], dg[QPL], inv[QPL];
  int off[QPL];
#pragma unroll
  for (int s = 0; s < QPL; s++) {
    const int i = s * 64 + lane;
    off[s] = i < d ? i : d - 1;
    xy[s] = i < d ? a.xy[i] : 0.0;
    dg[s] = i < d ? a.G[(i64)i * d + i] : 1.0;
    inv[s] = 1.0 / (dg[s] + a.l2reg);                  // :69 as a reciprocal (<= 1 ulp)
    q[s] = 0.0;
    if (i < d) thl[i] = a.theta[i];
  }
  __builtin_amdgcn_wave_barrier();                     // single wave from here on: LDS ops are in order
  if (a.q_load) {
    // q comes from the previous batch of the same inner loop, or from k_gtheta: one wavefront walking all d rows
    // of a Gram matrix in L2 for every batch was 150 us at d = 501
#pragma unroll
    for (int s = 0; s < QPL; s++) { const int i = s * 64 + lane; q[s] = i < d ? a.q_io[i] : 0.0; }
  } else {
    // q = G theta (skipping zero coefficients)
    for (int k = 0; k < d; k++) {
      const double t = thl[k];
      if (t != 0.0) {
#pragma unroll
        for (int s = 0; s < QPL; s++) q[s] += gram_at<GLDS>(Gg, (i64)k * d + off[s]) * t;
      }
    }
  }
  int done = 0;
  for (int k = 0; k < a.K; k++) {
    double told[QPL];
#pragma unroll
    for (int s = 0; s < QPL; s++) {
      const int i = s * 64 + lane;
      told[s] = 0.0;
      if (i < d) {
        const double t = thl[i];
        told[s] = t;
        pk[4 * i] = xy[s] + dg[s] * t;                 // :56
        pk[4 * i + 1] = t;
        pk[4 * i + 2] = inv[s];
      }
    }
    __builtin_amdgcn_wave_barrier();
    const double l1 = a.l1reg;
    if constexpr (PF2) {
      // d >= 129 here: slot 0 is full.  Sets X / Y / Z hold rows j, j+1, j+2 in rotation; operands c/o/i one step ahead.
      // unclamped row reads (scalar row base + lane * 8 + slot * 512 as an immediate: no address arithmetic per load):
      // lanes past d run into the next row -- or, on the last row, up to 512 * QPL - 8 d bytes past the matrix, into
      // the integer Gram that follows it in the same workspace block (d * d * 8 bytes); their q slots are never read
      const unsigned lane8 = (unsigned)lane * 8u;
      double gX[QPL], gY[QPL], gZ[QPL], cX, oX, iX, cY, oY, iY, cZ, oZ, iZ;
#pragma unroll
      for (int s = 0; s < QPL; s++) { gX[s] = Gg[off[s]]; gY[s] = Gg[(i64)(d > 1 ? 1 : 0) * d + off[s]]; }
      cX = pk[0]; oX = pk[1]; iX = pk[2];
#pragma unroll
      for (int s0 = 0; s0 < QPL; s0++) {
        const int j0 = s0 * 64, j1 = d < j0 + 64 ? d : j0 + 64;
        int j = j0;
        if (s0 == 0) {                                               // the bias is not penalised (:60)
          CD_STEP3(s0, 0, 0.0, gX, gZ, cX, oX, iX, cY, oY, iY);
          CD_STEP3(s0, 1, l1, gY, gX, cY, oY, iY, cZ, oZ, iZ);
          CD_STEP3(s0, 2, l1, gZ, gY, cZ, oZ, iZ, cX, oX, iX);
          j = 3;
        }
        for (; j + 2 < j1; j += 3) {
          CD_STEP3(s0, j, l1, gX, gZ, cX, oX, iX, cY, oY, iY);
          CD_STEP3(s0, j + 1, l1, gY, gX, cY, oY, iY, cZ, oZ, iZ);
          CD_STEP3(s0, j + 2, l1, gZ, gY, cZ, oZ, iZ, cX, oX, iX);
        }
        if (j + 1 == j1) {                                           // one step left: rotate the sets by one
          CD_STEP3(s0, j, l1, gX, gZ, cX, oX, iX, cY, oY, iY);
#pragma unroll
          for (int s = 0; s < QPL; s++) { gX[s] = gY[s]; gY[s] = gZ[s]; }
          cX = cY; oX = oY; iX = iY;
        } else if (j + 2 == j1) {                                    // two steps left: rotate by two
          CD_STEP3(s0, j, l1, gX, gZ, cX, oX, iX, cY, oY, iY);
          CD_STEP3(s0, j + 1, l1, gY, gX, cY, oY, iY, cZ, oZ, iZ);
#pragma unroll
          for (int s = 0; s < QPL; s++) { const double t = gX[s]; gX[s] = gZ[s]; gY[s] = t; }
          cX = cZ; oX = oZ; iX = iZ;
        }
      }
    } else {
      double gA[QPL], gB[QPL], cA, oA, iA, cB, oB, iB;
  #pragma unroll
      for (int s = 0; s < QPL; s++) gA[s] = gram_at<GLDS>(Gg, off[s]);
      cA = pk[0]; oA = pk[1]; iA = pk[2];
  #pragma unroll
      for (int s0 = 0; s0 < QPL; s0++) {
        const int j0 = s0 * 64, j1 = d < j0 + 64 ? d : j0 + 64;     // an even number of steps unless it is the last slot
        int j = j0;
        if (s0 == 0 && j < j1) {                                      // the bias is not penalised (:60)
          CD_STEP(s0, j, 0.0, gA, cA, oA, iA, gB, cB, oB, iB);
          j++;
          if (j < j1) { CD_STEP(s0, j, l1, gB, cB, oB, iB, gA, cA, oA, iA); j++; }
          else {                                                      // d == 1: keep the sets aligned
  #pragma unroll
            for (int s = 0; s < QPL; s++) gA[s] = gB[s];
            cA = cB; oA = oB; iA = iB;
          }
        }
        for (; j + 1 < j1; j += 2) {
          CD_STEP(s0, j, l1, gA, cA, oA, iA, gB, cB, oB, iB);
          CD_STEP(s0, j + 1, l1, gB, cB, oB, iB, gA, cA, oA, iA);
        }
        if (j < j1) {                                                 // odd tail (last slot only): one step, sets swapped back
          CD_STEP(s0, j, l1, gA, cA, oA, iA, gB, cB, oB, iB);
  #pragma unroll
          for (int s = 0; s < QPL; s++) gA[s] = gB[s];
          cA = cB; oA = oB; iA = iB;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // eval_stopping (proximal.go:30-52) from the old and the new iterate, lane parallel
    double max_x = 0.0, max_delta = 0.0;
    int isnan_ = 0;
#pragma unroll
    for (int s = 0; s < QPL; s++) {
      const int i = s * 64 + lane;
      if (i < d) {
        const double tn = thl[i];
        a.hist[(i64)k * d + i] = tn;
        isnan_ |= (tn != tn) ? 1 : 0;
        max_x = fmax(max_x, fabs(tn));
        max_delta = fmax(max_delta, fabs(tn - told[s]));
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      max_x = fmax(max_x, __shfl_xor(max_x, o, 64));
      max_delta = fmax(max_delta, __shfl_xor(max_delta, o, 64));
      isnan_ |= __shfl_xor(isnan_, o, 64);
    }
    double delta; int stop;
    if (isnan_) { delta = nan(""); stop = 1; }
    else {
      delta = (max_x != 0.0) ? max_delta / max_x : max_delta;
      stop = ((max_x != 0.0 && max_delta / max_x <= a.eps) || (max_x == 0.0 && max_delta == 0.0)) ? 1 : 0;
    }
    if (lane == 0) { a.delta[k] = delta; a.stopf[k] = stop; }
    done = k + 1;
    if (stop) break;
  }
#pragma unroll
  for (int s = 0; s < QPL; s++) { const int i = s * 64 + lane; if (i < d) { a.theta[i] = thl[i]; if (a.q_io) a.q_io[i] = q[s]; } }
  if (lane == 0) *a.ndone = done;
}
#undef CD_STEP
#undef CD_STEP3

// ---- d > 512: blocked Gauss-Seidel over 16 wavefronts ----------------------------------------------------------
// Coordinates in blocks of 64; q = G theta lives in registers, slot s (coordinates 64 s .. 64 s + 63) on wavefront
// s % 16.  For block B the wavefront that owns it runs the 64 dependent steps ALONE, against the 64 x 64 diagonal
// block of G staged in LDS (the step of k_cd_lds2 with one q value per lane) and leaves the 64 deltas in LDS; after
// one barrier every wavefront adds  sum_j delta_j G[64 B + j][.]  to its own slots, j ascending -- the very fma
// sequence a coordinate-at-a-time sweep applies to those q values, which nobody reads in between: same results,
// but the other slots' updates (all the memory traffic: one pass over G per sweep) run beside the next block's
// dependent steps instead of inside them.  The wavefront that owns block B+1 stages its diagonal block while it
// would otherwise wait for block B's steps.  k_cd_sweeps walked the coordinates on one thread with two barriers per
// moving coordinate: 568 us per sweep at d = 1001.
#define CDB_ROWS 65                                  // 64 rows + the harmless prefetch of "row 64"
#define CDB_STEP(JL, L1, G, C, O, I, GN, CN, ON, IN)                                                          \
  {                                                                                                            \
    GN = grow[0]; grow += 64;                                                                                  \
    const double2 p01_ = *reinterpret_cast<const double2*>(pj + 4);                                            \
    CN = p01_.x; ON = p01_.y; IN = pj[6];                                                                      \
    const double qj_ = bcast_lane(qb, (JL));                                                                   \
    const double t0_ = C - qj_;                                                                                \
    const double m_ = fmax(fabs(t0_) - (L1), 0.0);                                                             \
    const double tn_ = copysign(m_, t0_) * I;                                                                  \
    const double dl_ = tn_ - O;                                                                                \
    qb = fma(dl_, G, qb);                                                                                      \
    pj[3] = tn_; pj[0] = dl_;                                     /* c_j is spent: its place takes the delta */ \
    pj += 4;                                                                                                   \
  }

// rows `first`, first + step, ... of the 64 x 64 diagonal block of block Bn into dst (one wavefront per call; the
// helpers of a block share the 64 rows so that each waits for one round trip, not eight)
__device__ __forceinline__ void cdb_stage_diag(const double* __restrict__ G, int d, int Bn, int lane, int first, int step,
                                               double* __restrict__ dst) {
  const int c0 = Bn * 64 + lane, cc = c0 < d ? c0 : d - 1;
#pragma unroll 5
  for (int r = first; r < 64; r += step) {
    const int row = Bn * 64 + r < d ? Bn * 64 + r : d - 1;
    dst[r * 64 + lane] = G[(i64)row * d + cc];
  }
}

template <int QW>
__global__ __launch_bounds__(1024) void k_cd_blocked(CdArgs a) {
  __shared__ double Gd[2][CDB_ROWS * 64];            // diagonal blocks, double buffered
  __shared__ double pk[2][CDB_ROWS * 4];             // {c_j -> delta_j, old theta_j, 1/(norm_j + l2), theta_j}
  __shared__ double sh_mx[16], sh_md[16];
  __shared__ int sh_nan[16], sh_stop;
  const int d = a.d, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int S = (d + 63) >> 6;                       // blocks / slots
  const double* __restrict__ G = a.G;
  double q[QW], th[QW], xy[QW], dg[QW], inv[QW];
#pragma unroll
  for (int t = 0; t < QW; t++) {
    const int i = (w + 16 * t) * 64 + lane;
    const bool ok = i < d;
    th[t] = ok ? a.theta[i] : 0.0;
    xy[t] = ok ? a.xy[i] : 0.0;
    dg[t] = ok ? G[(i64)i * d + i] : 1.0;
    inv[t] = 1.0 / (dg[t] + a.l2reg);
    q[t] = ok ? a.q_io[i] : 0.0;                     // q = G theta from k_gtheta or the previous batch
  }
  if (threadIdx.x < 64) { Gd[0][64 * 64 + lane] = 0.0; Gd[1][64 * 64 + lane] = 0.0; }
  if (threadIdx.x < 8) { pk[0][64 * 4 + (threadIdx.x & 3)] = 0.0; pk[1][64 * 4 + (threadIdx.x & 3)] = 0.0; }
  cdb_stage_diag(G, d, 0, lane, w, 16, Gd[0]);           // diagonal block of block 0, four rows per wavefront
  __syncthreads();
  int zero_v;
  asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));
  const double l1 = a.l1reg;
  int done = 0;
  for (int k = 0; k < a.K; k++) {
    double told[QW];
#pragma unroll
    for (int t = 0; t < QW; t++) told[t] = th[t];
    for (int B = 0; B < S; B++) {
      const int ow = B & 15, bb = (k * S + B) & 1;         // LDS buffers alternate over ALL blocks of the batch
      if (w == ow) {
        // ---- the 64 dependent steps of block B ----
        const int nbk = d - B * 64 < 64 ? d - B * 64 : 64;             // coordinates in this block
        double qb = 0.0, thb = 0.0, xyb = 0.0, dgb = 1.0, invb = 1.0;
#pragma unroll
        for (int t = 0; t < QW; t++) if (B == ow + 16 * t) { qb = q[t]; thb = th[t]; xyb = xy[t]; dgb = dg[t]; invb = inv[t]; }
        double* pkb = pk[bb];
        pkb[4 * lane] = xyb + dgb * thb; pkb[4 * lane + 1] = thb; pkb[4 * lane + 2] = invb; pkb[4 * lane + 3] = thb;   // :56
        __builtin_amdgcn_wave_barrier();
        const double* grow = Gd[bb] + 64 + lane;                        // row 1 of the diagonal block
        double* pj = pkb + zero_v;
        double gA = Gd[bb][lane], cA = pkb[0], oA = pkb[1], iA = pkb[2], gB, cB, oB, iB;
        int j = 0;
        if (B == 0) {                                                   // the bias is not penalised (:60)
          CDB_STEP(0, 0.0, gA, cA, oA, iA, gB, cB, oB, iB);
          j = 1;
          if (j < nbk) { CDB_STEP(1, l1, gB, cB, oB, iB, gA, cA, oA, iA); j = 2; }
          else { gA = gB; cA = cB; oA = oB; iA = iB; }
        }
        for (; j + 1 < nbk; j += 2) {
          CDB_STEP(j, l1, gA, cA, oA, iA, gB, cB, oB, iB);
          CDB_STEP(j + 1, l1, gB, cB, oB, iB, gA, cA, oA, iA);
        }
        if (j < nbk) { CDB_STEP(j, l1, gA, cA, oA, iA, gB, cB, oB, iB); }
        __builtin_amdgcn_wave_barrier();
        // coordinates past the end of the last block: no move
        if (lane >= nbk) pkb[4 * lane] = 0.0;
        const double thn = pkb[4 * lane + 3];
#pragma unroll
        for (int t = 0; t < QW; t++) if (B == ow + 16 * t) { q[t] = qb; th[t] = thn; }
      } else {
        // ---- meanwhile the fifteen others stage the diagonal block of the next block (block 0 of the next sweep after
        // the last), four or five rows each ----
        cdb_stage_diag(G, d, B + 1 == S ? 0 : B + 1, lane, (w - ow - 1) & 15, 15, Gd[bb ^ 1]);
      }
      __syncthreads();                                                  // deltas of block B visible
      // ---- every wavefront: its own slots += sum_j delta_j G[64 B + j][slot], j ascending ----
      // One compute unit reads all of G once per sweep (64 bytes per clock: 3.4 us per block at d = 1001, next to
      // 4.2 us of steps).  Letting the next owner update first and alone and the others stream beside its steps was
      // slower (318 ms against 300 for the lambda-auto=1000 run): the streaming wavefronts share the SIMDs with the
      // one that runs the dependent steps.
      const double* dlp = pk[bb];
      const int nbk = d - B * 64 < 64 ? d - B * 64 : 64;
#pragma unroll
      for (int t = 0; t < QW; t++) {
        const int sl = w + 16 * t;
        if (sl < S && sl != B) {
          const int ci = sl * 64 + lane;
          const double* col = G + (i64)B * 64 * d + (ci < d ? ci : d - 1);
          double qq = q[t];
          int j = 0;
          for (; j + 8 <= nbk; j += 8) {
            double g[8];
#pragma unroll
            for (int u = 0; u < 8; u++) g[u] = col[(i64)(j + u) * d];
#pragma unroll
            for (int u = 0; u < 8; u++) qq = fma(dlp[4 * (j + u)], g[u], qq);
          }
          for (; j < nbk; j++) qq = fma(dlp[4 * j], col[(i64)j * d], qq);
          q[t] = qq;
        }
      }
    }
    // ---- eval_stopping (proximal.go:30-52), the iterate ----
    double max_x = 0.0, max_delta = 0.0; int isnan_ = 0;
#pragma unroll
    for (int t = 0; t < QW; t++) {
      const int i = (w + 16 * t) * 64 + lane;
      if (i < d) {
        const double tn = th[t];
        a.hist[(i64)k * d + i] = tn;
        isnan_ |= (tn != tn) ? 1 : 0;
        max_x = fmax(max_x, fabs(tn));
        max_delta = fmax(max_delta, fabs(tn - told[t]));
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      max_x = fmax(max_x, __shfl_xor(max_x, o, 64));
      max_delta = fmax(max_delta, __shfl_xor(max_delta, o, 64));
      isnan_ |= __shfl_xor(isnan_, o, 64);
    }
    if (lane == 0) { sh_mx[w] = max_x; sh_md[w] = max_delta; sh_nan[w] = isnan_; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double mx = 0.0, md = 0.0; int nn = 0;
      for (int t = 0; t < 16; t++) { mx = fmax(mx, sh_mx[t]); md = fmax(md, sh_md[t]); nn |= sh_nan[t]; }
      double delta; int stop;
      if (nn) { delta = nan(""); stop = 1; }
      else {
        delta = (mx != 0.0) ? md / mx : md;
        stop = ((mx != 0.0 && md / mx <= a.eps) || (mx == 0.0 && md == 0.0)) ? 1 : 0;
      }
      a.delta[k] = delta; a.stopf[k] = stop; sh_stop = stop;
    }
    done = k + 1;
    __syncthreads();
    if (sh_stop) break;
  }
#pragma unroll
  for (int t = 0; t < QW; t++) {
    const int i = (w + 16 * t) * 64 + lane;
    if (i < d) { a.theta[i] = th[t]; a.q_io[i] = q[t]; }
  }
  if (threadIdx.x == 0) *a.ndone = done;
}

// ---- d > 2048: the same blocked sweep cut into super-blocks of 8 blocks (512 coordinates), two launches each ------
// One compute unit streaming all of G per sweep is 2.35 ms at d = 5001 (200 MB).  k_cd_sb_steps runs the blocked
// sweep INSIDE super-block sb (its 8 q slots, the 512 x 512 diagonal part of G) and leaves the 512 deltas in global
// memory; k_cd_sb_update adds  sum_j delta_j G[512 sb + j][.]  (j ascending) to every other q slot, a wavefront per
// slot on as many compute units as there are slots -- again the fma sequence of the coordinate-at-a-time sweep;
// k_cd_sb_stop closes a sweep (eval_stopping, the iterate count, the halt flag the later launches of the batch obey).
#define SBW 8
__global__ __launch_bounds__(1024) void k_cd_sb_steps(CdArgs a, int k, int sb, const int* __restrict__ halt,
                                                      double* __restrict__ dlbuf) {
  __shared__ double Gd[2][CDB_ROWS * 64];
  __shared__ double pk[2][CDB_ROWS * 4];
  if (*halt) return;
  const int d = a.d, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int S = (d + 63) >> 6;
  const int B0 = sb * SBW, nbl = S - B0 < SBW ? S - B0 : SBW;          // blocks of this super-block
  const double* __restrict__ G = a.G;
  const int mi = (B0 + w) * 64 + lane;                                // my coordinate (waves 0 .. nbl-1)
  const bool mine = w < nbl && mi < d;
  double q = mine ? a.q_io[mi] : 0.0, th = mine ? a.theta[mi] : 0.0;
  const double xy = mine ? a.xy[mi] : 0.0, dg = mine ? G[(i64)mi * d + mi] : 1.0;
  const double inv = 1.0 / (dg + a.l2reg);
  if (threadIdx.x < 64) { Gd[0][64 * 64 + lane] = 0.0; Gd[1][64 * 64 + lane] = 0.0; }
  if (threadIdx.x < 8) { pk[0][64 * 4 + (threadIdx.x & 3)] = 0.0; pk[1][64 * 4 + (threadIdx.x & 3)] = 0.0; }
  cdb_stage_diag(G, d, B0, lane, w, 16, Gd[0]);          // diagonal block of the first block, four rows per wavefront
  __syncthreads();
  int zero_v;
  asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));
  const double l1 = a.l1reg;
  for (int bl = 0; bl < nbl; bl++) {
    const int B = B0 + bl, bb = bl & 1;
    const int nbk = d - B * 64 < 64 ? d - B * 64 : 64;
    if (w == bl) {
      double qb = q;
      double* pkb = pk[bb];
      pkb[4 * lane] = xy + dg * th; pkb[4 * lane + 1] = th; pkb[4 * lane + 2] = inv; pkb[4 * lane + 3] = th;   // :56
      __builtin_amdgcn_wave_barrier();
      const double* grow = Gd[bb] + 64 + lane;
      double* pj = pkb + zero_v;
      double gA = Gd[bb][lane], cA = pkb[0], oA = pkb[1], iA = pkb[2], gB, cB, oB, iB;
      int j = 0;
      if (B == 0) {                                                   // the bias is not penalised (:60)
        CDB_STEP(0, 0.0, gA, cA, oA, iA, gB, cB, oB, iB);
        j = 1;
        if (j < nbk) { CDB_STEP(1, l1, gB, cB, oB, iB, gA, cA, oA, iA); j = 2; }
        else { gA = gB; cA = cB; oA = oB; iA = iB; }
      }
      for (; j + 1 < nbk; j += 2) {
        CDB_STEP(j, l1, gA, cA, oA, iA, gB, cB, oB, iB);
        CDB_STEP(j + 1, l1, gB, cB, oB, iB, gA, cA, oA, iA);
      }
      if (j < nbk) { CDB_STEP(j, l1, gA, cA, oA, iA, gB, cB, oB, iB); }
      __builtin_amdgcn_wave_barrier();
      if (lane >= nbk) pkb[4 * lane] = 0.0;
      dlbuf[bl * 64 + lane] = pkb[4 * lane];                           // for k_cd_sb_update
      q = qb; th = pkb[4 * lane + 3];
    } else if (bl + 1 < nbl) {
      // the fifteen others stage the next block's diagonal block meanwhile
      cdb_stage_diag(G, d, B + 1, lane, (w - bl - 1) & 15, 15, Gd[bb ^ 1]);
    }
    __syncthreads();
    if (w < nbl && w != bl) {                                          // the other slots of the super-block
      const double* dlp = pk[bb];
      const int ci = (B0 + w) * 64 + lane;
      const double* col = G + (i64)B * 64 * d + (ci < d ? ci : d - 1);
      int j = 0;
      for (; j + 8 <= nbk; j += 8) {
        double g[8];
#pragma unroll
        for (int u = 0; u < 8; u++) g[u] = col[(i64)(j + u) * d];
#pragma unroll
        for (int u = 0; u < 8; u++) q = fma(dlp[4 * (j + u)], g[u], q);
      }
      for (; j < nbk; j++) q = fma(dlp[4 * j], col[(i64)j * d], q);
    }
  }
  if (mine) { a.q_io[mi] = q; a.theta[mi] = th; a.hist[(i64)k * d + mi] = th; }
}

__global__ __launch_bounds__(256) void k_cd_sb_update(const double* __restrict__ G, int d, int sb,
                                                      const double* __restrict__ dlbuf, double* __restrict__ q,
                                                      const int* __restrict__ halt) {
  if (*halt) return;
  const int S = (d + 63) >> 6, lane = threadIdx.x & 63;
  int sl = blockIdx.x * 4 + (threadIdx.x >> 6);                        // slots outside the super-block
  if (sl >= sb * SBW) sl += SBW;
  if (sl >= S) return;
  const int ci = sl * 64 + lane;
  const int nj = d - sb * SBW * 64 < SBW * 64 ? d - sb * SBW * 64 : SBW * 64;
  const double* col = G + (i64)sb * SBW * 64 * d + (ci < d ? ci : d - 1);
  double qq = ci < d ? q[ci] : 0.0;
  int j = 0;
  for (; j + 16 <= nj; j += 16) {
    double g[16];
#pragma unroll
    for (int u = 0; u < 16; u++) g[u] = col[(i64)(j + u) * d];
#pragma unroll
    for (int u = 0; u < 16; u++) qq = fma(dlbuf[j + u], g[u], qq);
  }
  for (; j < nj; j++) qq = fma(dlbuf[j], col[(i64)j * d], qq);
  if (ci < d) q[ci] = qq;
}

__global__ __launch_bounds__(1024) void k_cd_sb_stop(CdArgs a, int k, const double* __restrict__ prev, int* __restrict__ halt) {
  __shared__ double sh_mx[16], sh_md[16];
  __shared__ int sh_nan[16];
  if (*halt) return;
  const int d = a.d, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const double* cur = a.hist + (i64)k * d;
  double max_x = 0.0, max_delta = 0.0; int isnan_ = 0;
  for (int i = threadIdx.x; i < d; i += 1024) {
    const double tn = cur[i];
    isnan_ |= (tn != tn) ? 1 : 0;
    max_x = fmax(max_x, fabs(tn));
    max_delta = fmax(max_delta, fabs(tn - prev[i]));
  }
  for (int o = 32; o > 0; o >>= 1) {
    max_x = fmax(max_x, __shfl_xor(max_x, o, 64));
    max_delta = fmax(max_delta, __shfl_xor(max_delta, o, 64));
    isnan_ |= __shfl_xor(isnan_, o, 64);
  }
  if (lane == 0) { sh_mx[w] = max_x; sh_md[w] = max_delta; sh_nan[w] = isnan_; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double mx = 0.0, md = 0.0; int nn = 0;
    for (int t = 0; t < 16; t++) { mx = fmax(mx, sh_mx[t]); md = fmax(md, sh_md[t]); nn |= sh_nan[t]; }
    double delta; int stop;
    if (nn) { delta = nan(""); stop = 1; }
    else {
      delta = (mx != 0.0) ? md / mx : md;
      stop = ((mx != 0.0 && md / mx <= a.eps) || (mx == 0.0 && md == 0.0)) ? 1 : 0;
    }
    a.delta[k] = delta; a.stopf[k] = stop; *a.ndone = k + 1;
    if (stop) *halt = 1;
  }
}
#undef CDB_STEP

// q = G theta (zero coefficients skipped), a workgroup per 64 coordinates: four wavefronts each sum a quarter of the
// rows (k ascending inside a quarter), the quarters are added in a fixed order.  (One wavefront per 64 coordinates
// walking all d rows took 370 us at d = 5001.)
__global__ __launch_bounds__(256) void k_gtheta(const double* __restrict__ G, const double* __restrict__ theta, int d,
                                                double* __restrict__ q) {
  __shared__ double sh[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int ic = i < d ? i : d - 1;
  const int per = (d + 3) / 4, k0 = w * per, k1 = k0 + per < d ? k0 + per : d;
  double s = 0.0;
  // loads unconditional and eight in flight (a load behind the `t != 0` branch is one L2 round trip per k); the zero
  // coefficients are still skipped in the sum
  int k = k0;
  for (; k + 8 <= k1; k += 8) {
    double g[8], t[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { g[u] = G[(i64)(k + u) * d + ic]; t[u] = theta[k + u]; }
#pragma unroll
    for (int u = 0; u < 8; u++) { const double v = s + g[u] * t[u]; s = t[u] != 0.0 ? v : s; }
  }
  for (; k < k1; k++) {
    const double t = theta[k];
    const double v = s + G[(i64)k * d + ic] * t;
    s = t != 0.0 ? v : s;
  }
  sh[w][lane] = s;
  __syncthreads();
  if (w == 0 && i < d) q[i] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}

// d <= 128: the same wave-synchronous sweeps with every operand in LDS at padded strides, so that a step is its
// arithmetic chain and little else (a single wavefront pays ~4-8 cycles of issue per instruction of ANY kind; the
// k_cd_wave step spent 29 instructions, twelve of them scalar address arithmetic and clamps, for a chain of 9):
//   * Gram rows staged at a stride of 128 doubles, columns >= d and row d zero: row j+1 is read without a clamp at
//     [row pointer] and [row pointer + 512 bytes], the pointer a per-lane register that advances by one add;
//   * {c_j, old theta_j, 1/(norm_j + l2), theta_j} packed 32 bytes per coordinate (entry d a dummy): one pointer
//     (kept in a vector register on purpose: a uniform address would be re-materialised from scalar registers for
//     every LDS instruction) serves the two operand reads of step j+1 and the theta_j write of step j.
// Same arithmetic, same order, same results as k_cd_wave<2, true>.  LDS: (d+1) * 132 doubles <= 136 KiB.
#define CDL_STRIDE 128
#define CDL_STEP(Q, JL, L1, G0, G1, C, O, I, GN0, GN1, CN, ON, IN)                                            \
  {                                                                                                            \
    GN0 = grow[0]; GN1 = grow[64]; grow += CDL_STRIDE;                                                        \
    const double2 p01_ = *reinterpret_cast<const double2*>(pj + 4);                                            \
    CN = p01_.x; ON = p01_.y; IN = pj[6];                                                                      \
    const double qj_ = bcast_lane(Q, (JL));                                                                    \
    const double t0_ = C - qj_;                                   /* coordinate.go:56-59 */                    \
    const double m_ = fmax(fabs(t0_) - (L1), 0.0);                /* :60-67 soft threshold */                  \
    const double tn_ = copysign(m_, t0_) * I;                     /* :69 (reciprocal) */                       \
    const double dl_ = tn_ - O;                                                                                \
    q0 = fma(dl_, G0, q0); q1 = fma(dl_, G1, q1);                                                              \
    pj[3] = tn_;                                                  /* every lane, same value, same address */   \
    pj += 4;                                                                                                   \
  }

struct CdTheta { int n; double v[128]; };     // n > 0: the start iterate travels in the kernel arguments (no upload)

struct CdlLane { double xy0, xy1, dg0, dg1, inv0, inv1; };

// stage the Gram matrix and the start iterate (all threads of the block), then __syncthreads
__device__ __forceinline__ void cdl_stage(const CdArgs& a, const CdTheta& th0, double* Gl, double* pk) {
  const int d = a.d;
  // ALL loads of a thread in flight at once (d <= 128: at most 17): the Gram matrix was written a moment ago by workgroups
  // all over the chip, so the first batch of a re-linearisation fetches it from memory -- one trip instead of four
  const int total = (d + 1) * CDL_STRIDE;
  constexpr int NLD = ((128 + 1) * CDL_STRIDE + 1023) / 1024;
  double v[NLD];
#pragma unroll
  for (int u = 0; u < NLD; u++) {
    const int t = (int)threadIdx.x + u * 1024, j = t >> 7, i = t & (CDL_STRIDE - 1);
    v[u] = (t < total && j < d && i < d) ? a.G[(i64)j * d + i] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < NLD; u++) { const int t = (int)threadIdx.x + u * 1024; if (t < total) Gl[t] = v[u]; }
  for (int t = threadIdx.x; t < 4 * (d + 1); t += blockDim.x)
    pk[t] = ((t & 3) == 3 && (t >> 2) < d) ? (th0.n > 0 ? th0.v[t >> 2] : a.theta[t >> 2]) : 0.0;
  __syncthreads();
}

// per-lane constants and q = G theta (skipping zero coefficients); one wavefront
__device__ __forceinline__ void cdl_init(const CdArgs& a, int lane, const double* Gl, const double* pk, double xy0, double xy1,
                                         CdlLane& c, double& q0, double& q1) {
  const int d = a.d, i0 = lane, i1 = 64 + lane;
  c.xy0 = xy0; c.xy1 = xy1;
  c.dg0 = i0 < d ? Gl[(size_t)i0 * CDL_STRIDE + i0] : 1.0; c.dg1 = i1 < d ? Gl[(size_t)i1 * CDL_STRIDE + i1] : 1.0;
  c.inv0 = 1.0 / (c.dg0 + a.l2reg); c.inv1 = 1.0 / (c.dg1 + a.l2reg);      // :69 as a reciprocal (<= 1 ulp)
  q0 = 0.0; q1 = 0.0;
  if (a.q_load) return;                                    // a later batch of the same inner loop: q comes from q_io
  // theta in registers (lane l: coordinates l and 64 + l), picked out with v_readlane: the chain is the two additions,
  // not an LDS round trip per coordinate.  Zero coefficients are skipped as before (same sums, same order).
  const double tA = i0 < d ? pk[4 * i0 + 3] : 0.0, tB = i1 < d ? pk[4 * i1 + 3] : 0.0;
  for (int k0 = 0; k0 < d; k0 += 4) {
    double g0[4], g1[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {                          // row d is staged as zeros
      const int kk = k0 + u < d ? k0 + u : d;
      g0[u] = Gl[(size_t)kk * CDL_STRIDE + i0]; g1[u] = Gl[(size_t)kk * CDL_STRIDE + i1];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 + u;
      const double t = k < d ? bcast_lane(k < 64 ? tA : tB, k & 63) : 0.0;
      const double a0 = q0 + g0[u] * t, a1 = q1 + g1[u] * t;
      q0 = t != 0.0 ? a0 : q0; q1 = t != 0.0 ? a1 : q1;
    }
  }
}

// one sweep over all coordinates; told = the iterate before it (this lane's two coordinates)
__device__ __forceinline__ void cdl_sweep(int d, int lane, const double* Gl, double* pk, const CdlLane& c, double l1,
                                          int zero_v, double& q0, double& q1, double& told0, double& told1) {
  const int i0 = lane, i1 = 64 + lane;
  told0 = 0.0; told1 = 0.0;
  if (i0 < d) { const double t = pk[4 * i0 + 3]; told0 = t; pk[4 * i0] = c.xy0 + c.dg0 * t; pk[4 * i0 + 1] = t; pk[4 * i0 + 2] = c.inv0; }   // :56
  if (i1 < d) { const double t = pk[4 * i1 + 3]; told1 = t; pk[4 * i1] = c.xy1 + c.dg1 * t; pk[4 * i1 + 1] = t; pk[4 * i1 + 2] = c.inv1; }
  __builtin_amdgcn_wave_barrier();
  double gA0 = Gl[i0], gA1 = Gl[i1], cA = pk[0], oA = pk[1], iA = pk[2];
  double gB0, gB1, cB, oB, iB;
  const double* grow = Gl + CDL_STRIDE + lane;        // row 1
  double* pj = pk + zero_v;                           // coordinate 0
  {
    const int j1 = d < 64 ? d : 64;
    int j = 0;
    CDL_STEP(q0, j, 0.0, gA0, gA1, cA, oA, iA, gB0, gB1, cB, oB, iB);     // the bias is not penalised (:60)
    j++;
    if (j < j1) { CDL_STEP(q0, j, l1, gB0, gB1, cB, oB, iB, gA0, gA1, cA, oA, iA); j++; }
    else { gA0 = gB0; gA1 = gB1; cA = cB; oA = oB; iA = iB; }            // d == 1: keep the sets aligned
    for (; j + 1 < j1; j += 2) {
      CDL_STEP(q0, j, l1, gA0, gA1, cA, oA, iA, gB0, gB1, cB, oB, iB);
      CDL_STEP(q0, j + 1, l1, gB0, gB1, cB, oB, iB, gA0, gA1, cA, oA, iA);
    }
    if (j < j1) {
      CDL_STEP(q0, j, l1, gA0, gA1, cA, oA, iA, gB0, gB1, cB, oB, iB);
      gA0 = gB0; gA1 = gB1; cA = cB; oA = oB; iA = iB;
    }
  }
  {
    int j = 64;
    for (; j + 1 < d; j += 2) {
      CDL_STEP(q1, j - 64, l1, gA0, gA1, cA, oA, iA, gB0, gB1, cB, oB, iB);
      CDL_STEP(q1, j - 63, l1, gB0, gB1, cB, oB, iB, gA0, gA1, cA, oA, iA);
    }
    if (j < d) {
      CDL_STEP(q1, j - 64, l1, gA0, gA1, cA, oA, iA, gB0, gB1, cB, oB, iB);
      gA0 = gB0; gA1 = gB1; cA = cB; oA = oB; iA = iB;
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// eval_stopping (proximal.go:30-52) from the old and the new iterate, lane parallel; also the nonzero count
__device__ __forceinline__ void cdl_stop(int d, int lane, const double* pk, double told0, double told1, double eps,
                                         double& tn0, double& tn1, double& delta, int& stop, int& nz) {
  const int i0 = lane, i1 = 64 + lane;
  double max_x = 0.0, max_delta = 0.0;
  int isnan_ = 0;
  tn0 = 0.0; tn1 = 0.0;
  if (i0 < d) {
    tn0 = pk[4 * i0 + 3];
    isnan_ |= (tn0 != tn0) ? 1 : 0;
    max_x = fmax(max_x, fabs(tn0));
    max_delta = fmax(max_delta, fabs(tn0 - told0));
  }
  if (i1 < d) {
    tn1 = pk[4 * i1 + 3];
    isnan_ |= (tn1 != tn1) ? 1 : 0;
    max_x = fmax(max_x, fabs(tn1));
    max_delta = fmax(max_delta, fabs(tn1 - told1));
  }
  nz = __popcll(__ballot(i0 < d && tn0 != 0.0)) + __popcll(__ballot(i1 < d && tn1 != 0.0));
  for (int o = 32; o > 0; o >>= 1) {
    max_x = fmax(max_x, __shfl_xor(max_x, o, 64));
    max_delta = fmax(max_delta, __shfl_xor(max_delta, o, 64));
    isnan_ |= __shfl_xor(isnan_, o, 64);
  }
  if (isnan_) { delta = nan(""); stop = 1; }
  else {
    delta = (max_x != 0.0) ? max_delta / max_x : max_delta;
    stop = ((max_x != 0.0 && max_delta / max_x <= eps) || (max_x == 0.0 && max_delta == 0.0)) ? 1 : 0;
  }
}

__global__ __launch_bounds__(1024) void k_cd_lds2(CdArgs a, CdTheta th0) {
  extern __shared__ double lds[];
  const int d = a.d, lane = threadIdx.x & 63;
  double* Gl = lds;                                         // (d+1) x 128
  double* pk = lds + (size_t)(d + 1) * CDL_STRIDE;          // (d+1) x 4
  const int i0 = lane, i1 = 64 + lane;
  // this lane's xy and (a later batch of the same inner loop) q are requested beside the Gram matrix, not after it
  const double xy0 = i0 < d ? a.xy[i0] : 0.0, xy1 = i1 < d ? a.xy[i1] : 0.0;
  double qin0 = 0.0, qin1 = 0.0;
  if (a.q_load) { qin0 = i0 < d ? a.q_io[i0] : 0.0; qin1 = i1 < d ? a.q_io[i1] : 0.0; }
  cdl_stage(a, th0, Gl, pk);
  if (threadIdx.x >= 64) return;                       // the sweeps are one wavefront's work
  CdlLane c; double q0, q1;
  cdl_init(a, lane, Gl, pk, xy0, xy1, c, q0, q1);
  // a later batch of the same inner loop continues with the q it left (re-forming q = G theta from theta rounds
  // differently: the results would depend on where the batches end)
  if (a.q_load) { q0 = qin0; q1 = qin1; }
  int zero_v;
  asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));      // opaque: keeps the operand pointer in a vector register
  int done = 0;
  for (int k = 0; k < a.K; k++) {
    double told0, told1, tn0, tn1, delta; int stop, nz;
    cdl_sweep(d, lane, Gl, pk, c, a.l1reg, zero_v, q0, q1, told0, told1);
    cdl_stop(d, lane, pk, told0, told1, a.eps, tn0, tn1, delta, stop, nz);
    if (i0 < d) a.hist[(i64)k * d + i0] = tn0;
    if (i1 < d) a.hist[(i64)k * d + i1] = tn1;
    if (lane == 0) { a.delta[k] = delta; a.stopf[k] = stop; }
    done = k + 1;
    if (stop) break;
  }
  if (i0 < d) { a.theta[i0] = pk[4 * i0 + 3]; if (a.q_io) a.q_io[i0] = q0; }
  if (i1 < d) { a.theta[i1] = pk[4 * i1 + 3]; if (a.q_io) a.q_io[i1] = q1; }
  if (lane == 0) *a.ndone = done;
}

#undef CDL_STEP

// log-loss of up to KSPEC iterates in one pass over the reduced rows: lane (row, k) -- 8 lanes per row, one per
// iterate -- so the exp / log1p of the KSPEC iterates of a row run side by side instead of one after the other (a
// thread per row spent most of its 30 us there); the entries of a row are same-address loads for its 8 lanes; the
// iterates are staged in LDS as [column][k].  32 rows per block, block partials in a fixed order.
#define LM_THREADS 1024                 // 128 rows per block: the iterates are staged once per 128 rows
#define LM_ROWS (LM_THREADS / KSPEC)
// TPB: threads per block -- 1024 for a batch (the iterates are staged once per 128 rows), 256 for a probe batch of one
// sweep (a lane per row: 1024 rows per block would leave 98 blocks for 256 CUs at configs[2])
template <bool LDS, int TPB = LM_THREADS>
__global__ __launch_bounds__(TPB) void k_loss_multi(const i64* __restrict__ rowptr, const u32* __restrict__ col,
                                                    const double* __restrict__ val, const uint8_t* __restrict__ labels,
                                                    const uint8_t* __restrict__ active,
                                                    i64 n, int d, const double* __restrict__ hist, const int* __restrict__ ndone,
                                                    double cw0, double cw1, double* __restrict__ blk, int kl_log2) {
  // kl = 1 << kl_log2 lanes per row (the batch holds at most kl iterates): a probe batch of one sweep keeps every
  // lane on a row of its own -- the exp / log1p chain costs a wavefront the same whether 1 or 8 of 8 lanes need it
  extern __shared__ double lm_lds[];            // kl * d iterates, layout [column][k]
  __shared__ double sh[KSPEC][TPB / 64];
  const int K = *ndone;
  const int kl = 1 << kl_log2;
  if (LDS) {
    for (int t = threadIdx.x; t < (d << kl_log2); t += TPB) {           // [column][kl]: only the batch's iterates are staged
      const int c = t >> kl_log2, k = t & (kl - 1);
      lm_lds[t] = k < K ? hist[(i64)k * d + c] : 0.0;
    }
    __syncthreads();
  }
  const int k = threadIdx.x & (kl - 1);
  const i64 i = (i64)blockIdx.x * (TPB >> kl_log2) + (threadIdx.x >> kl_log2);
  double term = 0.0;
  if (i < n && k < K && (!active || active[i])) {
    const i64 b = rowptr[i], e = rowptr[i + 1];
    const int y = labels[i] != 0;
    double r = 0.0;
    // four entries requested at a time (a row is otherwise one dependent round trip per entry); added in entry order
    for (i64 p = b + 1; p < e; p += 4) {
      u32 c4[4]; double v4[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const bool h = p + u < e; c4[u] = h ? col[p + u] : 0u; v4[u] = h ? val[p + u] : 0.0; }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (p + u < e) r += v4[u] * (LDS ? lm_lds[((size_t)c4[u] << kl_log2) + k] : hist[(i64)k * d + c4[u]]);
    }
    const double rr = (LDS ? lm_lds[k] : hist[(i64)k * d]) + r;            // theta[0] + sum
    const double l1p = log1p(exp(-fabs(rr)));
    // y: -log sigma(r) = LogAdd(0,-r);  !y: LogAdd(0,r)
    term = y ? cw1 * (fmax(-rr, 0.0) + l1p) : cw0 * (fmax(rr, 0.0) + l1p);
  }
  // rows of the wave: lanes k, k + kl, ...
  double v = term;
  for (int o = 32; o >= kl; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((int)(threadIdx.x & 63) < kl) sh[k][threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x < KSPEC) {
    double acc = 0.0;
    if ((int)threadIdx.x < kl) for (int ww = 0; ww < TPB / 64; ww++) acc += sh[threadIdx.x][ww];
    blk[(i64)blockIdx.x * KSPEC + threadIdx.x] = acc;
  }
}

__global__ __launch_bounds__(1024) void k_sum_partials_multi(const double* __restrict__ blk, i64 nb, double* __restrict__ out) {
  // block k sums column k of blk[nb][KSPEC] in a fixed order
  __shared__ double sh[1024];
  const int k = blockIdx.x;
  double v = 0.0;
  for (i64 i = threadIdx.x; i < nb; i += 1024) v += blk[i * KSPEC + k];
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[k] = sh[0];
}

static bool host_eval_stopping(const double* xs, const double* x1, i64 len, double eps, double* delta_out) {
  double max_x = 0.0, max_delta = 0.0;
  for (i64 i = 0; i < len; i++) {
    if (std::isnan(x1[i])) { *delta_out = NAN; return true; }
    max_x = std::fmax(max_x, std::fabs(x1[i]));
    max_delta = std::fmax(max_delta, std::fabs(x1[i] - xs[i]));
  }
  *delta_out = (max_x != 0.0) ? max_delta / max_x : max_delta;
  return (max_x != 0.0 && max_delta / max_x <= eps) || (max_x == 0.0 && max_delta == 0.0);
}

extern "C" int klr_cd_solve(klr_ctx* ctx, const klr_mat* red, double* theta, const double cw[2],
                            double l1reg, double l2reg, double eps, double eps_loss, int eval_loss,
                            int64_t max_iter, klr_iter_rec* recs, int64_t recs_cap, int64_t* n_recs) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, red != nullptr && theta != nullptr && cw != nullptr);
  KLR_ARG(ctx, red->val_f64);   // reduced matrices hold f64 values (select_columns output)
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  const i64 n = red->n;
  const int d = (int)red->dim;
  const i64 n_total = red->n_total;
  if (max_iter <= 0) max_iter = INT32_MAX;
  const size_t lds_small = (size_t)4 * d * sizeof(double);
  const size_t lds_big = lds_small + (size_t)d * d * sizeof(double);
  const size_t LDS_MAX = 160 * 1024 - 1024;
  if (d > 16384)
    return klr_fail(ctx, KLR_EINVAL, "%d selected features exceed the device solver's limit (16383)", d - 1);
  const int g_in_lds = lds_big <= LDS_MAX;
  const size_t lds_bytes = g_in_lds ? lds_big : lds_small;
  // wave variant: packed per-coordinate operands (4 per coordinate), theta (+ G) in LDS
  const size_t lds_wave = (size_t)5 * d * sizeof(double) + (g_in_lds ? (size_t)d * d * sizeof(double) : 0);
  KLR_TRY(ctx_func_lds(ctx, (const void*)k_cd_sweeps, (int)LDS_MAX));
  KLR_TRY(ctx_func_lds(ctx, (const void*)k_cd_lds2, (int)LDS_MAX));
  KLR_TRY(ctx_func_lds(ctx, (const void*)k_gram_offdiag<true>, 128 * 1024));

  hipStream_t st = ctx->stream;
  double *w, *z, *G, *vec, *lossb;
  i64* Gi;
  KLR_TRY(ws_get(ctx, WS_S, (size_t)std::max<i64>(n, 1) * 8, (void**)&w));
  KLR_TRY(ws_get(ctx, WS_Z, (size_t)std::max<i64>(n, 1) * 8, (void**)&z));
  // LDS-privatised Gram build: one partial per workgroup after the result slot
  // ... unless the rows are so short that there are few pairs in all (k-mer data: 2-3 stored entries per reduced
  // row): then zeroing, writing and re-reading one 80 KiB partial per workgroup is the whole cost (150 us per
  // re-weighting at configs[2]) and plain global integer atomics are the cheaper build
  const double est_pairs = (double)n * std::max(0.0, red->avg_row - 1.0) * std::max(0.0, red->avg_row - 2.0) * 0.5 + (double)n;
  // KLR_GRAM_PATH=lds|global forces the privatised (LDS / tiled) or the global-atomics build (tests)
  const char* gram_env = ctx_opt(ctx, "KLR_GRAM_PATH");
  const bool gram_many = gram_env ? (gram_env[0] == 'l') : est_pairs > 4e6;
  // one Gram row per workgroup once there are enough columns to fill the chip with workgroups (KLR_GRAM_ROWS_MIN)
  const int gram_rows_min = ctx_opt(ctx, "KLR_GRAM_ROWS_MIN") ? atoi(ctx_opt(ctx, "KLR_GRAM_ROWS_MIN")) : 384;
  const bool gram_rows = d >= gram_rows_min && gram_many && (size_t)d * 8 <= 128 * 1024 && red->d_collen != nullptr;
  const bool gram_lds = !gram_rows && (size_t)d * d * 8 <= 128 * 1024 && gram_many;
  const i64 gram_blocks_cap = ctx_opt(ctx, "KLR_GRAM_BLOCKS") ? std::max(1, atoi(ctx_opt(ctx, "KLR_GRAM_BLOCKS"))) : ctx->n_cu;   // A/B tooling
  const unsigned gram_blocks = gram_lds ? (unsigned)std::max<i64>(1, std::min<i64>(gram_blocks_cap, ceil_div(n * 16, 1024))) : 0u;
  // 128 < d <= 1024 with many pairs: tile pairs privatised in LDS (k_gram_offdiag_tiled)
  const int gram_nt = (d + GT - 1) / GT, gram_npairs = gram_nt * (gram_nt + 1) / 2;
  // every row stores all d columns (dense score tables): the value array is the n x d panel -> Gram on the matrix cores
  const bool gram_dense_ok = red->nnz == n * (i64)d && d >= 2 && d <= GD_MAXD && n >= 256 && !ctx_opt(ctx, "KLR_GRAM_NO_MFMA");
  // ... decided TOGETHER by all ranks (below, with the first re-linearisation's all-reduce): a shard that lacks one entry
  // of a selected column, or has fewer than 256 rows, would post the int64 Gram all-reduce against its neighbours' f64 one
  const bool gram_dense_local = gram_dense_ok && (!gram_env || gram_env[0] == 'm');
  bool gram_dense = gram_dense_local;
  const bool gram_dense_vote = ctx->ar_fn != nullptr && ctx->nranks > 1 && d <= GD_MAXD;   // every rank knows d: all vote or none
  const unsigned gram_dense_blocks = gram_dense_local ? (unsigned)std::max<i64>(1, std::min<i64>(ctx->n_cu, ceil_div(n, (i64)GD_ROWS))) : 0u;
  const bool gram_tiled = !gram_lds && !gram_rows && (size_t)d * d * 8 > 128 * 1024 && d <= 1024 && gram_many;
  const unsigned gram_groups = gram_tiled ? (unsigned)std::max<i64>(1, std::min<i64>(std::max(1, ctx->n_cu / gram_npairs), ceil_div(n, 2 * 1024))) : 0u;
  if (gram_tiled) {
    KLR_TRY(ctx_func_lds(ctx, (const void*)k_gram_offdiag_tiled, GT * GT * 8));
  }
  const int gd_T = (d + 15) / 16;
  const size_t gram_dense_bytes = gram_dense_local ? (size_t)gram_dense_blocks * (gd_T * (gd_T + 1) / 2) * 256 * 8 : 0;
  const size_t gram_part_bytes = std::max(gram_dense_bytes, gram_tiled ? (size_t)gram_npairs * gram_groups * GT * GT * 8 : (size_t)gram_blocks * d * d * 8);
  const size_t gram_toff_bytes = gram_tiled ? (size_t)n * (gram_nt + 1) * sizeof(unsigned short) : 0;
  KLR_TRY(ws_get(ctx, WS_CD_G, (size_t)2 * d * d * 8 + gram_part_bytes + gram_toff_bytes, (void**)&G));
  unsigned short* gram_toff = (unsigned short*)((char*)G + (size_t)2 * d * d * 8 + gram_part_bytes);
  if (gram_tiled && n)
    hipLaunchKernelGGL(k_row_tile_offsets, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, red->rowptr, red->col, n,
                       gram_nt, gram_toff);
  Gi = (i64*)(G + (size_t)d * d);
  // vec: a0[d] nm[d] xy[d] theta[d] | hist[K*d] delta[K] losses[K] gsc[2] | ints: stopf[K] ndone
  // everything a batch of sweeps reports (hist .. ndone) is contiguous: ONE device-to-host copy per batch
  const size_t vec_bytes = ((size_t)4 * d + 2 + (size_t)KSPEC * d + 2 * KSPEC + 2) * 8 + (KSPEC + 8) * sizeof(int) + 16 + (size_t)d * 8
                           + (size_t)d * 8 + SBW * 64 * 8 + 8;
  KLR_TRY(ws_get(ctx, WS_CD_VEC, vec_bytes, (void**)&vec));
  // xy[d]: this rank's vote on the dense-panel Gram (1 = not dense here), summed with [a0 | nm | xy]
  double* a0 = vec; double* nm = a0 + d; double* xy = nm + d; double* thd = xy + d + 2; double* hist = thd + d;
  double* delta = hist + (size_t)KSPEC * d; double* losses = delta + KSPEC; double* gsc = losses + KSPEC;
  int* stopf = (int*)(gsc + 2); int* ndone = stopf + KSPEC;
  // {ticket, max bits} of k_irls_finish, which resets both: zeroed once, when the block is allocated (a memset per call
  // was 5 us at the head of every epoch)
  unsigned long long* fin_ctl;
  {
    const bool fresh = ctx->ws[WS_CD_CTL].p == nullptr;
    KLR_TRY(ws_get(ctx, WS_CD_CTL, 64, (void**)&fin_ctl));
    if (fresh) KLR_HIP(ctx, hipMemsetAsync(fin_ctl, 0, 64, st));
  }
  double* q_dev = (double*)((unsigned long long*)(stopf + KSPEC + 8) + 2);     // q = G theta between batches (k_cd_wave)
  double* sb_th0 = q_dev + d; double* sb_dl = sb_th0 + d; int* sb_halt = (int*)(sb_dl + SBW * 64);   // super-block solver
  const size_t report_bytes = ((size_t)KSPEC * d + 2 * KSPEC + 2) * 8 + (KSPEC + 1) * sizeof(int);
  char* pin_report;
  KLR_TRY(pin_get(ctx, report_bytes, (void**)&pin_report));
  const i64 nb = std::max<i64>(1, ceil_div(n, (i64)LM_ROWS));          // blocks of k_loss_multi
  KLR_TRY(ws_get(ctx, WS_CD_LOSS, ((size_t)nb * KSPEC + KSPEC) * 8, (void**)&lossb));

  std::vector<double> th(theta, theta + d), tp(d), hhist((size_t)KSPEC * d), hdelta(KSPEC), hloss(KSPEC);
  std::vector<int> hstop(KSPEC + 1);
  // the device copy of theta is refreshed lazily: a small iterate (d <= 128) rides in the kernel arguments of the
  // next row pass / first batch of sweeps instead of a pageable upload per rewind
  bool thd_stale = true;
  auto sync_thd = [&]() -> int {
    if (thd_stale) { KLR_TRY(h2d(ctx, thd, th.data(), (size_t)d * 8)); thd_stale = false; }
    return KLR_OK;
  };
  const bool inline_theta = d <= 128;
  if (!inline_theta) KLR_TRY(sync_thd());

  const auto t0 = std::chrono::steady_clock::now();
  const double lambda = l1reg / (double)n_total;
  double loss_old = NAN, loss_new = NAN, loss_outer_prev = NAN;
  i64 count = 0;
  auto full_loss = [&](const double* t, double sum) {
    double r = n_total ? sum / (double)n_total : 0.0;
    if (!std::isnan(lambda) && lambda != 0.0) for (int j = 1; j < d; j++) r += lambda * std::fabs(t[j]);
    return r;
  };

  int relin = 0;                   // re-linearisations so far
  // accepted sweeps per re-linearisation of the PREVIOUS solve of this run (klr_solve_run_begin); this solve leaves its own
  std::vector<int> cd_prev;
  if (ctx->cd_pred_on) { cd_prev.swap(ctx->cd_pred); ctx->cd_pred.clear(); }
  const int first_k = ctx_opt(ctx, "KLR_CD_FIRST_BATCH") ? std::max(1, std::min(KSPEC, atoi(ctx_opt(ctx, "KLR_CD_FIRST_BATCH")))) : 1;
  // sweeps per speculative batch: a batch costs ~45 us beside its sweeps (launches, loss pass, report, host turn) and
  // wastes (K-1)/2 sweeps on average when the loss rule stops it; a sweep is ~8 us at d = 101 and ~67 us at d = 501:
  // small problems want full batches, large ones shorter ones (measured at d = 501: 8 -> 190 ms, 4 -> 180, 2 -> 187)
  const int batch_k = ctx_opt(ctx, "KLR_CD_BATCH") ? std::max(1, std::min(KSPEC, atoi(ctx_opt(ctx, "KLR_CD_BATCH")))) : (d <= 128 ? 8 : 4);
  const bool dbg_sync = ctx_opt(ctx, "KLR_CD_DEBUG") && atoi(ctx_opt(ctx, "KLR_CD_DEBUG")) >= 2;
#define DBG_STEP(name) do { if (dbg_sync) { hipError_t e_ = hipStreamSynchronize(st); fprintf(stderr, "[cd] %s: %s\n", name, hipGetErrorString(e_)); } } while (0)
  for (i64 iter = 0; iter < max_iter; iter++, relin++) {
    // ---- IRLS linearisation + Gram --------------------------------------------------------------
    if (thd_stale) KLR_TRY(launch_row_pass(ctx, red, thd, ROW_IRLS, cw, w, z, nullptr, th.data(), d));
    else KLR_TRY(launch_row_pass(ctx, red, thd, ROW_IRLS, cw, w, z, nullptr));
    // [a0 | nm | xy] = sum_i w x | sum_i w x^2 (diagonal, norm) | sum_i w z x (inner_xy): one pass, one finish
    const bool single = ctx->ar_fn == nullptr;
    double* part3;
    KLR_TRY(launch_irls_cols(ctx, red, w, z, &part3));
    const unsigned fin_blocks = (unsigned)ceil_div((i64)d, (i64)4);
    if (single) hipLaunchKernelGGL(k_irls_finish<true>, dim3(fin_blocks), dim3(256), 0, st, red->colchunk, d, red->nchunks, part3, a0, gsc, fin_ctl);
    else {
      hipLaunchKernelGGL(k_irls_finish<false>, dim3(fin_blocks), dim3(256), 0, st, red->colchunk, d, red->nchunks, part3, a0, gsc, fin_ctl);
      if (gram_dense_vote && relin == 0) {
        const double vote = gram_dense_local ? 0.0 : 1.0;
        KLR_TRY(h2d(ctx, xy + d, &vote, sizeof(double)));
        KLR_TRY(klr_allreduce(ctx, a0, 3 * (i64)d + 1, KLR_F64));              // [a0 | nm | xy | vote] contiguous
        double votes = 1.0;
        KLR_TRY(d2h_sync(ctx, &votes, xy + d, sizeof(double)));
        gram_dense = votes == 0.0;                                             // the matrix cores only if EVERY shard is a dense panel
      } else
      KLR_TRY(klr_allreduce(ctx, a0, 3 * (i64)d, KLR_F64));                    // [a0 | nm | xy] contiguous
      hipLaunchKernelGGL(k_gram_scale, dim3(1), dim3(256), 0, st, nm, d, gsc);
    }
    DBG_STEP("irls sums");
    // the LDS-privatised build writes whole partials and the reduce overwrites the result slot: nothing to zero
    if ((!gram_lds && !gram_tiled && !gram_rows && !gram_dense) || !n) KLR_HIP(ctx, hipMemsetAsync(Gi, 0, (size_t)d * d * 8, st));
    bool assembled = false;
    if (n && gram_dense) {
      KLR_TRY(ctx_func_lds(ctx, (const void*)k_gram_dense_mfma, 96 * 1024));
      double* gpart = (double*)(Gi + (size_t)d * d);
      const size_t lds = ((size_t)GD_ROWS * (gd_T * 16 + 2) + GD_ROWS) * 8;
      hipLaunchKernelGGL(k_gram_dense_mfma, dim3(gram_dense_blocks), dim3(1024), lds, st, (const double*)red->val, w, n, d, gpart);
      if (single) {
        hipLaunchKernelGGL(k_gram_dense_reduce, dim3((unsigned)ceil_div((i64)d * d, 256)), dim3(256), 0, st, (const double*)gpart, (int)gram_dense_blocks, d,
                           a0, nm, 1, G);
      } else {
        // local partial Gram in f64, summed over the ranks, then row / column 0 and the diagonal from the all-reduced sums
        hipLaunchKernelGGL(k_gram_dense_reduce, dim3((unsigned)ceil_div((i64)d * d, 256)), dim3(256), 0, st, (const double*)gpart, (int)gram_dense_blocks, d,
                           a0, nm, 0, G);
        KLR_TRY(klr_allreduce(ctx, G, (i64)d * d, KLR_F64));
        hipLaunchKernelGGL(k_gram_dense_fix, dim3((unsigned)ceil_div((i64)d, 256)), dim3(256), 0, st, (const double*)a0, (const double*)nm, d, G);
      }
      assembled = true;
    } else if (n) {
      if (gram_lds) {
        hipLaunchKernelGGL(k_gram_offdiag<true>, dim3(gram_blocks), dim3(1024), (size_t)d * d * 8, st, red->rowptr, red->col,
                           (const double*)red->val, w, n, d, gsc, (unsigned long long*)Gi);
        if (single) {
          hipLaunchKernelGGL(k_gram_reduce_assemble, dim3((unsigned)ceil_div((i64)d * d, 64)), dim3(256), 0, st,
                             (const unsigned long long*)Gi, d, (int)gram_blocks, a0, nm, gsc, G);
          assembled = true;
        } else
          hipLaunchKernelGGL(k_gram_reduce, dim3((unsigned)ceil_div((i64)d * d, 64)), dim3(256), 0, st, (unsigned long long*)Gi,
                             d, (int)gram_blocks);
      } else if (gram_rows) {
        KLR_TRY(ctx_func_lds(ctx, (const void*)k_gram_rows, 128 * 1024));
        hipLaunchKernelGGL(k_gram_rows, dim3((unsigned)d), dim3(1024), (size_t)d * 8, st, red->rowptr, red->col, (const double*)red->val,
                           red->colptr, red->d_collen, red->crow, (const double*)red->cval, w, d, gsc, Gi);
      } else if (gram_tiled) {
        unsigned long long* gpart = (unsigned long long*)(Gi + (size_t)d * d);
        hipLaunchKernelGGL(k_gram_offdiag_tiled, dim3(gram_groups, (unsigned)gram_npairs), dim3(1024), (size_t)GT * GT * 8, st,
                           red->rowptr, red->col, (const double*)red->val, w, gram_toff, n, d, gsc, gpart);
        hipLaunchKernelGGL(k_gram_tiled_reduce, dim3((unsigned)ceil_div((i64)d * d, 256)), dim3(256), 0, st,
                           (const unsigned long long*)gpart, d, (int)gram_groups, a0, nm, gsc, single ? G : (double*)nullptr, Gi);
        assembled = single;
      } else {
        const unsigned gb = (unsigned)std::max<i64>(1, std::min<i64>(4 * ctx->n_cu, ceil_div(n * 16, 1024)));
        hipLaunchKernelGGL(k_gram_offdiag<false>, dim3(gb), dim3(1024), 0, st, red->rowptr, red->col,
                           (const double*)red->val, w, n, d, gsc, (unsigned long long*)Gi);
      }
    }
    if (!assembled) {
      KLR_TRY(klr_allreduce(ctx, Gi, (i64)d * d, KLR_I64));
      hipLaunchKernelGGL(k_gram_assemble, dim3((unsigned)ceil_div((i64)d * d, 256)), dim3(256), 0, st, a0, nm, Gi, gsc, d, G);
    }
    KLR_HIP(ctx, hipGetLastError());
    DBG_STEP("gram");
    tp = th;

    // ---- sweeps (speculative batches of KSPEC) -----------------------------------------------------
    bool inner_done = false, first_batch = true;
    bool q_stale = true;             // a new Gram matrix
    const i64 dbg_iter0 = iter; int dbg_stopk = 0;
    double accepted_sum = NAN; bool have_sum = false;
    while (!inner_done && iter < max_iter) {
      // sweeps are launched in speculative batches; the loss rule usually ends the inner loop of a LATER
      // re-linearisation at its first sweep (the model barely moves any more): probe with one sweep first
      const bool probe = relin >= 1 && first_batch && eps_loss != 0.0;
      // ... unless the same re-linearisation of the previous solve of this run took sweeps (the early epochs of a leapfrog
      // run: 7, 10, 9, 8, ... accepted sweeps in the second re-linearisation): then that many + the stopping one
      int probe_k = first_k;
      if (probe && relin < (int)cd_prev.size() && cd_prev[(size_t)relin] > 0)
        probe_k = std::min(batch_k, cd_prev[(size_t)relin] + 1);
      first_batch = false;
      int want_k = probe ? probe_k : batch_k;
      // a later batch of a re-linearisation whose like took p sweeps last time: no more than what is left of p (+ the stopping
      // sweep); past p, full batches again
      if (!probe && relin < (int)cd_prev.size() && cd_prev[(size_t)relin] > 0) {
        const i64 left = (i64)cd_prev[(size_t)relin] + 1 - (iter - dbg_iter0);
        if (left > 0) want_k = (int)std::min<i64>(KSPEC, left);     // (also past the cautious default of large problems)
      }
      const int Kb = (int)std::min<i64>(want_k, max_iter - iter);
      CdArgs a;
      a.G = G; a.xy = xy; a.d = d; a.theta = thd; a.l1reg = l1reg; a.l2reg = l2reg; a.eps = eps; a.K = Kb;
      a.hist = hist; a.delta = delta; a.stopf = stopf; a.ndone = ndone; a.g_in_lds = g_in_lds;
      a.q_io = nullptr; a.q_load = 0;
      if (d > 128) {
        a.q_io = q_dev; a.q_load = 1;
        if (q_stale) hipLaunchKernelGGL(k_gtheta, dim3((unsigned)ceil_div((i64)d, (i64)64)), dim3(256), 0, st, G, thd, d, q_dev);
        q_stale = false;
      }
      hipEvent_t pe0 = nullptr, pe1 = nullptr;           // klr_prof_*: device time of the sweep kernels of this batch
      if (ctx->prof_on) {
        KLR_TRY(prof_event_get(ctx, &pe0)); KLR_TRY(prof_event_get(ctx, &pe1));
        KLR_HIP(ctx, hipEventRecord(pe0, st));
      }
      if (d <= 128) {
        a.q_io = q_dev; a.q_load = q_stale ? 0 : 1;       // formed in the kernel (Gram in LDS) when theta or G changed
        q_stale = false;
        CdTheta t0; t0.n = 0;
        if (thd_stale) { t0.n = d; memcpy(t0.v, th.data(), (size_t)d * 8); thd_stale = false; }   // the kernel writes thd
        hipLaunchKernelGGL(k_cd_lds2, dim3(1), dim3(1024), (size_t)(d + 1) * (CDL_STRIDE + 4) * sizeof(double), st, a, t0);
      }
      else if (d <= 256) hipLaunchKernelGGL((k_cd_wave<4, false, true>), dim3(1), dim3(256), lds_wave, st, a);
      else if (d <= 512) hipLaunchKernelGGL((k_cd_wave<8, false, true>), dim3(1), dim3(256), lds_wave, st, a);
      else if (d <= 1024 && !ctx_opt(ctx, "KLR_CD_SWEEPS")) hipLaunchKernelGGL(k_cd_blocked<1>, dim3(1), dim3(1024), 0, st, a);
      else if (d <= 2048 && !ctx_opt(ctx, "KLR_CD_SWEEPS")) hipLaunchKernelGGL(k_cd_blocked<2>, dim3(1), dim3(1024), 0, st, a);
      else if (!ctx_opt(ctx, "KLR_CD_SWEEPS") || lds_small > LDS_MAX) {
        // super-blocks: two launches per 512 coordinates, one per sweep for the stop rule; a halted batch's later
        // launches return at once
        const int S = (d + 63) / 64, NSB = (S + SBW - 1) / SBW;
        KLR_HIP(ctx, hipMemsetAsync(sb_halt, 0, 8, st));
        KLR_HIP(ctx, hipMemsetAsync(ndone, 0, sizeof(int), st));
        KLR_HIP(ctx, hipMemcpyAsync(sb_th0, thd, (size_t)d * 8, hipMemcpyDeviceToDevice, st));
        for (int k = 0; k < Kb; k++) {
          for (int sb = 0; sb < NSB; sb++) {
            hipLaunchKernelGGL(k_cd_sb_steps, dim3(1), dim3(1024), 0, st, a, k, sb, (const int*)sb_halt, sb_dl);
            hipLaunchKernelGGL(k_cd_sb_update, dim3((unsigned)ceil_div((i64)S, (i64)4)), dim3(256), 0, st, (const double*)G, d, sb,
                               (const double*)sb_dl, q_dev, (const int*)sb_halt);
          }
          hipLaunchKernelGGL(k_cd_sb_stop, dim3(1), dim3(1024), 0, st, a, k, (const double*)(k == 0 ? sb_th0 : hist + (size_t)(k - 1) * d), sb_halt);
        }
      }
      else               hipLaunchKernelGGL(k_cd_sweeps, dim3(1), dim3(CD_THREADS), lds_bytes, st, a);
      if (pe0) KLR_HIP(ctx, hipEventRecord(pe1, st));
      if (eval_loss) {
        // iterates staged in LDS only while that is cheaper than reading them where they are: a block of LM_ROWS rows
        // stages KSPEC * d values but uses ~KSPEC * 32 * (row length) of them
        int kl_log2 = 0;
        while ((1 << kl_log2) < Kb) kl_log2++;
        const bool lm_small = kl_log2 == 0 && n >= 64 * 256;         // a probe batch: blocks of 256 rows
        const i64 nbl = std::max<i64>(1, ceil_div(n, (i64)((lm_small ? 256 : LM_THREADS) >> kl_log2)));
        const size_t lm_bytes = ((size_t)d << kl_log2) * 8;
        const bool lm_lds = (double)d <= (double)(LM_THREADS >> kl_log2) * std::max(2.0, red->avg_row) && lm_bytes <= 48 * 1024;
        if (lm_small && lm_lds)
          hipLaunchKernelGGL((k_loss_multi<true, 256>), dim3((unsigned)nbl), dim3(256), lm_bytes, st, red->rowptr, red->col,
                             (const double*)red->val, red->labels, red->active, n, d, hist, ndone, cw[0], cw[1], lossb, kl_log2);
        else if (lm_small)
          hipLaunchKernelGGL((k_loss_multi<false, 256>), dim3((unsigned)nbl), dim3(256), 0, st, red->rowptr, red->col,
                             (const double*)red->val, red->labels, red->active, n, d, hist, ndone, cw[0], cw[1], lossb, kl_log2);
        else if (lm_lds)
          hipLaunchKernelGGL(k_loss_multi<true>, dim3((unsigned)nbl), dim3(LM_THREADS), lm_bytes, st, red->rowptr, red->col,
                             (const double*)red->val, red->labels, red->active, n, d, hist, ndone, cw[0], cw[1], lossb, kl_log2);
        else
          hipLaunchKernelGGL(k_loss_multi<false>, dim3((unsigned)nbl), dim3(LM_THREADS), 0, st, red->rowptr, red->col,
                             (const double*)red->val, red->labels, red->active, n, d, hist, ndone, cw[0], cw[1], lossb, kl_log2);
        hipLaunchKernelGGL(k_sum_partials_multi, dim3(KSPEC), dim3(1024), 0, st, lossb, nbl, losses);
        KLR_TRY(klr_allreduce(ctx, losses, KSPEC, KLR_F64));
      }
      KLR_HIP(ctx, hipGetLastError());
      DBG_STEP("batch");
      KLR_HIP(ctx, hipMemcpyAsync(pin_report, hist, report_bytes, hipMemcpyDeviceToHost, st));
      KLR_HIP(ctx, hipStreamSynchronize(st));
      {
        const double* pr = (const double*)pin_report;
        memcpy(hhist.data(), pr, (size_t)KSPEC * d * 8);
        memcpy(hdelta.data(), pr + (size_t)KSPEC * d, KSPEC * 8);
        memcpy(hloss.data(), pr + (size_t)KSPEC * d + KSPEC, KSPEC * 8);
        memcpy(hstop.data(), pr + (size_t)KSPEC * d + 2 * KSPEC + 2, (KSPEC + 1) * sizeof(int));
      }
      const int nd = hstop[KSPEC];
      if (pe0) ctx->prof_cd.push_back({pe0, pe1, nd});
      bool cut = false;     // stopped before the end of the batch -> device theta must be rewound
      for (int k = 0; k < nd; k++) {
        th.assign(hhist.begin() + (size_t)k * d, hhist.begin() + (size_t)(k + 1) * d);
        if (eval_loss) { accepted_sum = hloss[k]; have_sum = true; }
        if (hstop[k]) { inner_done = true; cut = (k + 1 < nd); dbg_stopk = 1; break; }                  // coordinate.go:74-75
        // hook (kmerLr_estimator_hook.go:46-99)
        std::swap(loss_old, loss_new);
        if (eval_loss) loss_new = full_loss(th.data(), hloss[k]);
        if (recs && count < recs_cap) {
          int nz = 0;
          for (int j = 0; j < d; j++) if (th[j] != 0.0) nz++;
          klr_iter_rec& r = recs[count];
          r.iteration = (int32_t)iter; r.nonzero = nz; r.change = hdelta[k]; r.lambda = lambda; r.loss = loss_new;
          r.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        count++;
        if (eps_loss != 0.0 && std::fabs(loss_old - loss_new) < eps_loss) { inner_done = true; cut = (k + 1 < nd); break; }
        iter++;
        if (iter >= max_iter) { inner_done = true; cut = (k + 1 < nd); break; }
      }
      if (cut) { thd_stale = true; q_stale = true; if (!inline_theta) KLR_TRY(sync_thd()); }
      if (nd == 0) inner_done = true;
    }
    if (ctx_opt(ctx, "KLR_CD_DEBUG")) fprintf(stderr, "[cd] d=%d inner sweeps=%lld stopf=%d\n", d, (long long)(iter - dbg_iter0), dbg_stopk);
    if (ctx->cd_pred_on) {
      if ((int)ctx->cd_pred.size() <= relin) ctx->cd_pred.resize((size_t)relin + 1, 0);
      ctx->cd_pred[(size_t)relin] = (int)std::min<i64>(iter - dbg_iter0, 1 << 20);
    }
    // ---- outer checks -----------------------------------------------------------------------------
    double dl;
    if (host_eval_stopping(tp.data(), th.data(), d, eps, &dl)) break;                   // coordinate.go:119-120
    if (eval_loss && eps_loss != 0.0) {
      if (!have_sum) {
        double* acc = lossb;
        KLR_TRY(sync_thd());
        KLR_TRY(launch_row_pass(ctx, red, thd, ROW_LOSS, cw, nullptr, nullptr, acc));
        KLR_TRY(klr_allreduce(ctx, acc, 1, KLR_F64));
        KLR_TRY(d2h_sync(ctx, &accepted_sum, acc, 8));
      }
      const double lc = full_loss(th.data(), accepted_sum);
      if (std::fabs(loss_outer_prev - lc) < eps_loss) break;
      loss_outer_prev = lc;
    }
  }
  KLR_HIP(ctx, hipStreamSynchronize(st));
  memcpy(theta, th.data(), (size_t)d * 8);
  if (n_recs) *n_recs = count;
  return KLR_OK;
}
