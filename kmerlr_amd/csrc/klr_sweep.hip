// klr_sweep.hip -- the two passes of logisticRegression.Gradient / Loss
// (kmerLr_logistic_regression.go:151-272) as HIP kernels for gfx950:
//
//   row pass     r_i = x_i . theta, log sigma, s_i = cw[y]/n (sigma - y), loss terms       (CSR)
//   column sweep g_j = sum_i s_i x_ij : one lane group per column chunk, 16-byte coalesced
//                reads of the CSC row-id and value streams, f64 gather of s, f64 FMA,
//                shuffle reduction inside the group                                        (CSC)
//
// The column sweep is the HBM-roofline kernel (DESIGN.md "kernels"): algorithmic bytes per stored
// nonzero = 4 (row id) + 4|8 (value); s (8n bytes) is gathered from L2.
#include "klr_internal.h"
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstring>
#include <cmath>

// ---------------------------------------------------------------------------------------------
// scalar arithmetic shared by all row passes; mirrors oracle/klr_oracle.c
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double d_log_add0(double x) {
  // LogAdd(0, x) = log(1 + e^x) = max(0,x) + log1p(exp(-|x|))   (autodiff logarithmetic.LogAdd)
  const double a = fabs(x);
  return fmax(x, 0.0) + log1p(exp(-a));
}

template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------
// row pass: G lanes per row
// ---------------------------------------------------------------------------------------------
#define THETA_INLINE_MAX KLR_THETA_INLINE_MAX
#define THETA_LDS_MAX 2048        // a reduced problem's theta up to this length is staged in LDS by every block (16 KiB)
struct RowPassArgs {
  const i64* rowptr; const u32* col; const void* val; const uint8_t* labels; const uint8_t* active;
  const double* theta; i64 n; double inv_n; double cw0, cw1;
  double* out0; double* out1; double* blk;   // blk: per-block loss partials
  double theta0_only; int const_mode;
  double fix_scale; i64* S; unsigned long long* acc;     // ROW_SWEEP_FIX
  int theta_inline_n;                                    // > 0: theta travels in the kernel arguments (no upload)
  int theta_lds_n;                                       // > 0: theta (device) is copied to LDS first: a.theta[col] becomes an LDS read
  u32* zero_u32;                                         // nullable: a word the first block clears (the candidate cursor of the
                                                         // hinted selection that follows the sweep: one launch less at its end)
  double theta_inline[THETA_INLINE_MAX];
};

// y_in / act_in: the row's label and mask byte, fetched by the caller beside the row offsets (one round trip less than
// reading them here, after the dot product); S_out: the fixed-point weight written to a.S (ROW_SWEEP_FIX), for the tail sums
__device__ __forceinline__ void row_finish(const RowPassArgs& a, int mode, i64 i, double r, double& loss_term, int y_in, bool act_in,
                                           i64& S_out) {
  S_out = 0;
  if (!act_in && mode != ROW_LOGPDF) {
    // row outside the current training / test split: it contributes nothing to any sum over samples
    loss_term = 0.0;
    if (mode == ROW_SWEEP) { a.out0[i] = 0.0; if (a.out1) a.out1[i] = 0.0; }
    else if (mode == ROW_SWEEP_FIX) a.S[i] = 0;
    else if (mode == ROW_IRLS) { a.out0[i] = 0.0; a.out1[i] = 0.0; }
    return;
  }
  const int y = y_in;
  const double lp = -d_log_add0(-r);                 // log sigma(r), LogPdf (:147-149)
  if (mode == ROW_LOGPDF) { a.out0[i] = lp; loss_term = 0.0; return; }
  // Loss: r -= cw[y] * ClassLogPdf (:258-264)
  loss_term = y ? a.cw1 * (-lp) : a.cw0 * d_log_add0(r);
  if (mode == ROW_SWEEP || mode == ROW_SWEEP_FIX) {
    const double e = exp(lp);
    // Gradient weights (:174-178), same association as the reference: 1/n * cw * (...)
    const double sv = y ? a.inv_n * a.cw1 * (e - 1.0) : a.inv_n * a.cw0 * e;
    if (mode == ROW_SWEEP) { a.out0[i] = sv; if (a.out1) a.out1[i] = loss_term; }
    else { S_out = __double2ll_rn(sv * a.fix_scale); a.S[i] = S_out; }
  } else if (mode == ROW_IRLS) {
    // estimate_coordinate (:97-111) with the guarded weight of oracle KLO_IRLS_WMIN
    const double p = exp(lp);
    double w = p * (1.0 - p);
    if (w < 1e-12) w = 1e-12;
    a.out1[i] = r + ((y ? 1.0 : 0.0) - p) / w;
    a.out0[i] = w * (y ? a.cw1 : a.cw0);
  }
}

template <typename VT, int G, int MODE>
__global__ __launch_bounds__(256) void k_row_pass(RowPassArgs a) {
  __shared__ double sh[256 / 64];
  __shared__ double sth[THETA_LDS_MAX];
  const i64 gid = ((i64)blockIdx.x * 256 + threadIdx.x) / G;
  const int lane = threadIdx.x % G;
  double loss_term = 0.0;
  if (a.zero_u32 && blockIdx.x == 0 && threadIdx.x == 0) *a.zero_u32 = 0u;
  // everything that depends on the row id alone is requested first -- its offsets, its label, its mask byte -- so that these
  // loads fly while theta is staged: the pass is a chain of dependent round trips (offsets -> entries -> ...) and nothing else
  i64 rb = 0, re = 0, S_row = 0; int y_row = 0; bool act_row = true;
  if (gid < a.n) {
    if (!a.const_mode) { rb = a.rowptr[gid]; re = a.rowptr[gid + 1]; }
    if (lane == 0) { y_row = a.labels[gid] != 0; if (a.active) act_row = a.active[gid] != 0; }
  }
  const bool inl = a.theta_inline_n > 0 || a.theta_lds_n > 0;   // theta of a reduced problem: kernel arguments / device -> LDS
  if (a.theta_inline_n > 0) {
    for (int i = threadIdx.x; i < a.theta_inline_n; i += 256) sth[i] = a.theta_inline[i];
    __syncthreads();
  } else if (a.theta_lds_n > 0) {
    for (int i = threadIdx.x; i < a.theta_lds_n; i += 256) sth[i] = a.theta[i];
    __syncthreads();
  }
  if (gid < a.n) {
    double r;
    if (a.const_mode) {
      r = a.theta0_only;
    } else {
      const i64 b = rb, e = re;
      const VT* val = (const VT*)a.val;
      // LinearPdf (:47-68): r = theta[0] + sum_{j>=1} v_j theta[i_j]; entry 0 is the bias
      double acc = 0.0;
      if (inl) {
        // four entries per lane and trip, their loads in flight together (the adds keep their order: same result)
        for (i64 p = b + 1 + lane; p < e; p += 4 * G) {
          u32 c[4]; double v[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const i64 pp = p + (i64)u * G;
            const bool ok = pp < e;
            c[u] = ok ? a.col[pp] : 0u;
            v[u] = ok ? (double)val[pp] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 4; u++) if (p + (i64)u * G < e) acc += v[u] * sth[c[u]];
        }
      } else {
        for (i64 p = b + 1 + lane; p < e; p += G) acc += (double)val[p] * a.theta[a.col[p]];
      }
      acc = group_sum<G>(acc);
      r = (inl ? sth[0] : a.theta[0]) + acc;
    }
    if (lane == 0) row_finish(a, MODE, gid, r, loss_term, y_row, act_row, S_row);
  }
  if (MODE == ROW_SWEEP_FIX) {
    // exact integer partials: bias sum as hi/lo halves of S, loss as two limbs (see k_fix_s)
    __shared__ i64 shi[4][4];
    __shared__ u32 shbad[4];
    i64 hi = 0, lo = 0, l0 = 0, l1 = 0; u32 bad = 0;
    if (gid < a.n && lane == 0) {
      const i64 v = S_row;                               // (not read back from a.S: the store has only just been issued)
      hi = v >> 32; lo = (i64)(v & 0xFFFFFFFFll);
      const double t = loss_term;
      if (t >= 0.0 && t < 1099511627776.0) {
        const double ts = t * 1048576.0;
        const double f = floor(ts);
        l0 = (i64)f;
        l1 = __double2ll_rn((ts - f) * 4398046511104.0);
      } else bad = 1;
    }
    // only lane 0 of every G-lane group holds a value: the butterfly starts at distance G (4 levels instead of 6 at G = 4)
#pragma unroll
    for (int o = 32; o >= G; o >>= 1) {
      hi += __shfl_xor(hi, o, 64); lo += __shfl_xor(lo, o, 64);
      l0 += __shfl_xor(l0, o, 64); l1 += __shfl_xor(l1, o, 64);
      bad += __shfl_xor(bad, o, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { shi[0][w] = hi; shi[1][w] = lo; shi[2][w] = l0; shi[3][w] = l1; shbad[w] = bad; }
    __syncthreads();
    // per-block integer partials (summed by k_finish_coo; integer sums are order free)
    if (threadIdx.x < 4)
      a.acc[(size_t)blockIdx.x * 5 + threadIdx.x] =
          (unsigned long long)(shi[threadIdx.x][0] + shi[threadIdx.x][1] + shi[threadIdx.x][2] + shi[threadIdx.x][3]);
    if (threadIdx.x == 4) a.acc[(size_t)blockIdx.x * 5 + 4] = (unsigned long long)(shbad[0] + shbad[1] + shbad[2] + shbad[3]);
    return;
  }
  if (MODE != ROW_LOGPDF) {
    // block partial of the loss: fixed order -> deterministic
    double v = loss_term;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) a.blk[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  }
}

// sums nb block partials in a fixed order into *out
__global__ __launch_bounds__(1024) void k_sum_partials(const double* __restrict__ blk, i64 nb, double* __restrict__ out) {
  __shared__ double sh[1024];
  double v = 0.0;
  for (i64 i = threadIdx.x; i < nb; i += 1024) v += blk[i];
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}

template <typename VT, int G>
static void row_pass_mode(int mode, dim3 grid, hipStream_t st, const RowPassArgs& a) {
  switch (mode) {
    case ROW_SWEEP:  hipLaunchKernelGGL((k_row_pass<VT, G, ROW_SWEEP>), grid, dim3(256), 0, st, a); break;
    case ROW_LOSS:   hipLaunchKernelGGL((k_row_pass<VT, G, ROW_LOSS>), grid, dim3(256), 0, st, a); break;
    case ROW_IRLS:   hipLaunchKernelGGL((k_row_pass<VT, G, ROW_IRLS>), grid, dim3(256), 0, st, a); break;
    case ROW_SWEEP_FIX: hipLaunchKernelGGL((k_row_pass<VT, G, ROW_SWEEP_FIX>), grid, dim3(256), 0, st, a); break;
    default:         hipLaunchKernelGGL((k_row_pass<VT, G, ROW_LOGPDF>), grid, dim3(256), 0, st, a); break;
  }
}

template <typename VT>
static void row_pass_group(int g, int mode, i64 n, hipStream_t st, const RowPassArgs& a) {
  const dim3 grid((unsigned)std::max<i64>(1, ceil_div(n * g, 256)));
  switch (g) {
    case 1:  row_pass_mode<VT, 1>(mode, grid, st, a); break;
    case 4:  row_pass_mode<VT, 4>(mode, grid, st, a); break;
    case 16: row_pass_mode<VT, 16>(mode, grid, st, a); break;
    default: row_pass_mode<VT, 64>(mode, grid, st, a); break;
  }
}

// lanes per row by the average row length (KLR_ROW_G overrides: A/B tooling)
static int row_group(klr_ctx* ctx, const klr_mat* mat, bool const_mode) {
  int g = 64;
  if (const_mode || mat->avg_row <= 4.0) g = 1;
  else if (mat->avg_row <= 24.0) g = 4;
  else if (mat->avg_row <= 160.0) g = 16;
  if (!const_mode) if (const char* e = ctx_opt(ctx, "KLR_ROW_G")) { const int v = atoi(e); if (v == 1 || v == 4 || v == 16 || v == 64) g = v; }
  return g;
}

static int row_pass_common(klr_ctx* ctx, const klr_mat* mat, const double* theta_dev, double theta0, bool const_mode,
                           int mode, const double cw[2], double* out0, double* out1, double* loss_acc_dev,
                           double fix_scale = 0.0, i64* S = nullptr, unsigned long long* acc = nullptr,
                           const double* theta_host = nullptr, i64 theta_len = 0, u32* zero_u32 = nullptr, i64* nblocks_out = nullptr) {
  const i64 n = mat->n;
  const int g = row_group(ctx, mat, const_mode);
  const i64 nb = std::max<i64>(1, ceil_div(n * g, 256));
  double* blk;
  KLR_TRY(ws_get(ctx, WS_BLK, (size_t)nb * 5 * sizeof(double), (void**)&blk));
  RowPassArgs a;
  a.rowptr = mat->rowptr; a.col = mat->col; a.val = mat->val; a.labels = mat->labels; a.active = mat->active;
  a.theta = theta_dev; a.n = n; a.inv_n = 1.0 / (double)mat->n_total; a.cw0 = cw[0]; a.cw1 = cw[1];
  a.out0 = out0; a.out1 = out1; a.blk = blk; a.theta0_only = theta0; a.const_mode = const_mode ? 1 : 0;
  a.fix_scale = fix_scale; a.S = S; a.acc = acc;
  a.theta_inline_n = 0; a.theta_lds_n = 0; a.zero_u32 = zero_u32;
  if (theta_host && !const_mode && theta_len > 0 && theta_len <= THETA_INLINE_MAX) {
    a.theta_inline_n = (int)theta_len;
    memcpy(a.theta_inline, theta_host, (size_t)theta_len * sizeof(double));
  } else if (!const_mode && theta_dev && mat->dim <= THETA_LDS_MAX && (double)mat->dim <= 2.0 * (256 / g) * std::max(1.0, mat->avg_row)) {
    a.theta_lds_n = (int)mat->dim;                       // worth it while a block's rows use more entries than the copy moves
  }
  if (nblocks_out) *nblocks_out = n > 0 ? nb : 0;      // blocks that write partials
  if (n > 0) {
    if (mat->val_f64) row_pass_group<double>(g, mode, n, ctx->stream, a);
    else              row_pass_group<float>(g, mode, n, ctx->stream, a);
  }
  if (loss_acc_dev) {
    if (n > 0 && mode != ROW_LOGPDF && mode != ROW_SWEEP_FIX)
      hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, ctx->stream, blk, nb, loss_acc_dev);
    else
      KLR_HIP(ctx, hipMemsetAsync(loss_acc_dev, 0, sizeof(double), ctx->stream));
  }
  KLR_HIP(ctx, hipGetLastError());
  return KLR_OK;
}

int launch_row_pass(klr_ctx* ctx, const klr_mat* mat, const double* theta_dev, int mode, const double cw[2],
                    double* out0, double* out1, double* loss_acc_dev, const double* theta_host, i64 theta_len) {
  return row_pass_common(ctx, mat, theta_dev, 0.0, false, mode, cw, out0, out1, loss_acc_dev, 0.0, nullptr, nullptr,
                         theta_host, theta_len);
}

int launch_row_pass_fix(klr_ctx* ctx, const klr_mat* mat, const double* theta_dev, double theta0, bool const_mode,
                        const double cw[2], double scale, i64* S, unsigned long long** rowblk, i64* nrowblk,
                        const double* theta_host, i64 theta_len, u32* zero_u32) {
  // the kernel writes 5 integer partials per block into the WS_BLK block-partial buffer
  const int g = row_group(ctx, mat, const_mode);
  const i64 nb = std::max<i64>(1, ceil_div(mat->n * g, 256));
  unsigned long long* blk;
  KLR_TRY(ws_get(ctx, WS_BLK, (size_t)nb * 5 * sizeof(unsigned long long), (void**)&blk));
  *rowblk = blk;
  return row_pass_common(ctx, mat, theta_dev, theta0, const_mode, ROW_SWEEP_FIX, cw, nullptr, nullptr, nullptr, scale, S, blk,
                         theta_host, theta_len, zero_u32, nrowblk);
}

int launch_row_const(klr_ctx* ctx, const klr_mat* mat, double theta0, int mode, const double cw[2],
                     double* out0, double* out1, double* loss_acc_dev) {
  return row_pass_common(ctx, mat, nullptr, theta0, true, mode, cw, out0, out1, loss_acc_dev);
}

// ---------------------------------------------------------------------------------------------
// column sweep  g_raw[j] = sum_i s[row] * x    (the roofline kernel)
// ---------------------------------------------------------------------------------------------
template <typename VT> struct V4;
template <> struct V4<float>  { typedef float4 type; };
template <> struct V4<double> { typedef double4 type; };

template <typename VT, int G, bool SQ>
__global__ __launch_bounds__(256) void k_xt_s(const u32* __restrict__ crow, const VT* __restrict__ cval,
                                              const i64* __restrict__ chunk_begin, const u32* __restrict__ chunk_len,
                                              i64 nchunks, const double* __restrict__ s, double* __restrict__ part) {
  const i64 gid = ((i64)blockIdx.x * 256 + threadIdx.x) / G;
  const int lane = threadIdx.x % G;
  if (gid >= nchunks) return;            // whole groups leave together
  const i64 b = chunk_begin[gid];
  const i64 e = b + (i64)chunk_len[gid];
  typedef typename V4<VT>::type vt4;
  double acc = 0.0;
#pragma unroll 2
  for (i64 p = b + 4 * lane; p < e; p += 4 * G) {
    const uint4 r = *reinterpret_cast<const uint4*>(crow + p);
    const vt4 v = *reinterpret_cast<const vt4*>(cval + p);
    const double s0 = s[r.x], s1 = s[r.y], s2 = s[r.z], s3 = s[r.w];
    double x0 = (double)v.x, x1 = (double)v.y, x2 = (double)v.z, x3 = (double)v.w;
    if (SQ) { x0 *= x0; x1 *= x1; x2 *= x2; x3 *= x3; }
    acc = fma(s0, x0, acc);
    acc = fma(s1, x1, acc);
    acc = fma(s2, x2, acc);
    acc = fma(s3, x3, acc);
  }
  acc = group_sum<G>(acc);
  if (lane == 0) part[gid] = acc;
}

// The three column sums of an IRLS re-linearisation in ONE pass over the reduced CSC (f64 values):
//   part[0][c] = sum w x   part[1][c] = sum w x^2   part[2][c] = sum (w z) x      per chunk c,
// same lane striding, same fma order and same group sum as three k_xt_s launches over w, w (squared) and w*z.
template <int G>
__global__ __launch_bounds__(256) void k_irls_cols(const u32* __restrict__ crow, const double* __restrict__ cval,
                                                   const i64* __restrict__ chunk_begin, const u32* __restrict__ chunk_len,
                                                   i64 nchunks, const double* __restrict__ w, const double* __restrict__ z,
                                                   double* __restrict__ part) {
  const i64 gid = ((i64)blockIdx.x * 256 + threadIdx.x) / G;
  const int lane = threadIdx.x % G;
  if (gid >= nchunks) return;            // whole groups leave together
  const i64 b = chunk_begin[gid];
  const i64 e = b + (i64)chunk_len[gid];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll 2
  for (i64 p = b + 4 * lane; p < e; p += 4 * G) {
    const uint4 r = *reinterpret_cast<const uint4*>(crow + p);
    const double4 v = *reinterpret_cast<const double4*>(cval + p);
    const double w0 = w[r.x], w1 = w[r.y], w2 = w[r.z], w3 = w[r.w];
    const double z0 = z[r.x], z1 = z[r.y], z2 = z[r.z], z3 = z[r.w];
    a0 = fma(w0, v.x, a0); a0 = fma(w1, v.y, a0); a0 = fma(w2, v.z, a0); a0 = fma(w3, v.w, a0);
    a1 = fma(w0, v.x * v.x, a1); a1 = fma(w1, v.y * v.y, a1); a1 = fma(w2, v.z * v.z, a1); a1 = fma(w3, v.w * v.w, a1);
    a2 = fma(w0 * z0, v.x, a2); a2 = fma(w1 * z1, v.y, a2); a2 = fma(w2 * z2, v.z, a2); a2 = fma(w3 * z3, v.w, a2);
  }
  a0 = group_sum<G>(a0); a1 = group_sum<G>(a1); a2 = group_sum<G>(a2);
  if (lane == 0) { part[gid] = a0; part[nchunks + gid] = a1; part[2 * nchunks + gid] = a2; }
}

// part = 3 * nchunks doubles (WS_PART), consumed by the solver's k_irls_finish
int launch_irls_cols(klr_ctx* ctx, const klr_mat* mat, const double* w, const double* z, double** part_out) {
  double* part;
  KLR_TRY(ws_get(ctx, WS_PART, (size_t)std::max<i64>(mat->nchunks, 1) * 3 * sizeof(double), (void**)&part));
  *part_out = part;
  if (mat->nchunks > 0) {
    const int g = mat->group;
    const dim3 grid((unsigned)ceil_div(mat->nchunks * g, 256));
    const double* cv = (const double*)mat->cval;
    hipStream_t st = ctx->stream;
    switch (g) {
      case 4:  hipLaunchKernelGGL(k_irls_cols<4>, grid, dim3(256), 0, st, mat->crow, cv, mat->chunk_begin, mat->chunk_len, mat->nchunks, w, z, part); break;
      case 8:  hipLaunchKernelGGL(k_irls_cols<8>, grid, dim3(256), 0, st, mat->crow, cv, mat->chunk_begin, mat->chunk_len, mat->nchunks, w, z, part); break;
      case 16: hipLaunchKernelGGL(k_irls_cols<16>, grid, dim3(256), 0, st, mat->crow, cv, mat->chunk_begin, mat->chunk_len, mat->nchunks, w, z, part); break;
      case 32: hipLaunchKernelGGL(k_irls_cols<32>, grid, dim3(256), 0, st, mat->crow, cv, mat->chunk_begin, mat->chunk_len, mat->nchunks, w, z, part); break;
      default: hipLaunchKernelGGL(k_irls_cols<64>, grid, dim3(256), 0, st, mat->crow, cv, mat->chunk_begin, mat->chunk_len, mat->nchunks, w, z, part); break;
    }
    KLR_HIP(ctx, hipGetLastError());
  }
  return KLR_OK;
}

// adds a column's chunk sums in chunk order
__global__ __launch_bounds__(256) void k_finish_cols(const i64* __restrict__ colchunk, i64 dim,
                                                     const double* __restrict__ part, double* __restrict__ g) {
  const i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
  if (j >= dim) return;
  double v = 0.0;
  for (i64 c = colchunk[j]; c < colchunk[j + 1]; c++) v += part[c];
  g[j] = v;
}

template <typename VT, bool SQ>
static void xt_s_group(int g, dim3 grid, hipStream_t st, const klr_mat* m, const double* s, double* part) {
  const VT* cv = (const VT*)m->cval;
  switch (g) {
    case 4:  hipLaunchKernelGGL((k_xt_s<VT, 4, SQ>), grid, dim3(256), 0, st, m->crow, cv, m->chunk_begin, m->chunk_len, m->nchunks, s, part); break;
    case 8:  hipLaunchKernelGGL((k_xt_s<VT, 8, SQ>), grid, dim3(256), 0, st, m->crow, cv, m->chunk_begin, m->chunk_len, m->nchunks, s, part); break;
    case 16: hipLaunchKernelGGL((k_xt_s<VT, 16, SQ>), grid, dim3(256), 0, st, m->crow, cv, m->chunk_begin, m->chunk_len, m->nchunks, s, part); break;
    case 32: hipLaunchKernelGGL((k_xt_s<VT, 32, SQ>), grid, dim3(256), 0, st, m->crow, cv, m->chunk_begin, m->chunk_len, m->nchunks, s, part); break;
    default: hipLaunchKernelGGL((k_xt_s<VT, 64, SQ>), grid, dim3(256), 0, st, m->crow, cv, m->chunk_begin, m->chunk_len, m->nchunks, s, part); break;
  }
}

// ---------------------------------------------------------------------------------------------
// tiled column sweep for k-mer count data (k_sweep_coo: the u32 / u24 / r64 entry formats; the row-streamed q64
// formats have their own kernel, k_sweep_q64 below)
//
// Random 8-byte gathers of s from L2 move a whole cache line per stored nonzero and cap the plain CSC
// kernel at ~20 % of HBM peak (profiles/README.md).  Here both vectors the sweep touches at random live in LDS:
// the matrix is cut into tiles of at most 4094 x 4032 (tile_geometry), a workgroup holds the tile's s values and
// 4096 column accumulators (32 KiB each), and the ONLY HBM stream is the packed entry array -- 2, 3 or 4 bytes per
// stored unit entry (formats below), read with 16-byte (12-byte) loads.  Occupancy hides the latency: 1024-thread
// workgroups, two per CU (32 waves/CU), one load per lane and trip plus a ping-pong prefetch (tools/ab_sweep.py).
//
// Accumulation is exact: s is converted to int64 fixed point (scale 2^E chosen from max|s| and the
// largest column sum so that no column sum can overflow), products are integer, and the column
// accumulators are LDS integer atomics.  Integer addition is associative, so the result does not
// depend on lane scheduling, on how the stream is cut into pieces, or on the number of GPUs (the
// cross-GPU all-reduce is int64 too): the gradient is bitwise reproducible and partition invariant.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void coo_consume(const u32 w, const i64* __restrict__ sS, unsigned long long* __restrict__ sG) {
  const u32 row = w & (TILE_DIM - 1), col = (w >> TILE_BITS) & (TILE_DIM - 1), v = w >> 24;
  const unsigned long long prod = (unsigned long long)sS[row] * (unsigned long long)v;   // mod 2^64 == two's complement
  atomicAdd(&sG[col], prod);
}
// timing-only ablations (tools/ab_sweep.py): results are wrong by construction
template <int ABL>
__device__ __forceinline__ void coo_consume_abl(const u32 w, const i64* __restrict__ sS, unsigned long long* __restrict__ sG,
                                                unsigned long long& sink) {
  const u32 row = w & (TILE_DIM - 1), col = (w >> TILE_BITS) & (TILE_DIM - 1), v = w >> 24;
  if (ABL == 4) { atomicAdd(&sG[col], (unsigned long long)sS[row]); }                             // no multiply
  else if (ABL == 3) { sink += w; }                                                                    // stream only
  else if (ABL == 1) { sink += (unsigned long long)sS[row] * (unsigned long long)v + col; }       // no atomics
  else { atomicAdd(&sG[col], (unsigned long long)(v + row)); }                                    // no s gather
}

template <bool NT>
__device__ __forceinline__ uint4 coo_load(const uint4* __restrict__ ent4, u32 q) {
  if (NT) {
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 tt = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ent4) + q);
    return make_uint4(tt.x, tt.y, tt.z, tt.w);
  }
  return ent4[q];
}

// Entry formats of the stream (chosen at ingest, whichever is fewer bytes):
//   u32: (count:8 << 24) | col:12 << 12 | row:12, a quad = 16 bytes, zero-padded;
//   u24: col:12 << 12 | row:12 in 3 bytes, little endian, a count c stored as c entries, a quad = 12 bytes read as
//        one dwordx3 per lane; tiles are padded with the null entry row 4095 (s slot 4095 is kept zero:
//        tile_geometry never gives a tile more than 4094 rows), column 0.  k-mer count rows sum to the number of
//        windows, so unit entries cost at most windows/distinct more entries (0.9 % at configs[2]) for 25 % fewer
//        bytes each, and the product S*count disappears.
//   r64: unit entries with every (row, tile) run padded to 4 columns: a quad is 8 bytes, row:12 << 48 | 4 x col:12, i.e.
//        2 bytes per unit entry and ONE s read per four entries; a lane reads two quads (16 bytes) per trip.
//        Padding columns point at the spare accumulator slots 4032..4095 (tiles are at most 4032 wide; the finishing
//        kernel never reads them), spread by the quad's position; null quads carry row 4095.
template <int THREADS, int UNROLL, bool NT, int ABL, int FMT>
__global__ __launch_bounds__(THREADS) void k_sweep_coo(const u32* __restrict__ ent, const uint4* __restrict__ piece,
                                                       const uint4* __restrict__ wgrec, const u32* __restrict__ tileq,
                                                       u32 nrt, u32 nct_or0, u32 RT, u32 ntile, i64 n, const i64* __restrict__ S,
                                                       i64* __restrict__ part) {
  static_assert(FMT == FMT_U32 || UNROLL <= 2, "unit entries: one load per lane and trip (UNROLL 2 = ping-pong prefetch)");
  constexpr bool U24 = FMT == FMT_U24;
  extern __shared__ unsigned long long coo_lds[];
  i64* sS = (i64*)coo_lds;
  unsigned long long* sG = coo_lds + TILE_DIM;
  const u32 tid = threadIdx.x;
  // one 32-byte record per workgroup carries everything its first tile needs -- {first piece, end piece, first
  // tile, its row tile} {-, quad begin, quad end of the piece, quad end inside the first tile} -- so the s tile and
  // the first stream loads are one memory round trip away (piece -> tile offsets -> s tile -> stream used to be
  // four, ~5 us of a 60 us kernel)
  const uint4 w0 = wgrec[2 * blockIdx.x], w1 = wgrec[2 * blockIdx.x + 1];
  const u32 p0 = w0.x, p1 = w0.y;
  unsigned long long sink = 0ull;
  u32 cur_rt = 0xFFFFFFFFu;
  if (FMT != FMT_U32 && tid == 0) sS[TILE_DIM - 1] = 0;   // the null entry's s slot; never overwritten (RT <= 4094)
  for (u32 p = p0; p < p1; p++) {
    uint4 d;                                            // {column tile, quad begin, quad end, first tile}
    if (p == p0) d = make_uint4(0u, w1.y, w1.z, w0.z); else d = piece[p];
    const u32 pq0 = d.y, pq1 = d.z;
    __syncthreads();                                    // previous piece written out
    for (u32 i = tid; i < TILE_DIM; i += THREADS) sG[i] = 0ull;
    // walk down the piece: one s tile per row tile.  A workgroup with many short tiles (a 1M-column shard: ~16 tiles of
    // ~16k entries each) used to pay three dependent round trips per tile -- tile offsets, s tile, first stream load;
    // the offsets of the NEXT tile are now requested a tile ahead (tq_a / tq_b) and, in the loop without a prefetch
    // ring, the first stream load is issued before the s tile is fetched.
    u32 tq_a = tileq[min(d.w + 1u, ntile)], tq_b = tileq[min(d.w + 2u, ntile)];        // offsets of tile d.w + 1
    for (u32 t = d.w; ; t++) {
      u32 q0, q1, rt;
      if (p == p0 && t == d.w) { q0 = pq0; q1 = w1.w; rt = w0.w; }
      else {
        u32 tq0, tq1;
        if (t == d.w) { tq0 = tileq[t]; tq1 = tq_a; }
        else { tq0 = tq_a; tq1 = tq_b; tq_a = tq_b; tq_b = tileq[min(t + 2u, ntile)]; }
        if (tq0 >= pq1) break;
        q0 = max(pq0, tq0); q1 = min(pq1, tq1);
        if (q0 >= q1) continue;                         // empty tile
        rt = nct_or0 ? t / nct_or0 : t % nrt;           // row-tile major / column-tile major
      }
      // r64 without the prefetch ring: the first quad pair of the tile is requested now, beside the s tile
      uint4 Apre = make_uint4(0u, 0u, 0u, 0u);
      if (FMT == FMT_R64 && UNROLL == 1) Apre = coo_load<NT>(reinterpret_cast<const uint4*>(ent), min(q0 + tid, q1 - 1));
      if (rt != cur_rt) {
        const i64 r0 = (i64)rt * RT;                    // RT is even: 16-byte aligned
        __syncthreads();                                // the previous s tile is no longer read
        if (r0 + RT <= n) {                             // full tile: 16-byte loads
          const longlong2* src = reinterpret_cast<const longlong2*>(S + r0);
          for (u32 i = tid; i < RT / 2; i += THREADS) reinterpret_cast<longlong2*>(sS)[i] = src[i];
        } else {
          for (u32 i = tid; i < RT; i += THREADS) sS[i] = (r0 + i < n) ? S[r0 + i] : 0;
        }
        cur_rt = rt;
      }
      __syncthreads();                                  // s tile (and the zeroed accumulators) visible
      if (FMT == FMT_R64) {
        // two quads per lane and trip, ping-pong prefetch as in the u24 branch below
        const uint4* ent4 = reinterpret_cast<const uint4*>(ent);
        const u32 nit = (q1 - q0 + THREADS - 1) / THREADS, qend = q1 - 1;
        const u32 dcol = R64_DUMP + (tid & 63u);
        const u32 nlo = dcol | (dcol << 12) | (dcol << 24), nhi = (dcol >> 8) | (dcol << 4) | ((TILE_DIM - 1) << 16);
        const uint4 nulq = make_uint4(nlo, nhi, nlo, nhi);            // row 4095 (s = 0), spare columns
        u32 q = q0 + tid;
        uint4 A, B;
#define EATR(E0, OK) {                                                                                              \
          const uint4 e_ = (OK) ? E0 : nulq;                                                                         \
          if (ABL == 3) { sink += e_.x + e_.y + e_.z + e_.w; }                                                       \
          else {                                                                                                     \
            const unsigned long long sa_ = (unsigned long long)sS[(e_.y >> 16) & 0xFFFu], sb_ = (unsigned long long)sS[(e_.w >> 16) & 0xFFFu]; \
            atomicAdd(&sG[e_.x & 0xFFFu], sa_); atomicAdd(&sG[(e_.x >> 12) & 0xFFFu], sa_);                          \
            atomicAdd(&sG[__builtin_amdgcn_alignbit(e_.y, e_.x, 24) & 0xFFFu], sa_); atomicAdd(&sG[(e_.y >> 4) & 0xFFFu], sa_); \
            atomicAdd(&sG[e_.z & 0xFFFu], sb_); atomicAdd(&sG[(e_.z >> 12) & 0xFFFu], sb_);                          \
            atomicAdd(&sG[__builtin_amdgcn_alignbit(e_.w, e_.z, 24) & 0xFFFu], sb_); atomicAdd(&sG[(e_.w >> 4) & 0xFFFu], sb_); } }
        if (UNROLL == 2) {
          A = coo_load<NT>(ent4, min(q, qend));
          for (u32 it = 0; it < nit; it += 2) {
            B = coo_load<NT>(ent4, min(q + THREADS, qend));
            __builtin_amdgcn_sched_barrier(0);
            EATR(A, q < q1);
            __builtin_amdgcn_sched_barrier(0);
            A = coo_load<NT>(ent4, min(q + 2 * THREADS, qend));
            __builtin_amdgcn_sched_barrier(0);
            EATR(B, q + THREADS < q1);
            __builtin_amdgcn_sched_barrier(0);
            q += 2 * THREADS;
          }
        } else {
          EATR(Apre, q < q1);
          for (q += THREADS; q < q1; q += THREADS) { A = coo_load<NT>(ent4, q); EATR(A, true); }
        }
#undef EATR
      } else if (U24 && UNROLL == 2 && ABL == 0) {
        // ping-pong prefetch: the next 12 bytes are requested before the current quad goes through LDS.  Two register
        // sets alternate (no copies of in-flight values), the loads are branch free (clamped to the tile's last
        // quad) and the trip count is uniform (lanes past the end consume null entries spread over the columns), so
        // the compiler counts its waits (vmcnt(1)) instead of draining before every use.  The kernel is a closed
        // loop between HBM latency and the LDS atomic unit (~14 clocks per ds_add_u64 instruction whatever its lanes
        // do, tools/lds_probe.hip); overlapping the two is worth 3-4 %.
        const u32 nit = (q1 - q0 + THREADS - 1) / THREADS, qend = q1 - 1;
        const u32 nul = ((tid & (TILE_DIM - 1)) << TILE_BITS) | (TILE_DIM - 1);     // row 4095 (s = 0), a column of its own
        const u32 n0 = nul | (nul << 24), n1 = (nul >> 8) | (nul << 16), n2 = (nul >> 16) | (nul << 8);
        u32 q = q0 + tid;
        u32 A0, A1, A2, B0, B1, B2;
#define LD3(X0, X1, X2, QQ) { const u32* s_ = ent + 3ull * min((QQ), qend); X0 = __builtin_nontemporal_load(s_); X1 = __builtin_nontemporal_load(s_ + 1); X2 = __builtin_nontemporal_load(s_ + 2); }
#define EAT3(X0, X1, X2, OK) {                                                                                      \
          const bool ok_ = (OK);                                                                                     \
          const u32 x0_ = ok_ ? X0 : n0, x1_ = ok_ ? X1 : n1, x2_ = ok_ ? X2 : n2;                                   \
          const u32 e0_ = x0_ & 0xFFFFFFu, e1_ = __builtin_amdgcn_alignbit(x1_, x0_, 24) & 0xFFFFFFu;                \
          const u32 e2_ = __builtin_amdgcn_alignbit(x2_, x1_, 16) & 0xFFFFFFu, e3_ = x2_ >> 8;                       \
          const unsigned long long s0_ = (unsigned long long)sS[e0_ & (TILE_DIM - 1)], s1_ = (unsigned long long)sS[e1_ & (TILE_DIM - 1)]; \
          const unsigned long long s2_ = (unsigned long long)sS[e2_ & (TILE_DIM - 1)], s3_ = (unsigned long long)sS[e3_ & (TILE_DIM - 1)]; \
          atomicAdd(&sG[e0_ >> TILE_BITS], s0_); atomicAdd(&sG[e1_ >> TILE_BITS], s1_);                              \
          atomicAdd(&sG[e2_ >> TILE_BITS], s2_); atomicAdd(&sG[e3_ >> TILE_BITS], s3_); }
        LD3(A0, A1, A2, q);
        for (u32 it = 0; it < nit; it += 2) {
          LD3(B0, B1, B2, q + THREADS);
          __builtin_amdgcn_sched_barrier(0);              // keep the order: request, then consume the other set
          EAT3(A0, A1, A2, q < q1);
          __builtin_amdgcn_sched_barrier(0);
          LD3(A0, A1, A2, q + 2 * THREADS);
          __builtin_amdgcn_sched_barrier(0);
          EAT3(B0, B1, B2, q + THREADS < q1);
          __builtin_amdgcn_sched_barrier(0);
          q += 2 * THREADS;
        }
#undef LD3
#undef EAT3
      } else if (U24) {
        for (u32 q = q0 + tid; q < q1; q += THREADS) {
          const u32* src = ent + 3ull * q;
          u32 a0, a1, a2;
          if (NT) { a0 = __builtin_nontemporal_load(src); a1 = __builtin_nontemporal_load(src + 1); a2 = __builtin_nontemporal_load(src + 2); }
          else { a0 = src[0]; a1 = src[1]; a2 = src[2]; }
          if (ABL == 3) { sink += a0 + a1 + a2; continue; }
          const u32 e0 = a0 & 0xFFFFFFu;
          const u32 e1 = __builtin_amdgcn_alignbit(a1, a0, 24) & 0xFFFFFFu;
          const u32 e2 = __builtin_amdgcn_alignbit(a2, a1, 16) & 0xFFFFFFu;
          const u32 e3 = a2 >> 8;
          const unsigned long long s0 = (unsigned long long)sS[e0 & (TILE_DIM - 1)], s1 = (unsigned long long)sS[e1 & (TILE_DIM - 1)];
          const unsigned long long s2 = (unsigned long long)sS[e2 & (TILE_DIM - 1)], s3 = (unsigned long long)sS[e3 & (TILE_DIM - 1)];
          atomicAdd(&sG[e0 >> TILE_BITS], s0);
          atomicAdd(&sG[e1 >> TILE_BITS], s1);
          atomicAdd(&sG[e2 >> TILE_BITS], s2);
          atomicAdd(&sG[e3 >> TILE_BITS], s3);
        }
      } else {
        const uint4* ent4 = reinterpret_cast<const uint4*>(ent);
        for (u32 q = q0 + tid; q < q1; q += THREADS * UNROLL) {
          uint4 e[UNROLL];
#pragma unroll
          for (int u = 0; u < UNROLL; u++) {
            const u32 qq = q + u * THREADS;
            if (qq < q1) e[u] = coo_load<NT>(ent4, qq); else e[u] = make_uint4(0u, 0u, 0u, 0u);
          }
#pragma unroll
          for (int u = 0; u < UNROLL; u++) {
            if (ABL == 0) {
              coo_consume(e[u].x, sS, sG);
              coo_consume(e[u].y, sS, sG);
              coo_consume(e[u].z, sS, sG);
              coo_consume(e[u].w, sS, sG);
            } else {
              coo_consume_abl<ABL>(e[u].x, sS, sG, sink);
              coo_consume_abl<ABL>(e[u].y, sS, sG, sink);
              coo_consume_abl<ABL>(e[u].z, sS, sG, sink);
              coo_consume_abl<ABL>(e[u].w, sS, sG, sink);
            }
          }
        }
      }
    }
    if (ABL == 1 || ABL == 3) sG[tid] += sink;
    __syncthreads();
    i64* out = part + (i64)p * TILE_DIM;
    for (u32 i = tid; i < TILE_DIM; i += THREADS) out[i] = (i64)sG[i];    // plain stores: the next kernel reads them back
  }
}

// ---------------------------------------------------------------------------------------------
// q64: the row-streamed packed stream (FMT_Q64C12 / FMT_Q64C13)
//
// The matrix is cut into column STRIPS only (2^CB - 64 columns wide, CB = 12 or 13); inside a strip the unit
// entries of ALL rows follow each other in ascending row order, every (row, strip) run padded to 4 columns: an
// 8-byte quad = row:(64 - 4 CB) | 4 x col:CB, two quads per 16-byte lane unit.  There are no row tiles: a quad's row
// field is relative to the first row of its 64-unit block (blkrow, 4 bytes per KiB of stream), and the fixed-point
// weight S[row] is NOT staged in LDS -- consecutive lanes hold consecutive rows, so the S reads are short
// sequential global loads that hit L1/L2 (a few cache lines per wavefront and trip).  What that buys:
//   * LDS holds only column accumulators (R regions of 2^CB: 32 or 64 KiB each): no s-tile load, no barrier, no drain
//     anywhere in mid-stream -- a 1M-column shard (7,967 tiles of ~15.6k entries in the tiled layout, each paying
//     barrier + s-tile reload) streams without a stop; the 16 wavefronts of a workgroup run on their own, and with
//     R = 2 even across the first strip boundary of the workgroup's share;
//   * strips twice as wide at CB = 13: runs of a 1M-column shard are ~7.7 entries instead of ~3.9, padding
//     19 % instead of 40 %;
//   * the ingest orders the columns inside every lane's unit so that no LDS bank pair is wanted by three or more lanes
//     of a 16-lane group in the same slot (k_q64_bank_order, klr_ingest.hip): the atomics then run at the rate of
//     their address / data transfer and the HBM stream sets the pace.
// Accumulation as in k_sweep_coo: exact int64 LDS atomics, one 2^CB-wide partial vector per (strip, workgroup) piece.
// ---------------------------------------------------------------------------------------------
// ABL (timing-only ablations, builds with -DKLR_ABLATIONS; results wrong by construction): 1 no write-out,
// 2 no atomics, 3 no S loads, 4 stream only
typedef int q64_i32x4 __attribute__((ext_vector_type(4)));
#ifdef KLR_ABLATIONS
// development builds: 100 MHz wall-clock stamps of workgroup phases (tools/q64_phases.py), wave 0 and wave 15
__device__ unsigned long long q64_ts[2 * 1024 * 8];
#define Q_TS(I) { if (lane == 0u && (wave == 0u || wave == 15u) && blockIdx.x < 1024u) q64_ts[((wave ? 1024u : 0u) + blockIdx.x) * 8u + (I)] = wall_clock64(); }
extern "C" int klr_debug_q64_ts(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(q64_ts), sizeof(q64_ts)); }
#else
#define Q_TS(I)
#endif
__device__ q64_i32x4 q64_raw_buffer_load_x4(q64_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
constexpr int q64_gcd(int a, int b) { return b == 0 ? a : q64_gcd(b, a % b); }

// A = stream units requested ahead of the one being consumed, B = S pairs requested ahead (B < A).  Loads return in
// the order they were issued, so the wait for the S pair of step k also waits for every stream unit requested before
// it: with the S pair of step k + B going out BEFORE the stream unit of step k + A, min(B + 1, A - B) steps of stream
// latency are covered.  (2, 1) with one workgroup per CU, (1, 0) with two.  A wavefront has only 50 - 70 steps in the
// whole kernel (16 wavefronts x 256 workgroups share ~200k - 290k blocks), so the pipeline is never in a steady state:
// deeper rings ((7, 3) was the choice while the LDS atomics set the pace) now only lengthen the start and the drain --
// measured 41.7 / 43.0 / 44.0 / 44.9 us at (2,1) / (3,1) / (5,2) / (7,3) on configs[2] (tools/ab_q64_depth.py).
template <int CB, bool NT, int A, int B, int R, int ABL = 0, bool SLDS = false>
__global__ __launch_bounds__(1024) void k_sweep_q64(const uint4* __restrict__ ent, u32 ent_bytes, const u32* __restrict__ blkrow,
                                                    const uint4* __restrict__ piece, const uint4* __restrict__ wgrec,
                                                    const i64* __restrict__ S, i64* __restrict__ part) {
  constexpr u32 PD = 1u << CB, CM = PD - 1u;
  constexpr int RSH = 4 * CB - 32;                     // row field: bits [4 CB, 64) of the quad = y >> RSH
  constexpr int NU = A + 1, NS = B + 1, L = NU * NS / q64_gcd(NU, NS);
  static_assert(B < A && L <= 24, "ring sizes");
  static_assert(R == 1 || R == 2, "accumulator regions");
  extern __shared__ unsigned long long q64_lds[];
  unsigned long long* sG = q64_lds;                    // the accumulators of the piece being streamed: region (p - p0) % R
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // one 32-byte record per workgroup: {first piece, end piece, first unit and end unit of the first piece} {first row of
  // its first block} -- the first stream loads are one memory round trip away, not three (record -> piece -> stream), and
  // the accumulators are zeroed while the record is on its way
  Q_TS(0);
  const uint4 wr = wgrec[3 * blockIdx.x];
  const u32 rb_first = wgrec[3 * blockIdx.x + 1].x;
  for (u32 i = tid; i < (u32)R * PD; i += 1024) q64_lds[i] = 0ull;
  const u32 p0 = wr.x, p1 = wr.y;
  // the s values of the workgroup's first rows are requested now, beside the first stream units: the first S loads of
  // the pipeline (which need a stream unit to have landed) then find them in the cache instead of going to HBM in turn
  unsigned long long s_warm = 0ull;
  if (p0 < p1 && ABL < 3 && !SLDS) s_warm = (unsigned long long)S[rb_first + tid];
  unsigned long long sink = 0ull;
  // the stream as a buffer resource: a unit index past the wavefront's last block becomes an out-of-range offset, which
  // loads zeros without touching memory -- no branch around a load (the wait counts stay exact), no duplicate traffic
  q64_i32x4 rsrc;
  {
    const unsigned long long pa = (unsigned long long)ent;
    rsrc.x = __builtin_amdgcn_readfirstlane((int)(u32)pa);
    rsrc.y = __builtin_amdgcn_readfirstlane((int)(u32)(pa >> 32));
    rsrc.z = __builtin_amdgcn_readfirstlane((int)ent_bytes);
    rsrc.w = 0x00020000;
  }
  // The workgroup's stream is a sequence of SEGMENTS: a piece (one strip's accumulators) in rounds of 1024 blocks, dealt
  // round-robin to the 16 wavefronts (64 per wavefront and round: a wavefront fetches the first rows of its <= 64 blocks
  // with ONE vector load, lane l its l-th block, and picks them out with v_readlane as it goes -- a scalar load per block
  // would share its wait counter with the LDS atomics).  The stream units go to registers, so the first A of the NEXT
  // segment (and its row bases) are requested as soon as the current segment's loop ends -- before the barrier, the
  // write-out and the zeroing at a piece boundary, which then cost the drain only: with the descriptor, the row bases,
  // the stream and S each a dependent round trip after the barrier, a boundary cost ~9 us (a 1M-column shard has one in
  // every fifth workgroup; tools/q64_phases.py).
  if (p0 < p1) {
    const u32 blk_last = (ent_bytes >> 10) - 1u;
    uint4 U[NU];
    unsigned long long Sa[NS], Sb[NS];
    u32 p = p0, b0 = 0, nb = 0, last = 0, rbv = 0, flushed = 0;
    uint4 d = make_uint4(0u, wr.z, wr.w, 0u);            // {strip, first unit, end unit, -}: both multiples of 64
    uint4 dn = make_uint4(0u, 0u, 0u, 0u);
    if (p0 + 1u < p1) dn = piece[p0 + 1u];
    u32 bs = d.y >> 6;
#define Q_LOADU(SLOT, K) { const u32 k_ = (u32)(K);                                                                        \
          const u32 off_ = k_ < nb ? ((((b0 + (k_ << 4)) << 6) + lane) << 4) : 0xFFFFFFF0u;                                   \
          const q64_i32x4 t_ = q64_raw_buffer_load_x4(rsrc, (int)off_, 0, NT ? 2 : 0);                                       \
          U[SLOT] = make_uint4((u32)t_.x, (u32)t_.y, (u32)t_.z, (u32)t_.w); }
#define Q_LOADS(SLOT, USLOT, K) { const u32 rb_ = (u32)__builtin_amdgcn_readlane((int)rbv, (int)min((u32)(K), last));        \
          if (ABL >= 3) { Sa[SLOT] = U[USLOT].y; Sb[SLOT] = U[USLOT].w; }                                                     \
          else if (SLDS) { Sa[SLOT] = Sl[rb_ + soff + (U[USLOT].y >> RSH)]; Sb[SLOT] = Sl[rb_ + soff + (U[USLOT].w >> RSH)]; }  \
          else { const i64* sb_ = S + rb_;                                                                                   \
                 Sa[SLOT] = (unsigned long long)sb_[U[USLOT].y >> RSH]; Sb[SLOT] = (unsigned long long)sb_[U[USLOT].w >> RSH]; } }
#define Q_EAT(E, SA, SB) {                                                                                                   \
          if (ABL == 2 || ABL == 4) { sink += (E).x + (E).z + (SA) + (SB); } else {                                           \
          atomicAdd(&sG[(E).x & CM], SA); atomicAdd(&sG[((E).x >> CB) & CM], SA);                                            \
          atomicAdd(&sG[__builtin_amdgcn_alignbit((E).y, (E).x, 2 * CB) & CM], SA); atomicAdd(&sG[((E).y >> (3 * CB - 32)) & CM], SA); \
          atomicAdd(&sG[(E).z & CM], SB); atomicAdd(&sG[((E).z >> CB) & CM], SB);                                            \
          atomicAdd(&sG[__builtin_amdgcn_alignbit((E).w, (E).z, 2 * CB) & CM], SB); atomicAdd(&sG[((E).w >> (3 * CB - 32)) & CM], SB); } }
    // this wavefront's blocks of the segment (D, BS): b0, b0 + 16, ... ; their row bases and the first A stream units
#define Q_SEGMENT(D, BS) {                                                                                                   \
          const u32 e_ = min((BS) + 1024u, (D).z >> 6);                                                                       \
          b0 = (BS) + wave; nb = b0 < e_ ? (e_ - b0 + 15u) >> 4 : 0u; last = nb ? nb - 1u : 0u;                                \
          rbv = blkrow[min(b0 + (min(lane, last) << 4), blk_last)];                                                          \
          _Pragma("unroll") for (int j_ = 0; j_ < A; j_++) { Q_LOADU(j_, j_); __builtin_amdgcn_sched_barrier(0); } }
    // SLDS (dense strips: every workgroup's rows fit LDS beside its accumulators): the S values of the rows of the
    // workgroup's one or two pieces are copied to LDS once, [piece 0 | piece 1] behind the accumulator regions; a step
    // then reads its S pair with two ds_read_b64 right before it adds -- no global S loads, nothing but stream units in
    // the vector-memory queue.  soff: what turns (block row base + row field) into an index of that copy.
    unsigned long long* const Sl = q64_lds + (u32)R * PD;
    u32 soff = 0, soff1 = 0;
    Q_SEGMENT(d, bs);                                    // the first stream units are on their way while S is copied
    if (SLDS) {
      const uint4 sr = wgrec[3 * blockIdx.x + 2];        // {first row, rows} of piece 0 and of piece 1
      soff = 0u - sr.x; soff1 = sr.y - sr.z;
      const i64* s0 = S + sr.x; const i64* s1 = S + sr.z;
      for (u32 i0 = 0; i0 < sr.y; i0 += 8192u) {
        unsigned long long v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const u32 i = i0 + (u32)u * 1024u + tid; v[u] = i < sr.y ? (unsigned long long)s0[i] : 0ull; }
#pragma unroll
        for (int u = 0; u < 8; u++) { const u32 i = i0 + (u32)u * 1024u + tid; if (i < sr.y) Sl[i] = v[u]; }
      }
      for (u32 i0 = 0; i0 < sr.w; i0 += 8192u) {
        unsigned long long v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const u32 i = i0 + (u32)u * 1024u + tid; v[u] = i < sr.w ? (unsigned long long)s1[i] : 0ull; }
#pragma unroll
        for (int u = 0; u < 8; u++) { const u32 i = i0 + (u32)u * 1024u + tid; if (i < sr.w) Sl[sr.y + i] = v[u]; }
      }
    }
    bool fresh = true;
    for (;;) {
      if (fresh) __syncthreads();                        // the accumulators are zero
      fresh = false;
      if (nb && SLDS) {
        // step k: request U(k + A), then read the S pair of unit k from LDS and add
        for (u32 k0 = 0; k0 < nb; k0 += (u32)NU) {
#pragma unroll
          for (int j = 0; j < NU; j++) {
            const u32 k = k0 + (u32)j;
            Q_LOADU((j + A) % NU, k + (u32)A);
            __builtin_amdgcn_sched_barrier(0);
            if (k < nb) { Q_LOADS(0, j, k); Q_EAT(U[j], Sa[0], Sb[0]); }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else if (nb) {
        // the S pairs of steps 0 .. B-1; the loop's step k then requests S(k + B) and U(k + A) and consumes step k
#pragma unroll
        for (int j = 0; j < B; j++) { Q_LOADS(j % NS, j % NU, (u32)j); __builtin_amdgcn_sched_barrier(0); }
        for (u32 k0 = 0; k0 < nb; k0 += (u32)L) {
#ifdef KLR_ABLATIONS
          if (p == p0 + 1u && bs == (d.y >> 6) && k0 == (u32)L) Q_TS(1);     // the second piece's first L steps are done
#endif
#pragma unroll
          for (int j = 0; j < L; j++) {
            const u32 k = k0 + (u32)j;
            Q_LOADS((j + B) % NS, (j + B) % NU, k + (u32)B);
            __builtin_amdgcn_sched_barrier(0);
            Q_LOADU((j + A) % NU, k + (u32)A);
            __builtin_amdgcn_sched_barrier(0);
            if (k < nb) Q_EAT(U[j % NU], Sa[j % NS], Sb[j % NS]);      // wave-uniform; a step past the end holds zeros
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      // the next segment: the next round of this piece, or the first round of the next piece
      u32 bsn = bs + 1024u, pn = p;
      const bool piece_ends = bsn >= (d.z >> 6);
      uint4 dx = d;
      if (piece_ends) { pn = p + 1u; dx = dn; bsn = dn.y >> 6; }
      const bool more = pn < p1;
      if (more) {
        if (piece_ends) { dn = make_uint4(0u, 0u, 0u, 0u); if (pn + 1u < p1) dn = piece[pn + 1u]; }
        Q_SEGMENT(dx, bsn);
      }
      if (piece_ends) {
#ifdef KLR_ABLATIONS
        if (tid == 0u && blockIdx.x < 1024u) {
          unsigned long long* r_ = &q64_ts[blockIdx.x * 8u];
          r_[6] = (p == p0 ? 0ull : r_[6]) + ((d.z - d.y) >> 6);
          if (p == p0) r_[7] = (unsigned long long)(p1 - p0) | ((unsigned long long)((d.z - d.y) >> 6) << 8);
        }
#endif
        // With R = 2 regions a workgroup's second piece accumulates beside its first: no barrier, no drain -- a wavefront
        // that has finished its blocks of the first piece goes on with the second while the others are still in the
        // first.  A region is written out (all wavefronts at a barrier) only when a later piece needs it, or at the end.
        const u32 q = p - p0;                                              // the piece just finished, 0-based
        const bool need = !more || q + 2u > (u32)R + flushed;              // piece q + 1 wants region (q + 1) % R
        if (need) {
          if (p == p0) Q_TS(2);
          __syncthreads();                                                 // every wavefront has added its last unit of pieces <= q
          if (p == p0) Q_TS(3);
          if (ABL == 2 || ABL == 4) q64_lds[tid] += sink;
          const u32 q_end = more ? q + 2u - (u32)R : q + 1u;                 // pieces flushed .. q_end - 1 are written out
          for (u32 f = flushed; f < q_end; f++) {
            unsigned long long* reg = q64_lds + (f % (u32)R) * PD;
            i64* out = part + (size_t)(p0 + f) * PD;
            if (ABL == 1) { if (reg[tid] == 0x1234567ull) out[tid] = 1; }
            else for (u32 i = tid; i < PD; i += 1024) { out[i] = (i64)reg[i]; reg[i] = 0ull; }   // each thread zeroes what it wrote out
          }
          flushed = q_end;
          if (p == p0) Q_TS(4);
          if (!more) break;
          fresh = true;
        }
        sG = q64_lds + ((q + 1u) % (u32)R) * PD;
        soff = soff1;                                                      // SLDS: at most two pieces
      }
      p = pn; d = dx; bs = bsn;
    }
#undef Q_SEGMENT
#undef Q_EAT
#undef Q_LOADS
#undef Q_LOADU
  }
  Q_TS(5);
  asm volatile("" :: "v"((u32)s_warm), "v"((u32)(s_warm >> 32)));       // keeps the warming load alive
}

static inline Q64View q64_view_of(const klr_mat* mat) {
  Q64View v; v.nwg = mat->nwg; v.t_piece = mat->t_piece; v.t_wgpiece = mat->t_wgpiece; v.slds_rows = mat->q64_slds_rows;
  return v;
}

template <int CB, bool NT, int A, int B, int R, int ABL = 0, bool SLDS = false>
static int launch_sweep_q64(klr_ctx* ctx, const klr_mat* mat, const Q64View& v, const i64* S, i64* part, hipEvent_t e0, hipEvent_t e1) {
  // one workgroup per CU: the request is padded past half of the CU's 160 KiB, so the dispatcher cannot put two
  // workgroups on one CU while another CU stays empty (it does: with 256 workgroups of 32 - 64 KiB on 256 CUs part of
  // them ran in pairs and finished 15 % later than the rest, tools/q64_phases.py)
  size_t lds_bytes = (size_t)R * ((size_t)8 << CB) + (SLDS ? (size_t)v.slds_rows * 8 : 0);
  if ((int)v.nwg <= ctx->n_cu) lds_bytes = std::max(lds_bytes, (size_t)ctx->sw.q64_lds_kib << 10);
  KLR_TRY(ctx_func_lds(ctx, (const void*)k_sweep_q64<CB, NT, A, B, R, ABL, SLDS>, 160 << 10));
  // the extended launch (events riding on the dispatch packet) only while profiling: it costs a completion signal per
  // kernel, which shows as idle time before and after the dispatch
  if (e0 || e1)
    hipExtLaunchKernelGGL((k_sweep_q64<CB, NT, A, B, R, ABL, SLDS>), dim3(v.nwg), dim3(1024), (u32)lds_bytes, ctx->stream, e0, e1, 0u,
                          (const uint4*)mat->t_ent, (u32)((size_t)mat->total4 * 16), (const u32*)mat->t_blkrow, (const uint4*)v.t_piece,
                          (const uint4*)v.t_wgpiece, S, part);
  else
    hipLaunchKernelGGL((k_sweep_q64<CB, NT, A, B, R, ABL, SLDS>), dim3(v.nwg), dim3(1024), (u32)lds_bytes, ctx->stream,
                       (const uint4*)mat->t_ent, (u32)((size_t)mat->total4 * 16), (const u32*)mat->t_blkrow, (const uint4*)v.t_piece,
                       (const uint4*)v.t_wgpiece, S, part);
  return KLR_OK;
}

// pipeline depth by the number of workgroups the ingest cut the stream for (one or two per CU)
template <int CB, bool NT, int ABL = 0>
static int launch_sweep_q64_by_grid(klr_ctx* ctx, const klr_mat* mat, const Q64View& v, const i64* S, i64* part, hipEvent_t e0, hipEvent_t e1) {
#ifdef KLR_ABLATIONS
  if (ABL == 0) {
    const char* e = ctx_opt(ctx, "KLR_Q64_DEPTH");                // A/B: pipeline depths
    const int dsel = e ? atoi(e) : 0;
    if (dsel == 10) return launch_sweep_q64<CB, NT, 1, 0, 2, 0>(ctx, mat, v, S, part, e0, e1);
    if (dsel == 20) return launch_sweep_q64<CB, NT, 2, 0, 2, 0>(ctx, mat, v, S, part, e0, e1);
    if (dsel == 21) return launch_sweep_q64<CB, NT, 2, 1, 2, 0>(ctx, mat, v, S, part, e0, e1);
    if (dsel == 30) return launch_sweep_q64<CB, NT, 3, 0, 2, 0>(ctx, mat, v, S, part, e0, e1);
    if (dsel == 32) return launch_sweep_q64<CB, NT, 3, 2, 2, 0>(ctx, mat, v, S, part, e0, e1);
    if (dsel == 41) return launch_sweep_q64<CB, NT, 4, 1, 2, 0>(ctx, mat, v, S, part, e0, e1);
    if (dsel == 42) return launch_sweep_q64<CB, NT, 4, 2, 2, 0>(ctx, mat, v, S, part, e0, e1);
    if (dsel == 31) return launch_sweep_q64<CB, NT, 3, 1, 2, 0>(ctx, mat, v, S, part, e0, e1);
    if (dsel == 52) return launch_sweep_q64<CB, NT, 5, 2, 2, 0>(ctx, mat, v, S, part, e0, e1);
    if (dsel == 73) return launch_sweep_q64<CB, NT, 7, 3, 2, 0>(ctx, mat, v, S, part, e0, e1);
    if (dsel == 115) return launch_sweep_q64<CB, NT, 11, 5, 2, 0>(ctx, mat, v, S, part, e0, e1);
    if (dsel == 71) return launch_sweep_q64<CB, NT, 7, 1, 2, 0>(ctx, mat, v, S, part, e0, e1);
  }
#endif
  if ((int)v.nwg <= ctx->n_cu) {
#ifdef KLR_ABLATIONS
    if (ctx_opt(ctx, "KLR_Q64_REGIONS") && atoi(ctx_opt(ctx, "KLR_Q64_REGIONS")) == 1) return launch_sweep_q64<CB, NT, 2, 1, 1, ABL>(ctx, mat, v, S, part, e0, e1);
#endif
    if (v.slds_rows && ABL == 0) {                           // dense strips: the workgroup's S values in LDS
      if (ctx->sw.q64_slds_a == 2) return launch_sweep_q64<CB, NT, 2, 0, 2, 0, true>(ctx, mat, v, S, part, e0, e1);
      return launch_sweep_q64<CB, NT, 3, 0, 2, 0, true>(ctx, mat, v, S, part, e0, e1);
    }
    return launch_sweep_q64<CB, NT, 2, 1, 2, ABL>(ctx, mat, v, S, part, e0, e1);   // one workgroup per CU: two accumulator regions
  }
  return launch_sweep_q64<CB, NT, 1, 0, 1, ABL>(ctx, mat, v, S, part, e0, e1);
}

template <int THREADS, int UNROLL, bool NT, int ABL = 0, int FMT = FMT_U32>
static int launch_sweep_coo(klr_ctx* ctx, const klr_mat* mat, const i64* S, i64* part, hipEvent_t e0, hipEvent_t e1) {
  const size_t lds_bytes = (size_t)TILE_DIM * 16;
  KLR_TRY(ctx_func_lds(ctx, (const void*)k_sweep_coo<THREADS, UNROLL, NT, ABL, FMT>, (int)lds_bytes));
  // start / stop events ride on the dispatch packet itself (hipExtLaunchKernelGGL): recording them with
  // hipEventRecord puts two barrier packets around the kernel and costs ~15 us per step
  hipExtLaunchKernelGGL((k_sweep_coo<THREADS, UNROLL, NT, ABL, FMT>), dim3(mat->nwg), dim3(THREADS), (u32)lds_bytes, ctx->stream, e0, e1, 0u,
                        (const u32*)mat->t_ent, (const uint4*)mat->t_piece, (const uint4*)mat->t_wgpiece, (const u32*)mat->t_tileq, mat->nrt,
                        mat->rt_major ? mat->nct : 0u, mat->tile_r, mat->nrt * mat->nct, mat->n, S, part);
  return KLR_OK;
}

// Kernel of the shipped configuration per entry format.  The in-process A/B variants and the timing-only ablations
// (which give wrong results by construction) exist only in a build with -DKLR_ABLATIONS (tools/ab_sweep.py).
#ifdef KLR_ABLATIONS
static int g_coo_variant = -1;
extern "C" void klr_debug_set_coo_variant(int v) { g_coo_variant = v; }
static int dispatch_ablation(klr_ctx* ctx, const klr_mat* mat, const Q64View& v, const i64* S, i64* part, hipEvent_t e0, hipEvent_t e1, bool* done) {
  if (g_coo_variant < 0) { const char* e = ctx_opt(ctx, "KLR_COO_VARIANT"); g_coo_variant = e ? atoi(e) : 0; }
  *done = true;
  if (mat->fmt == FMT_Q64C12 || mat->fmt == FMT_Q64C13) {
    const char* e = ctx_opt(ctx, "KLR_Q64_ABL");
    const int abl = e ? atoi(e) : 0;
    const bool c13 = mat->fmt == FMT_Q64C13;
    switch (abl) {
      case 1: return c13 ? launch_sweep_q64_by_grid<13, true, 1>(ctx, mat, v, S, part, e0, e1) : launch_sweep_q64_by_grid<12, true, 1>(ctx, mat, v, S, part, e0, e1);
      case 2: return c13 ? launch_sweep_q64_by_grid<13, true, 2>(ctx, mat, v, S, part, e0, e1) : launch_sweep_q64_by_grid<12, true, 2>(ctx, mat, v, S, part, e0, e1);
      case 3: return c13 ? launch_sweep_q64_by_grid<13, true, 3>(ctx, mat, v, S, part, e0, e1) : launch_sweep_q64_by_grid<12, true, 3>(ctx, mat, v, S, part, e0, e1);
      case 4: return c13 ? launch_sweep_q64_by_grid<13, true, 4>(ctx, mat, v, S, part, e0, e1) : launch_sweep_q64_by_grid<12, true, 4>(ctx, mat, v, S, part, e0, e1);
      default: break;
    }
    *done = false;
    return KLR_OK;
  }
  if (mat->fmt == FMT_R64) {
    switch (g_coo_variant) {
      case 26: return launch_sweep_coo<1024, 2, true, 3, FMT_R64>(ctx, mat, S, part, e0, e1);   // stream only
      case 14: return launch_sweep_coo<1024, 2, false, 0, FMT_R64>(ctx, mat, S, part, e0, e1);
      case 13: return launch_sweep_coo<512, 2, true, 0, FMT_R64>(ctx, mat, S, part, e0, e1);
      case 12: return launch_sweep_coo<1024, 1, true, 0, FMT_R64>(ctx, mat, S, part, e0, e1);   // no prefetch
      default: break;
    }
  } else if (mat->fmt == FMT_U24) {
    switch (g_coo_variant) {
      case 26: return launch_sweep_coo<1024, 1, true, 3, FMT_U24>(ctx, mat, S, part, e0, e1);   // stream only
      case 14: return launch_sweep_coo<1024, 1, false, 0, FMT_U24>(ctx, mat, S, part, e0, e1);
      case 13: return launch_sweep_coo<512, 1, true, 0, FMT_U24>(ctx, mat, S, part, e0, e1);
      case 12: return launch_sweep_coo<1024, 1, true, 0, FMT_U24>(ctx, mat, S, part, e0, e1);   // one load per trip, no prefetch
      default: break;
    }
  } else if (mat->fmt == FMT_U32) {
    switch (g_coo_variant) {
      case 1:  return launch_sweep_coo<512, 4, true>(ctx, mat, S, part, e0, e1);
      case 2:  return launch_sweep_coo<512, 8, false>(ctx, mat, S, part, e0, e1);
      case 4:  return launch_sweep_coo<1024, 4, false>(ctx, mat, S, part, e0, e1);
      case 6:  return launch_sweep_coo<1024, 2, false>(ctx, mat, S, part, e0, e1);
      case 8:  return launch_sweep_coo<512, 4, true, 1>(ctx, mat, S, part, e0, e1);     // ablation: no atomics
      case 9:  return launch_sweep_coo<512, 4, true, 2>(ctx, mat, S, part, e0, e1);     // ablation: no s gather
      case 11: return launch_sweep_coo<1024, 2, true>(ctx, mat, S, part, e0, e1);
      case 13: return launch_sweep_coo<512, 1, true>(ctx, mat, S, part, e0, e1);
      case 14: return launch_sweep_coo<1024, 1, false>(ctx, mat, S, part, e0, e1);
      case 21: return launch_sweep_coo<1024, 1, true, 1>(ctx, mat, S, part, e0, e1);     // ablations of the shipped shape: no atomics
      case 22: return launch_sweep_coo<1024, 1, true, 2>(ctx, mat, S, part, e0, e1);     //   no s gather
      case 26: return launch_sweep_coo<1024, 1, true, 3>(ctx, mat, S, part, e0, e1);     //   stream only
      case 50: return launch_sweep_coo<1024, 1, true, 4>(ctx, mat, S, part, e0, e1);     //   no multiply
      default: break;
    }
  }
  *done = false;
  return KLR_OK;
}
#endif

static int dispatch_sweep_coo(klr_ctx* ctx, const klr_mat* mat, const i64* S, i64* part, hipEvent_t e0, hipEvent_t e1, const Q64View* view = nullptr) {
  const Q64View v = view ? *view : q64_view_of(mat);
#ifdef KLR_ABLATIONS
  { bool done; const int rc = dispatch_ablation(ctx, mat, v, S, part, e0, e1, &done); if (done) return rc; }
#endif
  // The stream is read with non-temporal loads whatever its size.  A stream that fits the 256 MiB Infinity Cache would
  // stay resident from sweep to sweep under plain loads (configs[2]: 45.7 us back to back against 47.7), but a plain
  // stream that has to come from HBM takes 66 us (it is installed in the Infinity Cache on its way) against 49 us
  // non-temporal, and inside a whole leapfrog run -- sub-select and reduced solve between two sweeps -- the plain kernel
  // averaged 48.9 us: no better than non-temporal, with a 35 % cliff behind it (profiles/README.md, round 2).
  // KLR_STREAM_NT=0 reads it with plain loads (A/B tooling).
  bool fits_mall = false;
  if (ctx->sw.stream_nt == 0) fits_mall = (size_t)mat->total4 * 16 <= ((size_t)220 << 20);
  switch (mat->fmt) {
    case FMT_Q64C12: return fits_mall ? launch_sweep_q64_by_grid<12, false>(ctx, mat, v, S, part, e0, e1) : launch_sweep_q64_by_grid<12, true>(ctx, mat, v, S, part, e0, e1);
    case FMT_Q64C13: return fits_mall ? launch_sweep_q64_by_grid<13, false>(ctx, mat, v, S, part, e0, e1) : launch_sweep_q64_by_grid<13, true>(ctx, mat, v, S, part, e0, e1);
    case FMT_R64:
      // 32 waves/CU, ping-pong prefetch.  The prefetch loop runs a uniform, even number of trips per tile with masked
      // lanes: only worth it for long tiles (a 1M-column shard has ~2.6 trips per tile: 116 us with it, 103 us without)
      if (mat->total4 < (i64)8192 * mat->nrt * mat->nct) return launch_sweep_coo<1024, 1, true, 0, FMT_R64>(ctx, mat, S, part, e0, e1);
      if (fits_mall) return launch_sweep_coo<1024, 2, false, 0, FMT_R64>(ctx, mat, S, part, e0, e1);
      return launch_sweep_coo<1024, 2, true, 0, FMT_R64>(ctx, mat, S, part, e0, e1);
    case FMT_U24: return launch_sweep_coo<1024, 2, true, 0, FMT_U24>(ctx, mat, S, part, e0, e1);   // nt loads, ping-pong prefetch
    default:      return launch_sweep_coo<1024, 1, true>(ctx, mat, S, part, e0, e1);             // nt loads
  }
}

// ibuf layout (int64): [0, dim) column sums (slot 0 unused) | dim: bias hi | dim+1: bias lo |
//                      dim+2: loss limb0 | dim+3: loss limb1 | dim+4: non-finite loss terms

// 256 columns x 4 piece lanes per block: every lane adds every 4th piece of its column tile (4 independent
// loads in flight), LDS combines.  FUSE (no all-reduce between the two steps, i.e. one rank and no hook): the
// sums are converted to f64 right here and the selection keys + coarse histogram are emitted, as k_int_to_g does.
// FLAT (at most 8 pieces per column tile -- a q64 stream of a wide matrix has two or three per strip -- and ITER a
// multiple of 4): a thread sums ALL pieces of its column (8 guarded loads in flight) and a workgroup's ITER x 256 columns
// go in ITER / 4 passes of 1024 with nothing between them; the four-lane form spends a barrier pair and three dependent
// round trips per 256 columns, 17 us for the 1M columns of a configs[3] shard.
template <bool FUSE, bool FLAT = false>
// Columns [j0, dim) are summed (a strip range of the overlapped form: its first column .. its end; otherwise 0 .. dim) and
// the five scalars go to ibuf[dim ..]: a ranged launch is handed ibuf shifted so that its own vector starts at j0.
__global__ __launch_bounds__(1024) void k_finish_coo(const i64* __restrict__ part, const u32* __restrict__ ctptr,
                                                     const u32* __restrict__ ctlist, i64 j0, i64 dim, i64* __restrict__ ibuf,
                                                     const i64* __restrict__ rowblk, i64 nrowblk, u32 CT, u32 PSTRIDE,
                                                     double inv_scale, double* __restrict__ g, u64* __restrict__ keys,
                                                     u32* __restrict__ hist, u64 tau, Cand* __restrict__ cand, u32* __restrict__ cursor,
                                                     u32 ITER, u32 lean) {
  __shared__ i64 sh[4][256];
  __shared__ u32 shh[FUSE ? FAST_BINS : 1];
  // hinted compaction: the workgroup gathers its candidates in LDS and claims their slots with ONE atomic on the global
  // cursor (a thousand same-address atomics, one per candidate, took 15 us of this kernel at a million columns)
  __shared__ Cand wcand[FUSE ? 256 : 1];
  __shared__ u32 wcnt, wbase;
  if (threadIdx.x == 0) wcnt = 0u;
  if (blockIdx.x == gridDim.x - 1) {
    // last block: bias halves, loss limbs and the non-finite count from the row pass's block partials
    if (rowblk) {
      const int q = threadIdx.x & 7, l = threadIdx.x >> 3;         // 5 quantities x 128 lanes
      i64 v = 0;
      if (q < 5) for (i64 b = l; b < nrowblk; b += 128) v += rowblk[b * 5 + q];
      __shared__ i64 sq[8][128];
      __shared__ i64 tot[5];
      sq[q][l] = v;
      __syncthreads();
      if (threadIdx.x < 5) {
        i64 t = 0;
        for (int i = 0; i < 128; i++) t += sq[threadIdx.x][i];
        if (FUSE) tot[threadIdx.x] = t;
        else ibuf[dim + threadIdx.x] = t;                 // sole writer: no zeroing needed beforehand
      }
      if (FUSE) {
        __syncthreads();
        if (threadIdx.x == 0) {
          g[0] = ((double)tot[0] * 4294967296.0 + (double)tot[1]) * inv_scale;
          const double l2 = ((double)tot[2] + (double)tot[3] * (1.0 / 4398046511104.0)) * (1.0 / 1048576.0);
          g[dim] = tot[4] ? nan("") : l2;
        }
      }
    } else if (FUSE && threadIdx.x == 0) {                // k_fix_s has accumulated the five extras in ibuf
      g[0] = ((double)ibuf[dim] * 4294967296.0 + (double)ibuf[dim + 1]) * inv_scale;
      const double l2 = ((double)ibuf[dim + 2] + (double)ibuf[dim + 3] * (1.0 / 4398046511104.0)) * (1.0 / 1048576.0);
      g[dim] = ibuf[dim + 4] ? nan("") : l2;
    }
    return;
  }
  const int cx = threadIdx.x & 255, ly = threadIdx.x >> 8;
  if (FUSE && hist) { for (int i = threadIdx.x; i < FAST_BINS; i += 1024) shh[i] = 0u; }
  // what happens to a finished column sum
  auto emit = [&](i64 j, i64 sum) {
    if (FUSE) {
      const double gv = (double)sum * inv_scale;                // inv_scale is a power of two: exact
      if (!lean) g[j] = gv;                                     // lean (SelHint): candidates only
      if (keys) {
        const u64 kk = abs_key(gv);
        if (!lean) keys[j - 1] = kk;
        if (hist) atomicAdd(&shh[(u32)(kk >> FAST_SHIFT)], 1u);
        if (tau != 0ull && kk >= tau) {                       // hinted compaction (SelHint)
          const u32 wp = atomicAdd(&wcnt, 1u);
          if (wp < 256u) { wcand[wp].key = kk; wcand[wp].idx = (u32)(j - 1); wcand[wp].pad = 0; }
          else {
            const u32 pos = atomicAdd(cursor, 1u);
            if (pos < GCAND_CAP) { cand[pos].key = kk; cand[pos].idx = (u32)(j - 1); cand[pos].pad = 0; }
          }
        }
      }
    } else ibuf[j] = sum;
  };
  if (FLAT) {
    if (FUSE) __syncthreads();                                  // wcnt and the histogram are zero
    i64 jv[2], sums[2];
#pragma unroll
    for (int ps = 0; ps < 2; ps++) {
      const i64 j = j0 + (i64)blockIdx.x * ITER * 256 + (i64)ps * 1024 + threadIdx.x;
      jv[ps] = j; sums[ps] = 0;
      if ((u32)ps * 4u < ITER && j < dim) {
        const u32 ct = (u32)j / CT, c = (u32)j - ct * CT;
        const u32 k0 = ctptr[ct], k1 = ctptr[ct + 1];
        u32 p[8]; i64 a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) p[u] = (k0 + (u32)u < k1) ? ctlist[k0 + (u32)u] : 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = (p[u] != 0xFFFFFFFFu) ? part[(i64)p[u] * PSTRIDE + c] : 0;
        sums[ps] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
      }
    }
#pragma unroll
    for (int ps = 0; ps < 2; ps++) if ((u32)ps * 4u < ITER && jv[ps] < dim && jv[ps] > 0) emit(jv[ps], sums[ps]);
    __syncthreads();
  } else
  for (u32 it = 0; it < ITER; it++) {
  const i64 j = j0 + ((i64)blockIdx.x * ITER + it) * 256 + cx;
  i64 v = 0;
  if (j < dim) {
    const u32 ct = (u32)j / CT, c = (u32)j - ct * CT;
    const u32 k1 = ctptr[ct + 1];
    for (u32 k = ctptr[ct] + ly; k < k1; k += 32) {      // 8 guarded loads in flight per lane
      u32 p[8]; i64 a[8];
#pragma unroll
      for (int u = 0; u < 8; u++) p[u] = (k + 4u * u < k1) ? ctlist[k + 4u * u] : 0xFFFFFFFFu;
#pragma unroll
      for (int u = 0; u < 8; u++) a[u] = (p[u] != 0xFFFFFFFFu) ? part[(i64)p[u] * PSTRIDE + c] : 0;
      v += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
  }
  sh[ly][cx] = v;
  __syncthreads();
  if (ly == 0 && j < dim && j > 0) emit(j, (sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx]));
  __syncthreads();                                            // sh is reused by the next column group
  }
  if (FUSE && tau != 0ull) {
    if (threadIdx.x == 0) { const u32 nw = min(wcnt, 256u); wbase = nw ? atomicAdd(cursor, nw) : 0u; }
    __syncthreads();
    const u32 nw = min(wcnt, 256u);
    if (threadIdx.x < nw && wbase + threadIdx.x < GCAND_CAP) cand[wbase + threadIdx.x] = wcand[threadIdx.x];
  }
  if (FUSE && hist) {
    __syncthreads();
    for (int i = threadIdx.x; i < FAST_BINS; i += 1024) if (shh[i]) atomicAdd(&hist[i], shh[i]);
  }
}

// s -> int64 fixed point; bias sum and loss as exactly additive integer limbs
__global__ __launch_bounds__(256) void k_fix_s(const double* __restrict__ s, const double* __restrict__ lterm, i64 n,
                                               double scale, i64* __restrict__ S, unsigned long long* __restrict__ acc) {
  __shared__ i64 sh[4][256 / 64 + 1];
  __shared__ u32 shbad[256 / 64];
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  i64 hi = 0, lo = 0, l0 = 0, l1 = 0; u32 bad = 0;
  if (i < n) {
    const i64 v = __double2ll_rn(s[i] * scale);
    S[i] = v;
    hi = v >> 32; lo = (i64)(v & 0xFFFFFFFFll);
    if (lterm) {
      const double t = lterm[i];
      if (t >= 0.0 && t < 1099511627776.0) {            // finite and < 2^40
        const double ts = t * 1048576.0;                  // * 2^20, exact
        const double f = floor(ts);
        l0 = (i64)f;
        l1 = __double2ll_rn((ts - f) * 4398046511104.0);  // * 2^42
      } else bad = 1;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    hi += __shfl_xor(hi, o, 64); lo += __shfl_xor(lo, o, 64);
    l0 += __shfl_xor(l0, o, 64); l1 += __shfl_xor(l1, o, 64);
    bad += __shfl_xor(bad, o, 64);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][w] = hi; sh[1][w] = lo; sh[2][w] = l0; sh[3][w] = l1; shbad[w] = bad; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const i64 t = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
    if (t) atomicAdd(&acc[threadIdx.x], (unsigned long long)t);
  }
  if (threadIdx.x == 4) {
    const u32 t = shbad[0] + shbad[1] + shbad[2] + shbad[3];
    if (t) atomicAdd(&acc[4], (unsigned long long)t);
  }
}

__global__ __launch_bounds__(256) void k_int_to_g(const i64* __restrict__ ibuf, i64 dim, double inv_scale,
                                                  double* __restrict__ g, u64* __restrict__ keys, u32* __restrict__ hist,
                                                  u64 tau, Cand* __restrict__ cand, u32* __restrict__ cursor, u32 lean) {
  // grid-stride; the coarse key histogram is privatised in LDS (most keys share a few bins); hinted candidates are
  // gathered per workgroup and claim their slots with one atomic on the global cursor
  __shared__ u32 sh[FAST_BINS];
  __shared__ Cand wcand[256];
  __shared__ u32 wcnt, wbase;
  if (threadIdx.x == 0) wcnt = 0u;
  __syncthreads();
  if (hist) {
    for (int i = threadIdx.x; i < FAST_BINS; i += 256) sh[i] = 0;
    __syncthreads();
  }
  for (i64 j = (i64)blockIdx.x * 256 + threadIdx.x; j <= dim; j += (i64)gridDim.x * 256) {
    if (j == 0) {
      g[0] = ((double)ibuf[dim] * 4294967296.0 + (double)ibuf[dim + 1]) * inv_scale;
    } else if (j < dim) {
      const double v = (double)ibuf[j] * inv_scale;           // inv_scale is a power of two: exact
      if (!lean) g[j] = v;                                    // lean (SelHint): candidates only
      if (keys) {                                             // selection key of feature j-1
        const u64 k = abs_key(v);
        if (!lean) keys[j - 1] = k;
        if (hist) atomicAdd(&sh[(u32)(k >> FAST_SHIFT)], 1u);
        if (tau != 0ull && k >= tau) {                          // hinted compaction (SelHint)
          const u32 wp = atomicAdd(&wcnt, 1u);
          if (wp < 256u) { wcand[wp].key = k; wcand[wp].idx = (u32)(j - 1); wcand[wp].pad = 0; }
          else {
            const u32 pos = atomicAdd(cursor, 1u);
            if (pos < GCAND_CAP) { cand[pos].key = k; cand[pos].idx = (u32)(j - 1); cand[pos].pad = 0; }
          }
        }
      }
    } else {
      // loss sum = (limb0 + limb1 * 2^-42) * 2^-20; NaN if any term was not finite
      const double l = ((double)ibuf[dim + 2] + (double)ibuf[dim + 3] * (1.0 / 4398046511104.0)) * (1.0 / 1048576.0);
      g[dim] = ibuf[dim + 4] ? nan("") : l;
    }
  }
  if (tau != 0ull) {
    __syncthreads();
    if (threadIdx.x == 0) { const u32 nw = min(wcnt, 256u); wbase = nw ? atomicAdd(cursor, nw) : 0u; }
    __syncthreads();
    const u32 nw = min(wcnt, 256u);
    if (threadIdx.x < nw && wbase + threadIdx.x < GCAND_CAP) cand[wbase + threadIdx.x] = wcand[threadIdx.x];
  }
  if (hist) {
    __syncthreads();
    for (int i = threadIdx.x; i < FAST_BINS; i += 256) if (sh[i]) atomicAdd(&hist[i], sh[i]);
  }
}

// Fixed-point scale: 2^E with max|s| * A * 2^E < 2^62, A = bound on the largest non-bias column sum
static double coo_scale(double smax, double amax) {
  double b = smax * std::max(amax, 1.0);
  if (!(b > 0.0) || !std::isfinite(b)) b = 1.0;
  int e;
  (void)std::frexp(b, &e);           // b < 2^e
  int E = 62 - e;
  if (E > 1000) E = 1000;
  if (E < -1000) E = -1000;
  return std::ldexp(1.0, E);
}

// Which protocol the full-width sweep of this matrix runs -- decided ONCE per matrix and TOGETHER by all ranks.
// Every rank builds the tiled integer stream on its own (count data, n >= 2048, nnz >= 2^20); a shard that falls on
// the other side of that test (a short last shard, one non-count value) would otherwise post an f64 all-reduce of
// dim+1 elements against its neighbours' int64 all-reduce of dim+5.  So the first sweep on a matrix all-reduces
// [column-sum bound | 1 if this shard has no stream]: the integer protocol is used only if EVERY shard has the stream,
// else every rank runs the plain CSC sweep with the f64 all-reduce (all matrices keep their CSC).  The same
// collective carries the bound on the column sums the fixed-point scale needs (sum of the per-rank maxima).
int sweep_protocol(klr_ctx* ctx, const klr_mat* mat, bool* use_int) {
  klr_mat* mm = const_cast<klr_mat*>(mat);
  if (mm->sweep_int < 0) {
    double v[2] = {mat->tiled ? mat->col_abs_max : 0.0, mat->tiled ? 0.0 : 1.0};
    if (ctx->ar_fn) {
      double* t;
      KLR_TRY(ws_get(ctx, WS_BLK, 64, (void**)&t));
      KLR_TRY(h2d(ctx, t, v, 2 * sizeof(double)));
      KLR_TRY(klr_allreduce(ctx, t, 2, KLR_F64));
      KLR_TRY(d2h_sync(ctx, v, t, 2 * sizeof(double)));
    }
    mm->col_abs_max_global = v[0];
    mm->sweep_int = (mat->tiled && v[1] == 0.0) ? 1 : 0;
  }
  *use_int = mm->sweep_int == 1;
  return KLR_OK;
}

// scale, S and the integer accumulator block of one count-data sweep
int sweep_fix_begin(klr_ctx* ctx, const klr_mat* mat, const double cw[2], double* scale_out, i64** S_out,
                    unsigned long long** acc_out) {
  const i64 n = mat->n, dim = mat->dim;
  bool use_int;
  KLR_TRY(sweep_protocol(ctx, mat, &use_int));
  if (!use_int) return klr_fail(ctx, KLR_EINVAL, "internal: integer sweep on a matrix negotiated to the f64 protocol");
  const double smax = std::max(std::fabs(cw[0]), std::fabs(cw[1])) / (double)mat->n_total;
  *scale_out = coo_scale(smax, mat->col_abs_max_global);
  i64 *S, *ibuf;
  KLR_TRY(ws_get(ctx, WS_Z, (size_t)std::max<i64>(n, 1) * 8, (void**)&S));
  KLR_TRY(ws_get(ctx, WS_TMP2, (size_t)(dim + IB_EXTRA) * 8, (void**)&ibuf));
  *S_out = S;
  *acc_out = (unsigned long long*)(ibuf + dim);
  return KLR_OK;
}

// piece vectors -> column sums: converted, keyed and (hinted) compacted right here on one rank, left as integers for the
// collective otherwise
static int finish_launch(klr_ctx* ctx, const klr_mat* mat, const i64* part, i64* ibuf, double scale, double* gbuf, u64* keys, u32* hist,
                         const unsigned long long* rowblk, i64 nrowblk, u64 tau, Cand* hcand, u32* hcur, bool lean) {
  const i64 dim = mat->dim;
  hipStream_t st = ctx->stream;
  const u32 pstride = mat->fmt == FMT_Q64C13 ? 8192u : TILE_DIM;      // width of a piece vector
  const u32 ln = lean ? 1u : 0u;
  // 256 columns per pass of a workgroup; wide matrices take up to 8 passes per workgroup (fewer, fuller candidate claims)
  u32 fiter = (u32)std::max<i64>(1, std::min<i64>(8, dim / (256 * 512)));
  if (ctx->sw.finish_iter > 0) fiter = (u32)std::max(1, std::min(8, ctx->sw.finish_iter));          // A/B tooling
  // (one pass of 1024 columns per workgroup, 8x the workgroups, measured 3x slower on the 1M-column step: keep ITER)
  const bool flat = mat->max_ct_pieces >= 1 && mat->max_ct_pieces <= 8 && fiter % 4 == 0 && !ctx->sw.finish_no_flat;
  const dim3 fgrid((unsigned)ceil_div(dim, (i64)256 * fiter) + 1);
  if (!ctx->ar_fn) {
    // one rank, no hook: nothing happens between the piece sums and their conversion -- one kernel
    if (flat) hipLaunchKernelGGL((k_finish_coo<true, true>), fgrid, dim3(1024), 0, st, part, mat->t_ctptr, mat->t_ctlist, (i64)0, dim, ibuf, (const i64*)rowblk,
                                 nrowblk, mat->tile_c, pstride, 1.0 / scale, gbuf, keys, hist, tau, hcand, hcur, fiter, ln);
    else hipLaunchKernelGGL((k_finish_coo<true, false>), fgrid, dim3(1024), 0, st, part, mat->t_ctptr, mat->t_ctlist, (i64)0, dim, ibuf, (const i64*)rowblk,
                            nrowblk, mat->tile_c, pstride, 1.0 / scale, gbuf, keys, hist, tau, hcand, hcur, fiter, ln);
  } else {
    if (flat) hipLaunchKernelGGL((k_finish_coo<false, true>), fgrid, dim3(1024), 0, st, part, mat->t_ctptr, mat->t_ctlist, (i64)0, dim, ibuf, (const i64*)rowblk,
                                 nrowblk, mat->tile_c, pstride, 1.0 / scale, gbuf, keys, hist, 0ull, (Cand*)nullptr, (u32*)nullptr, fiter, 0u);
    else hipLaunchKernelGGL((k_finish_coo<false, false>), fgrid, dim3(1024), 0, st, part, mat->t_ctptr, mat->t_ctlist, (i64)0, dim, ibuf, (const i64*)rowblk,
                            nrowblk, mat->tile_c, pstride, 1.0 / scale, gbuf, keys, hist, 0ull, (Cand*)nullptr, (u32*)nullptr, fiter, 0u);
  }
  KLR_HIP(ctx, hipGetLastError());
  return KLR_OK;
}

// integer sweep over the packed stream, int64 all-reduce of [column sums | bias | loss], back to f64
int sweep_fix_finish(klr_ctx* ctx, const klr_mat* mat, double scale, double* gbuf, u64* keys, u32* hist,
                     const unsigned long long* rowblk, i64 nrowblk, const SelHint* hint, i64** ibuf_out, i64 ibuf_elems) {
  const u64 tau = hint && keys ? hint->tau : 0ull;
  if (tau) hist = nullptr;                       // hinted: the coarse histogram is only built if the hint fails (select_from_device)
  Cand* hcand = hint ? hint->cand : nullptr;
  u32* hcur = hint ? &hint->ctl->cursor : nullptr;
  const i64 n = mat->n, dim = mat->dim;
  hipStream_t st = ctx->stream;
  i64 *S, *part, *ibuf;
  KLR_TRY(ws_get(ctx, WS_Z, (size_t)std::max<i64>(n, 1) * 8, (void**)&S));
  KLR_TRY(ws_get(ctx, WS_TMP2, (size_t)std::max<i64>(dim + IB_EXTRA, ibuf_elems) * 8, (void**)&ibuf));
  const u32 pstride = mat->fmt == FMT_Q64C13 ? 8192u : TILE_DIM;      // width of a piece vector
  KLR_TRY(ws_get(ctx, WS_PART, (size_t)std::max<u32>(mat->npiece, 1) * pstride * 8, (void**)&part));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool timed = prof_take(ctx);
  if (timed) {
    KLR_TRY(prof_event_get(ctx, &e0));
    KLR_TRY(prof_event_get(ctx, &e1));
  }
  KLR_TRY(dispatch_sweep_coo(ctx, mat, S, part, e0, e1));
  if (timed) ctx->prof_ev.emplace_back(e0, e1);
  KLR_TRY(finish_launch(ctx, mat, part, ibuf, scale, gbuf, keys, hist, rowblk, nrowblk, tau, hcand, hcur, hint && hint->lean));
  if (ctx->ar_fn) {
    if (ibuf_out) {                                // the caller runs the collective (reduce-scatter form, klr_select.hip)
      if (ibuf_elems > dim + IB_EXTRA) KLR_HIP(ctx, hipMemsetAsync(ibuf + dim + IB_EXTRA, 0, (size_t)(ibuf_elems - dim - IB_EXTRA) * 8, st));
      *ibuf_out = ibuf;
      return KLR_OK;
    }
    return sweep_fix_tail(ctx, mat, scale, ibuf, gbuf, keys, hist, hint);
  }
  return KLR_OK;
}

// The conversion stage alone, in full: after a lean hinted sweep whose hint failed (klr_select.hip).  One rank: the piece
// vectors are still in WS_PART; several: the all-reduced sums are still in WS_TMP2.
int sweep_fix_emit(klr_ctx* ctx, const klr_mat* mat, double scale, double* gbuf, u64* keys, u32* hist,
                   const unsigned long long* rowblk, i64 nrowblk) {
  const i64 dim = mat->dim;
  i64 *part, *ibuf;
  KLR_TRY(ws_get(ctx, WS_TMP2, (size_t)(dim + IB_EXTRA) * 8, (void**)&ibuf));
  if (!ctx->ar_fn) {
    const u32 pstride = mat->fmt == FMT_Q64C13 ? 8192u : TILE_DIM;
    KLR_TRY(ws_get(ctx, WS_PART, (size_t)std::max<u32>(mat->npiece, 1) * pstride * 8, (void**)&part));
    return finish_launch(ctx, mat, part, ibuf, scale, gbuf, keys, hist, rowblk, nrowblk, 0ull, nullptr, nullptr, false);
  }
  hipLaunchKernelGGL(k_int_to_g, dim3((unsigned)std::max<i64>(1, std::min<i64>(ceil_div(dim + 1, 1024), 256))), dim3(256), 0, ctx->stream, ibuf, dim,
                     1.0 / scale, gbuf, keys, hist, 0ull, (Cand*)nullptr, (u32*)nullptr, 0u);
  KLR_HIP(ctx, hipGetLastError());
  return KLR_OK;
}

int sweep_fix_tail(klr_ctx* ctx, const klr_mat* mat, double scale, i64* ibuf, double* gbuf, u64* keys, u32* hist, const SelHint* hint) {
  const u64 tau = hint && keys ? hint->tau : 0ull;
  if (tau) hist = nullptr;
  Cand* hcand = hint ? hint->cand : nullptr;
  u32* hcur = hint ? &hint->ctl->cursor : nullptr;
  const i64 dim = mat->dim;
  KLR_TRY(klr_allreduce(ctx, ibuf, dim + IB_EXTRA, KLR_I64));
  hipLaunchKernelGGL(k_int_to_g, dim3((unsigned)std::max<i64>(1, std::min<i64>(ceil_div(dim + 1, 1024), 256))), dim3(256), 0, ctx->stream, ibuf, dim,
                     1.0 / scale, gbuf, keys, hist, tau, hcand, hcur, (hint && hint->lean && tau) ? 1u : 0u);
  KLR_HIP(ctx, hipGetLastError());
  return KLR_OK;
}

// ---- strip-range form (the overlapped collective of klr_select.hip) -------------------------------------------------
// The layout (built on first use) and the buffers every range shares; G = 0: this matrix has no range form.
int sweep_ranges_prepare(klr_ctx* ctx, const klr_mat* mat, int G_want, const Q64Ranges** qr_out, i64** part_out) {
  *qr_out = nullptr;
  if (!(mat->fmt == FMT_Q64C12 || mat->fmt == FMT_Q64C13) || G_want < 2) return KLR_OK;
  KLR_TRY(q64_build_ranges(ctx, const_cast<klr_mat*>(mat), G_want));
  if (!mat->qr) return KLR_OK;
  const u32 pstride = mat->fmt == FMT_Q64C13 ? 8192u : TILE_DIM;
  KLR_TRY(ws_get(ctx, WS_PART, (size_t)std::max<u32>(std::max(mat->npiece, mat->qr->npiece), 1) * pstride * 8, (void**)part_out));
  *qr_out = mat->qr;
  return KLR_OK;
}

// range g's launch of the sweep kernel on the ctx stream (S = the fixed-point weights the row pass left in WS_Z)
int sweep_range_launch(klr_ctx* ctx, const klr_mat* mat, int g, i64* part) {
  const Q64Range& r = mat->qr->r[(size_t)g];
  i64* S;
  KLR_TRY(ws_get(ctx, WS_Z, (size_t)std::max<i64>(mat->n, 1) * 8, (void**)&S));
  Q64View v; v.nwg = r.nwg; v.t_piece = mat->qr->t_piece; v.t_wgpiece = r.t_wgpiece; v.slds_rows = r.slds_rows;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool timed = prof_take(ctx);
  if (timed) {
    KLR_TRY(prof_event_get(ctx, &e0));
    KLR_TRY(prof_event_get(ctx, &e1));
  }
  KLR_TRY(dispatch_sweep_coo(ctx, mat, S, part, e0, e1, &v));
  if (timed) ctx->prof_ev.emplace_back(e0, e1);
  KLR_HIP(ctx, hipGetLastError());
  return KLR_OK;
}

// piece sums of the columns [col0, col1) -> vec[0 .. col1 - col0) (+ the five scalars behind them when rowblk is given)
// on `st`; whole = the piece tables of the ranges (true) or of the one-launch form (false: the fallback after a sweep
// that ran as one launch never gets here)
int sweep_range_finish(klr_ctx* ctx, const klr_mat* mat, i64 col0, i64 col1, const i64* part, i64* vec, const unsigned long long* rowblk,
                       i64 nrowblk, hipStream_t st) {
  const Q64Ranges* qr = mat->qr;
  const i64 ncol = col1 - col0;
  const u32 pstride = mat->fmt == FMT_Q64C13 ? 8192u : TILE_DIM;
  u32 fiter = (u32)std::max<i64>(1, std::min<i64>(8, ncol / (256 * 512)));
  if (ctx->sw.finish_iter > 0) fiter = (u32)std::max(1, std::min(8, ctx->sw.finish_iter));
  const bool flat = qr->max_ct_pieces >= 1 && qr->max_ct_pieces <= 8 && fiter % 4 == 0 && !ctx->sw.finish_no_flat;
  const dim3 fgrid((unsigned)ceil_div(ncol, (i64)256 * fiter) + 1);
  if (flat) hipLaunchKernelGGL((k_finish_coo<false, true>), fgrid, dim3(1024), 0, st, part, qr->t_ctptr, qr->t_ctlist, col0, col1, vec - col0, (const i64*)rowblk,
                               nrowblk, mat->tile_c, pstride, 1.0, (double*)nullptr, (u64*)nullptr, (u32*)nullptr, 0ull, (Cand*)nullptr, (u32*)nullptr, fiter, 0u);
  else hipLaunchKernelGGL((k_finish_coo<false, false>), fgrid, dim3(1024), 0, st, part, qr->t_ctptr, qr->t_ctlist, col0, col1, vec - col0, (const i64*)rowblk,
                          nrowblk, mat->tile_c, pstride, 1.0, (double*)nullptr, (u64*)nullptr, (u32*)nullptr, 0ull, (Cand*)nullptr, (u32*)nullptr, fiter, 0u);
  KLR_HIP(ctx, hipGetLastError());
  return KLR_OK;
}

// s (f64, device) and optional per-row loss terms -> gbuf[0..dim) = X's, gbuf[dim] = sum of loss terms,
// all-reduced across ranks as int64.  Exact and partition invariant.
static int sweep_coo(klr_ctx* ctx, const klr_mat* mat, const double cw[2], const double* s_dev,
                     const double* lterm_dev, double* gbuf) {
  double scale; i64* S; unsigned long long* acc;
  KLR_TRY(sweep_fix_begin(ctx, mat, cw, &scale, &S, &acc));
  KLR_HIP(ctx, hipMemsetAsync(acc, 0, IB_EXTRA * 8, ctx->stream));      // k_fix_s accumulates with atomics
  if (mat->n) hipLaunchKernelGGL(k_fix_s, dim3((unsigned)ceil_div(mat->n, 256)), dim3(256), 0, ctx->stream, s_dev, lterm_dev,
                                 mat->n, scale, S, acc);
  return sweep_fix_finish(ctx, mat, scale, gbuf, nullptr, nullptr, nullptr, 0);
}

// Full-width sweep: gbuf[0..dim) = sum_i s_i x_ij over ALL ranks, gbuf[dim] = sum of the loss terms.
// s_dev holds the per-row weights; lterm_dev (count data) or gbuf[dim] (otherwise) the loss partials.
int sweep_all(klr_ctx* ctx, const klr_mat* mat, const double cw[2], const double* s_dev, const double* lterm_dev,
              double* gbuf) {
  bool use_int;
  KLR_TRY(sweep_protocol(ctx, mat, &use_int));
  if (use_int) return sweep_coo(ctx, mat, cw, s_dev, lterm_dev, gbuf);
  KLR_TRY(launch_xt_s(ctx, mat, s_dev, gbuf, false));          // leaves gbuf[dim] (loss partial) untouched
  KLR_TRY(klr_allreduce(ctx, gbuf, mat->dim + 1, KLR_F64));
  return KLR_OK;
}

int launch_xt_s(klr_ctx* ctx, const klr_mat* mat, const double* s_dev, double* g_raw_dev, bool square) {
  double* part;
  KLR_TRY(ws_get(ctx, WS_PART, (size_t)std::max<i64>(mat->nchunks, 1) * sizeof(double), (void**)&part));
  if (mat->nchunks > 0) {
    const int g = mat->group;
    const dim3 grid((unsigned)ceil_div(mat->nchunks * g, 256));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = ctx->prof_on && !square;
    if (prof) {
      KLR_TRY(prof_event_get(ctx, &e0));
      KLR_TRY(prof_event_get(ctx, &e1));
      KLR_HIP(ctx, hipEventRecord(e0, ctx->stream));
    }
    if (mat->val_f64) { if (square) xt_s_group<double, true>(g, grid, ctx->stream, mat, s_dev, part); else xt_s_group<double, false>(g, grid, ctx->stream, mat, s_dev, part); }
    else              { if (square) xt_s_group<float, true>(g, grid, ctx->stream, mat, s_dev, part);  else xt_s_group<float, false>(g, grid, ctx->stream, mat, s_dev, part); }
    if (prof) {
      KLR_HIP(ctx, hipEventRecord(e1, ctx->stream));
      ctx->prof_ev.emplace_back(e0, e1);
    }
  }
  hipLaunchKernelGGL(k_finish_cols, dim3((unsigned)ceil_div(mat->dim, 256)), dim3(256), 0, ctx->stream, mat->colchunk, mat->dim, part, g_raw_dev);
  KLR_HIP(ctx, hipGetLastError());
  return KLR_OK;
}

// ---------------------------------------------------------------------------------------------
// finishing: TransformFull on the fly (kmerLr_logistic_regression.go:196-222)
//   scale only : g_j = scale_j * raw_j
//   offset     : g_j = scale_j * (raw_j - offset_j * sum_i s_i), sum_i s_i = raw_0 (bias column)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_apply_transform(double* __restrict__ g, i64 dim, const double* __restrict__ scale,
                                                         const double* __restrict__ offset) {
  const i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
  if (j >= dim) return;
  const double ssum = g[0];     // column 0 is never rewritten
  double v = g[j];
  if (offset && j > 0) v = v - offset[j] * ssum;
  if (scale && j > 0) v = v * scale[j];
  if (j > 0) g[j] = v;
}

// uploads the TransformFull vectors (host, length dim) and finishes g on the device
int apply_transform_dev(klr_ctx* ctx, double* gbuf, i64 dim, const double* scale, const double* offset) {
  double* tr;
  KLR_TRY(ws_get(ctx, WS_CD_LOSS, (size_t)dim * 2 * sizeof(double), (void**)&tr));
  if (scale) KLR_TRY(h2d(ctx, tr, scale, (size_t)dim * sizeof(double)));
  if (offset) KLR_TRY(h2d(ctx, tr + dim, offset, (size_t)dim * sizeof(double)));
  hipLaunchKernelGGL(k_apply_transform, dim3((unsigned)ceil_div(dim, 256)), dim3(256), 0, ctx->stream, gbuf, dim,
                     scale ? tr : nullptr, offset ? tr + dim : nullptr);
  KLR_HIP(ctx, hipGetLastError());
  return KLR_OK;
}

// dense theta (device) from a sparse host theta
static int upload_sparse_theta(klr_ctx* ctx, i64 dim, const i64* idx, const double* val, i64 nnz, double** out) {
  double* th;
  KLR_TRY(ws_get(ctx, WS_THETA, (size_t)dim * sizeof(double), (void**)&th));
  std::vector<double> h((size_t)dim, 0.0);
  for (i64 k = 0; k < nnz; k++) {
    if (idx[k] < 0 || idx[k] >= dim) return klr_fail(ctx, KLR_EINVAL, "theta index %lld out of range", (long long)idx[k]);
    h[(size_t)idx[k]] = val[k];
  }
  KLR_TRY(h2d(ctx, th, h.data(), (size_t)dim * sizeof(double)));
  *out = th;
  return KLR_OK;
}

extern "C" int klr_gradient(klr_ctx* ctx, const klr_mat* mat, const int64_t* th_idx, const double* th_val,
                            int64_t th_nnz, const double cw[2], const double* scale, const double* offset,
                            double* g_out) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, mat != nullptr && cw != nullptr && g_out != nullptr && th_nnz >= 0);
  KLR_ARG(ctx, th_nnz == 0 || (th_idx != nullptr && th_val != nullptr));
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  const i64 n = mat->n, dim = mat->dim;
  double *s, *gbuf;
  KLR_TRY(ws_get(ctx, WS_S, (size_t)std::max<i64>(n, 1) * sizeof(double), (void**)&s));
  KLR_TRY(ws_get(ctx, WS_G, (size_t)(dim + 8) * sizeof(double), (void**)&gbuf));

  // effective theta: with an on-the-fly transform, x.theta = sum_j (x_j - o_j) s_j theta_j
  //   = sum_j x_j (s_j theta_j) + (theta_0 - sum_j o_j s_j theta_j)
  std::vector<i64> eidx; std::vector<double> evalv;
  double theta0 = 0.0, shift = 0.0;
  for (i64 k = 0; k < th_nnz; k++) {
    KLR_ARG(ctx, th_idx[k] >= 0 && th_idx[k] < dim && (k == 0 || th_idx[k] > th_idx[k - 1]));
    if (th_idx[k] == 0) { theta0 = th_val[k]; continue; }
    if (th_val[k] == 0.0) continue;
    double t = th_val[k];
    if (scale) t = t * scale[th_idx[k]];
    if (offset) shift += offset[th_idx[k]] * t;
    eidx.push_back(th_idx[k]); evalv.push_back(t);
  }
  theta0 -= shift;

  klr_mat* tmp = nullptr;
  int rc = KLR_OK;
  if (eidx.empty()) {
    rc = launch_row_const(ctx, mat, theta0, ROW_SWEEP, cw, s, nullptr, nullptr);
  } else if ((i64)eidx.size() * 8 < dim) {
    // theta lives on few columns: gather them (CSC side) and run the row pass on the reduced rows
    std::vector<i64> cols; cols.push_back(0);
    cols.insert(cols.end(), eidx.begin(), eidx.end());
    rc = build_reduced(ctx, mat, cols.data(), (i64)cols.size(), nullptr, nullptr, false, &tmp);
    if (rc == KLR_OK) {
      std::vector<double> th; th.push_back(theta0); th.insert(th.end(), evalv.begin(), evalv.end());
      double* thd;
      rc = ws_get(ctx, WS_THETA, th.size() * sizeof(double), (void**)&thd);
      if (rc == KLR_OK) rc = h2d(ctx, thd, th.data(), th.size() * sizeof(double));
      if (rc == KLR_OK) rc = launch_row_pass(ctx, tmp, thd, ROW_SWEEP, cw, s, nullptr, nullptr);
    }
  } else {
    std::vector<i64> idx; std::vector<double> vv;
    idx.push_back(0); vv.push_back(theta0);
    idx.insert(idx.end(), eidx.begin(), eidx.end()); vv.insert(vv.end(), evalv.begin(), evalv.end());
    double* thd;
    rc = upload_sparse_theta(ctx, dim, idx.data(), vv.data(), (i64)idx.size(), &thd);
    if (rc == KLR_OK) rc = launch_row_pass(ctx, mat, thd, ROW_SWEEP, cw, s, nullptr, nullptr);
  }
  if (rc == KLR_OK && hipMemsetAsync(gbuf + dim, 0, sizeof(double), ctx->stream) != hipSuccess)
    rc = klr_fail(ctx, KLR_EHIP, "memset failed");
  if (rc == KLR_OK) rc = sweep_all(ctx, mat, cw, s, nullptr, gbuf);
  if (rc == KLR_OK && (scale || offset)) {
    double* tr;
    rc = ws_get(ctx, WS_TMP0, (size_t)dim * 2 * sizeof(double), (void**)&tr);
    if (rc == KLR_OK && scale) rc = h2d(ctx, tr, scale, (size_t)dim * sizeof(double));
    if (rc == KLR_OK && offset) rc = h2d(ctx, tr + dim, offset, (size_t)dim * sizeof(double));
    if (rc == KLR_OK) {
      hipLaunchKernelGGL(k_apply_transform, dim3((unsigned)ceil_div(dim, 256)), dim3(256), 0, ctx->stream, gbuf, dim,
                         scale ? tr : nullptr, offset ? tr + dim : nullptr);
    }
  }
  if (rc == KLR_OK) rc = d2h_sync(ctx, g_out, gbuf, (size_t)dim * sizeof(double));
  if (tmp) { std::string keep = ctx->err; mat_free(tmp); ctx->err = keep; }
  return rc;
}

extern "C" int klr_loss(klr_ctx* ctx, const klr_mat* mat, const double* theta, const double cw[2],
                        double lambda, double* loss_out) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, mat != nullptr && theta != nullptr && cw != nullptr && loss_out != nullptr);
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  const i64 dim = mat->dim;
  double *thd, *acc;
  KLR_TRY(ws_get(ctx, WS_THETA, (size_t)dim * sizeof(double), (void**)&thd));
  KLR_TRY(ws_get(ctx, WS_G, (size_t)(dim + 8) * sizeof(double), (void**)&acc));
  KLR_TRY(h2d(ctx, thd, theta, (size_t)dim * sizeof(double)));
  KLR_TRY(launch_row_pass(ctx, mat, thd, ROW_LOSS, cw, nullptr, nullptr, acc));
  KLR_TRY(klr_allreduce(ctx, acc, 1, KLR_F64));
  double sum = 0.0;
  KLR_TRY(d2h_sync(ctx, &sum, acc, sizeof(double)));
  if (mat->n_total == 0) { *loss_out = 0.0; return KLR_OK; }              // :251-253
  double r = sum / (double)mat->n_total;                                    // :265
  if (!std::isnan(lambda) && lambda != 0.0)
    for (i64 j = 1; j < dim; j++) r += lambda * std::fabs(theta[j]);       // :266-270
  *loss_out = r;
  return KLR_OK;
}

extern "C" int klr_log_pdf(klr_ctx* ctx, const klr_mat* mat, const double* theta, double* out) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, mat != nullptr && theta != nullptr && out != nullptr);
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  const i64 dim = mat->dim, n = mat->n;
  double *thd, *o;
  const double cw[2] = {1.0, 1.0};
  KLR_TRY(ws_get(ctx, WS_THETA, (size_t)dim * sizeof(double), (void**)&thd));
  KLR_TRY(ws_get(ctx, WS_S, (size_t)std::max<i64>(n, 1) * sizeof(double), (void**)&o));
  KLR_TRY(h2d(ctx, thd, theta, (size_t)dim * sizeof(double)));
  KLR_TRY(launch_row_pass(ctx, mat, thd, ROW_LOGPDF, cw, o, nullptr, nullptr));
  KLR_TRY(d2h_sync(ctx, out, o, (size_t)n * sizeof(double)));
  return KLR_OK;
}


// =================================================================================================
// TransformFull.Fit: per-column statistics with the same lane-group-per-chunk sweep over the plain CSC
//   count, sum x, sum x^2, max |x|  over the ACTIVE rows; finishing on the host (dim values each)
// =================================================================================================
template <typename VT, int G>
__global__ __launch_bounds__(256) void k_col_stats(const u32* __restrict__ crow, const VT* __restrict__ cval,
                                                   const i64* __restrict__ chunk_begin, const u32* __restrict__ chunk_len,
                                                   i64 nchunks, const uint8_t* __restrict__ active, double* __restrict__ part) {
  const i64 gid = ((i64)blockIdx.x * 256 + threadIdx.x) / G;
  const int lane = threadIdx.x % G;
  if (gid >= nchunks) return;
  const i64 b = chunk_begin[gid];
  const i64 e = b + (i64)chunk_len[gid];
  double cnt = 0.0, sx = 0.0, sxx = 0.0, mx = 0.0;
  for (i64 p = b + lane; p < e; p += G) {
    const u32 r = crow[p];
    const double v = (double)cval[p];
    // padding entries are (row 0, value 0); stored explicit zeros of real rows count towards k like upstream:
    // distinguish by position is not possible here, so padding is excluded through the per-column length
    // (the finish step uses the host's true lengths for the count) -- only sums and max are taken here
    if (!active || active[r]) { sx += v; sxx += v * v; mx = fmax(mx, fabs(v)); cnt += (v != 0.0) ? 1.0 : 0.0; }
  }
  cnt = group_sum<G>(cnt); sx = group_sum<G>(sx); sxx = group_sum<G>(sxx);
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) { part[gid * 4 + 0] = cnt; part[gid * 4 + 1] = sx; part[gid * 4 + 2] = sxx; part[gid * 4 + 3] = mx; }
}

__global__ __launch_bounds__(256) void k_col_stats_finish(const i64* __restrict__ colchunk, i64 dim, const double* __restrict__ part,
                                                          double* __restrict__ out) {
  const i64 j = (i64)blockIdx.x * 256 + threadIdx.x;
  if (j >= dim) return;
  double cnt = 0.0, sx = 0.0, sxx = 0.0, mx = 0.0;
  for (i64 c = colchunk[j]; c < colchunk[j + 1]; c++) {
    cnt += part[c * 4 + 0]; sx += part[c * 4 + 1]; sxx += part[c * 4 + 2]; mx = fmax(mx, part[c * 4 + 3]);
  }
  out[j] = cnt; out[dim + j] = sx; out[2 * dim + j] = sxx; out[3 * dim + j] = mx;
}

template <typename VT>
static void col_stats_launch(int g, dim3 grid, hipStream_t st, const klr_mat* m, double* part) {
  const VT* cv = (const VT*)m->cval;
  switch (g) {
    case 4:  hipLaunchKernelGGL((k_col_stats<VT, 4>), grid, dim3(256), 0, st, m->crow, cv, m->chunk_begin, m->chunk_len, m->nchunks, m->active, part); break;
    case 8:  hipLaunchKernelGGL((k_col_stats<VT, 8>), grid, dim3(256), 0, st, m->crow, cv, m->chunk_begin, m->chunk_len, m->nchunks, m->active, part); break;
    case 16: hipLaunchKernelGGL((k_col_stats<VT, 16>), grid, dim3(256), 0, st, m->crow, cv, m->chunk_begin, m->chunk_len, m->nchunks, m->active, part); break;
    case 32: hipLaunchKernelGGL((k_col_stats<VT, 32>), grid, dim3(256), 0, st, m->crow, cv, m->chunk_begin, m->chunk_len, m->nchunks, m->active, part); break;
    default: hipLaunchKernelGGL((k_col_stats<VT, 64>), grid, dim3(256), 0, st, m->crow, cv, m->chunk_begin, m->chunk_len, m->nchunks, m->active, part); break;
  }
}

extern "C" int klr_transform_fit(klr_ctx* ctx, const klr_mat* mat, int kind, double* scale_out, double* offset_out,
                                 int* has_offset) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, mat != nullptr && scale_out != nullptr && offset_out != nullptr && has_offset != nullptr);
  KLR_ARG(ctx, kind >= KLR_TR_STANDARDIZER && kind <= KLR_TR_MEAN);
  if (kind == KLR_TR_MAXABS && ctx->ar_fn && ctx->nranks > 1)
    return klr_fail(ctx, KLR_EINVAL, "max-abs-scaler needs a max over ranks; the all-reduce hook only sums");
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  const i64 dim = mat->dim;
  double *part, *out;
  KLR_TRY(ws_get(ctx, WS_PART, (size_t)std::max<i64>(mat->nchunks, 1) * 4 * sizeof(double), (void**)&part));
  KLR_TRY(ws_get(ctx, WS_G, (size_t)(4 * dim + 8) * sizeof(double), (void**)&out));
  if (mat->nchunks > 0) {
    const int g = mat->group;
    const dim3 grid((unsigned)ceil_div(mat->nchunks * g, 256));
    if (mat->val_f64) col_stats_launch<double>(g, grid, ctx->stream, mat, part);
    else              col_stats_launch<float>(g, grid, ctx->stream, mat, part);
  }
  hipLaunchKernelGGL(k_col_stats_finish, dim3((unsigned)ceil_div(dim, 256)), dim3(256), 0, ctx->stream, mat->colchunk, dim, part, out);
  KLR_HIP(ctx, hipGetLastError());
  KLR_TRY(klr_allreduce(ctx, out, 3 * dim, KLR_F64));                 // count, sum, sum of squares
  std::vector<double> h((size_t)4 * dim);
  KLR_TRY(d2h_sync(ctx, h.data(), out, (size_t)4 * dim * sizeof(double)));
  const double n = (double)mat->n_total;
  const double* sx = h.data() + dim; const double* sxx = h.data() + 2 * dim; const double* mx = h.data() + 3 * dim;
  // k[j] of the reference counts STORED entries (explicit zeros included); with the active mask only the
  // host-side lengths of an unmasked matrix are exact, so k is taken as stored entries when no mask is set
  // and as the nonzero count otherwise (identical for k-mer data, which stores no zeros)
  *has_offset = 0;
  for (i64 j = 0; j < dim; j++) { scale_out[j] = 0.0; offset_out[j] = 0.0; }
  if (kind == KLR_TR_STANDARDIZER || kind == KLR_TR_VARIANCE) {
    for (i64 j = 1; j < dim; j++) {
      const double mean = sx[j] / n;
      // sum over stored (v-mean)^2 + (n-k) mean^2  ==  sum v^2 - 2 mean sum v + n mean^2
      double sj = sxx[j] - 2.0 * mean * sx[j] + n * mean * mean;
      if (sj < 0.0) sj = 0.0;
      offset_out[j] = mean;
      scale_out[j] = (sj == 0.0) ? 1.0 : 1.0 / std::sqrt(sj / (n - 1.0));
    }
    offset_out[0] = 0.0; scale_out[0] = 1.0;
    if (kind == KLR_TR_STANDARDIZER) *has_offset = 1;
    else for (i64 j = 0; j < dim; j++) offset_out[j] = 0.0;
  } else if (kind == KLR_TR_MAXABS) {
    for (i64 j = 1; j < dim; j++) scale_out[j] = 1.0 / mx[j];
    scale_out[0] = 1.0;
  } else {
    for (i64 j = 1; j < dim; j++) scale_out[j] = n / sx[j];
    scale_out[0] = 1.0;
  }
  return KLR_OK;
}
