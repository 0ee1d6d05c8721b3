// klr_ingest.hip -- FASTA sequences -> resident k-mer count matrix, entirely on the device
// (scan_sequences / convert_counts, kmerLr_data.go:197-335, and the CSR -> CSC / tile build of
// klr_mat.hip's host ingest).
//
//   windows   one thread per (sequence, k, position): base-4 code of the window (a=0,c=1,g=2,t=3, case folded),
//             invalid if it holds any other character; class id through a per-k table (identity, or the dense
//             rank of min(code, revcomp(code)) with --revcomp)
//   counts    segmented radix sort of each row's class ids (rocPRIM), run-length encoding -> CSR with the
//             bias entry (0, 1.0) first; --binarize stores 1
//   CSC       stable radix sort of the entries by column (row order inside a column is kept)
//   tiles     stable radix sort by tile id (row-major order inside a tile is kept), counts > 255 split
// Column ids: for k = M..N in turn, classes in increasing code order (same convention as the CLI's host
// counter; gonetics' own class order cannot be verified offline).  All integer work: the result is
// bit-identical to the host ingest of the same counts (tests/test_gpu_ingest.py).
#include <cstring>

#include "klr_internal.h"
#include "klr_kmers.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>

#define INVALID_ID 0xFFFFFFFFu

// ---- windows ------------------------------------------------------------------------------------------
__device__ __forceinline__ int base_code(unsigned char ch) {
  switch (ch | 0x20) { case 'a': return 0; case 'c': return 1; case 'g': return 2; case 't': return 3; default: return -1; }
}

// Candidate k-mers of a window.  Nucleotide alphabet: the window itself.  Gapped alphabet: the window with every subset
// of its inner positions replaced by the gap letter (2^(k-2) candidates); the class table says which of them are
// k-mers of the alphabet (gaps never first or last, at most max-ambiguous of them) -- the others map to INVALID_ID.
// Iupac alphabet (alpha = 2): every inner position carries its nucleotide or one of the 7 ambiguous codes that hold it
// (8^(k-2) candidates, three bits of the variant number per inner position).
__host__ __device__ static inline u32 window_variants(int k, int alpha) {
  return k > 2 ? (alpha == 2 ? 1u << (3 * (k - 2)) : (alpha == 1 ? 1u << (k - 2) : 1u)) : 1u;
}
// the ambiguous iupac codes (letters a c g t r y s w k m b d h v n = 0 .. 14) that hold nucleotide b
__device__ static const unsigned char IUPAC_HOLDS[4][7] = {{4, 7, 9, 11, 12, 13, 14}, {5, 6, 9, 10, 12, 13, 14},
                                                           {4, 6, 8, 10, 11, 13, 14}, {5, 7, 8, 10, 11, 12, 14}};

// ids of row i for k = M..N are laid out k-major: [k=M windows x variants | k=M+1 ... | ...]
__global__ __launch_bounds__(256) void k_row_windows(const i64* __restrict__ seq_off, i64 n, int M, int N, int gapped,
                                                    u32* __restrict__ wcount, u32* __restrict__ overflow) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const i64 len = seq_off[i + 1] - seq_off[i];
  unsigned long long w = 0;
  for (int k = M; k <= N; k++) if (len >= k) w += (unsigned long long)(len - k + 1) * window_variants(k, gapped);
  if (w >= 0xFFFFFFFFull) { atomicOr(overflow, 1u); w = 0; }
  wcount[i] = (u32)w;
}

__global__ __launch_bounds__(256) void k_emit_ids(const unsigned char* __restrict__ seqs, const i64* __restrict__ seq_off,
                                                  const i64* __restrict__ woff, i64 n, int M, int N, int gapped,
                                                  const u32* __restrict__ maps, const unsigned long long* __restrict__ map_off,
                                                  u32* __restrict__ ids) {
  // one block per row; threads stride over the row's (window, variant) pairs
  const i64 i = blockIdx.x;
  const i64 s0 = seq_off[i], len = seq_off[i + 1] - s0;
  const u32 B = gapped == 2 ? 15u : (gapped ? 5u : 4u);
  i64 o = woff[i];
  for (int k = M; k <= N; k++) {
    if (len < k) continue;
    const u32 V = window_variants(k, gapped);
    const i64 nw = (len - k + 1) * (i64)V;
    const u32* map = maps + map_off[k - M];
    for (i64 t = threadIdx.x; t < nw; t += 256) {
      const i64 p = t / V;
      const u32 mask = (u32)(t - p * V);                 // bit q - 1 set: inner position q carries the gap letter
      u32 code = 0; bool ok = true;
      for (int q = 0; q < k; q++) {
        const unsigned char ch = seqs[s0 + p + q];
        int b = base_code(ch);
        if (gapped == 2) {                                 // iupac: three bits per inner position, 0 = the nucleotide itself
          const u32 sel = q > 0 && q < k - 1 ? (mask >> (3 * (q - 1))) & 7u : 0u;
          ok = ok && b >= 0;
          if (b >= 0 && sel) b = IUPAC_HOLDS[b][sel - 1];
          code = code * B + (u32)(b < 0 ? 0 : b);
          continue;
        }
        const bool gap = q > 0 && q < k - 1 && ((mask >> (q - 1)) & 1u);
        if (gapped && (ch | 0x20) == 'n') { b = 4; ok = ok && gap; }     // a gap in the sequence only matches a gap
        else if (gap) b = 4;
        ok = ok && b >= 0;
        code = code * B + (u32)(b < 0 ? 0 : b);
      }
      ids[o + t] = ok ? map[code] : INVALID_ID;
    }
    o += nw;
  }
}

// ---- run-length encoding of the sorted ids ------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mark_heads(const u32* __restrict__ ids, const i64* __restrict__ woff, i64 n,
                                                    u32* __restrict__ head, u32* __restrict__ rowheads) {
  // one block per row: head flags + number of distinct valid ids
  const i64 i = blockIdx.x;
  const i64 b = woff[i], e = woff[i + 1];
  u32 cnt = 0;
  for (i64 p = b + threadIdx.x; p < e; p += 256) {
    const u32 v = ids[p];
    const u32 h = (v != INVALID_ID && (p == b || ids[p - 1] != v)) ? 1u : 0u;
    head[p] = h;
    cnt += h;
  }
  __shared__ u32 sh[4];
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) rowheads[i] = sh[0] + sh[1] + sh[2] + sh[3] + 1u;      // + the bias entry
}

__global__ __launch_bounds__(256) void k_write_csr(const u32* __restrict__ ids, const u32* __restrict__ head,
                                                   const i64* __restrict__ woff, const i64* __restrict__ rowptr, i64 n, int binarize,
                                                   u32* __restrict__ col, float* __restrict__ val, u32* __restrict__ erow) {
  // one block per row; entries are few per thread: a head at p owns the run [p, next head or first invalid)
  const i64 i = blockIdx.x;
  const i64 b = woff[i], e = woff[i + 1];
  const i64 r0 = rowptr[i];
  __shared__ u32 base;
  if (threadIdx.x == 0) { col[r0] = 0u; val[r0] = 1.0f; erow[r0] = (u32)i; base = 1u; }
  __syncthreads();
  // process the row in tiles of 256 positions, ranks by ballot prefix
  for (i64 p0 = b; p0 < e; p0 += 256) {
    const i64 p = p0 + threadIdx.x;
    const bool h = p < e && head[p] != 0u;
    const unsigned long long m = __ballot(h);
    __shared__ u32 wcnt[4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) wcnt[w] = (u32)__popcll(m);
    __syncthreads();
    u32 off = base;
    for (int q = 0; q < w; q++) off += wcnt[q];
    if (h) {
      const u32 slot = off + (u32)__popcll(m & ((1ull << lane) - 1ull));
      const u32 v = ids[p];
      i64 q = p + 1;
      while (q < e && ids[q] == v) q++;                    // run length (runs are short: k-mer multiplicities)
      col[r0 + slot] = v + 1u;
      val[r0 + slot] = binarize ? 1.0f : (float)(q - p);
      erow[r0 + slot] = (u32)i;
    }
    __syncthreads();
    if (threadIdx.x == 0) base += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
}

// ---- CSC --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_col_hist(const u32* __restrict__ col, const float* __restrict__ val, i64 nnz,
                                                  u32* __restrict__ collen, unsigned long long* __restrict__ colsum) {
  const i64 a = (i64)blockIdx.x * 256 + threadIdx.x;
  if (a >= nnz) return;
  atomicAdd(&collen[col[a]], 1u);
  atomicAdd(&colsum[col[a]], (unsigned long long)val[a]);
}

__global__ __launch_bounds__(256) void k_iota(u32* __restrict__ v, i64 n) {
  const i64 a = (i64)blockIdx.x * 256 + threadIdx.x;
  if (a < n) v[a] = (u32)a;
}

__global__ __launch_bounds__(256) void k_csc_scatter(const u32* __restrict__ sorted_col, const u32* __restrict__ sorted_idx, i64 nnz,
                                                     const i64* __restrict__ colstart, const i64* __restrict__ colptr_pad,
                                                     const u32* __restrict__ erow, const float* __restrict__ val,
                                                     u32* __restrict__ crow, float* __restrict__ cval) {
  const i64 k = (i64)blockIdx.x * 256 + threadIdx.x;
  if (k >= nnz) return;
  const u32 c = sorted_col[k];
  const u32 idx = sorted_idx[k];
  const i64 pos = colptr_pad[c] + (k - colstart[c]);
  crow[pos] = erow[idx];
  cval[pos] = val[idx];
}

// ---- tiles -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tile_keys(const u32* __restrict__ col, const float* __restrict__ val, const u32* __restrict__ erow,
                                                   i64 nnz, u32 nrt, u32 nct, u32 RT, u32 CT, int rt_major, u32 ntile, int u24,
                                                   u32* __restrict__ tkey, u32* __restrict__ ne, u32* __restrict__ tile_ne) {
  const i64 a = (i64)blockIdx.x * 256 + threadIdx.x;
  if (a >= nnz) return;
  const u32 c = col[a];
  if (c == 0u) { tkey[a] = ntile; ne[a] = 0u; return; }          // bias column is not stored: sorts to the end
  const u32 rt = erow[a] / RT, ct = c / CT;
  const u32 t = rt_major ? rt * nct + ct : ct * nrt + rt;
  const u32 v = (u32)val[a];
  const u32 k = u24 ? v : (v + 254u) / 255u;          // unit entries / 8-bit counts
  tkey[a] = t; ne[a] = k;
  atomicAdd(&tile_ne[t], k);
}

__global__ __launch_bounds__(256) void k_tile_quads(const u32* __restrict__ tile_ne, u32 ntile, u32* __restrict__ tile_q4) {
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  if (t < ntile) tile_q4[t] = (tile_ne[t] + 3u) / 4u;
}

__global__ __launch_bounds__(256) void k_gather_u32(const u32* __restrict__ src, const u32* __restrict__ idx, i64 n, u32* __restrict__ dst) {
  const i64 a = (i64)blockIdx.x * 256 + threadIdx.x;
  if (a < n) dst[a] = src[idx[a]];
}

__global__ __launch_bounds__(256) void k_tile_scatter(const u32* __restrict__ skey, const u32* __restrict__ sidx, i64 nnz, u32 ntile,
                                                      const i64* __restrict__ ne_prefix /* over sorted order */,
                                                      const i64* __restrict__ tile_first /* exscan of per-tile ne */,
                                                      const i64* __restrict__ tile_q /* exscan of quads */,
                                                      const u32* __restrict__ col, const float* __restrict__ val,
                                                      const u32* __restrict__ erow, u32 RT, u32 CT, int u24, u32* __restrict__ ent) {
  const i64 k = (i64)blockIdx.x * 256 + threadIdx.x;
  if (k >= nnz) return;
  const u32 t = skey[k];
  if (t >= ntile) return;                                           // bias entries
  const u32 a = sidx[k];
  i64 pos = tile_q[t] * 4 + (ne_prefix[k] - tile_first[t]);
  u32 v = (u32)val[a];
  const u32 base = ((col[a] % CT) << TILE_BITS) | (erow[a] % RT);
  if (u24) {
    unsigned char* b = reinterpret_cast<unsigned char*>(ent) + (size_t)pos * 3;
    for (; v > 0u; v--, b += 3) { b[0] = (unsigned char)base; b[1] = (unsigned char)(base >> 8); b[2] = (unsigned char)(base >> 16); }
    return;
  }
  while (v > 0u) { const u32 q = min(v, 255u); ent[pos++] = (q << 24) | base; v -= q; }
}

// sums over the stored (non-bias) values: [0] unit entries (sum of counts), [1] 4-byte entries (counts > 255 split)
__global__ __launch_bounds__(256) void k_entry_totals(const u32* __restrict__ col, const float* __restrict__ val, i64 nnz,
                                                      unsigned long long* __restrict__ out) {
  unsigned long long t1 = 0, t4 = 0;
  for (i64 a = (i64)blockIdx.x * 256 + threadIdx.x; a < nnz; a += (i64)gridDim.x * 256) {
    if (col[a] == 0u) continue;
    const u32 v = (u32)val[a];
    t1 += v; t4 += (v + 254u) / 255u;
  }
  for (int o = 32; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o, 64); t4 += __shfl_xor(t4, o, 64); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], t1); atomicAdd(&out[1], t4); }
}

// ---- r64 quads: one thread per row walks the row's (column-sorted) entries twice -------------------------------------
// pass 1: quads of every (row, column tile) run -> rq[tile * RT + row in tile]
__global__ __launch_bounds__(256) void k_r64_count(const i64* __restrict__ rowptr, const u32* __restrict__ col, const float* __restrict__ val,
                                                   i64 n, u32 RT, u32 CT, u32 nrt, u32 nct, int rt_major, u32* __restrict__ rq) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u32 rt = (u32)(i / RT), ri = (u32)(i - (i64)rt * RT);
  u32 cur_ct = 0xFFFFFFFFu, run = 0;
  for (i64 a = rowptr[i] + 1; a < rowptr[i + 1]; a++) {                    // entry 0 is the bias
    const u32 ct = col[a] / CT;
    if (ct != cur_ct) {
      if (cur_ct != 0xFFFFFFFFu) rq[(size_t)(rt_major ? rt * nct + cur_ct : cur_ct * nrt + rt) * RT + ri] = (run + 3u) / 4u;
      cur_ct = ct; run = 0;
    }
    run += (u32)val[a];
  }
  if (cur_ct != 0xFFFFFFFFu) rq[(size_t)(rt_major ? rt * nct + cur_ct : cur_ct * nrt + rt) * RT + ri] = (run + 3u) / 4u;
}

// quads per tile -> 16-byte load units (two quads)
__global__ __launch_bounds__(256) void k_r64_tile_units(const i64* __restrict__ rq_prefix, u32 RT, u32 ntile, u32* __restrict__ units) {
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  if (t < ntile) units[t] = (u32)((rq_prefix[(size_t)(t + 1) * RT] - rq_prefix[(size_t)t * RT] + 1) / 2);
}

// pass 2: the quads themselves; every quad is written by exactly one thread
__global__ __launch_bounds__(256) void k_r64_fill(const i64* __restrict__ rowptr, const u32* __restrict__ col, const float* __restrict__ val,
                                                  i64 n, u32 RT, u32 CT, u32 nrt, u32 nct, int rt_major, const i64* __restrict__ rq_prefix,
                                                  const i64* __restrict__ tile_unit, unsigned long long* __restrict__ ent) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u32 rt = (u32)(i / RT), ri = (u32)(i - (i64)rt * RT);
  u32 cur_ct = 0xFFFFFFFFu, p = 0;
  i64 qpos = 0;
  unsigned long long w = 0;
  for (i64 a = rowptr[i] + 1; a < rowptr[i + 1]; a++) {
    const u32 c = col[a];
    const u32 ct = c / CT;
    if (ct != cur_ct) {
      if (cur_ct != 0xFFFFFFFFu && (p & 3u)) ent[qpos++] = w;
      const size_t t = (size_t)(rt_major ? rt * nct + ct : ct * nrt + rt);
      qpos = tile_unit[t] * 2 + (rq_prefix[t * RT + ri] - rq_prefix[t * RT]);
      cur_ct = ct; p = 0;
    }
    const unsigned long long cl = (unsigned long long)(c - ct * CT);
    for (u32 v = (u32)val[a]; v > 0u; v--) {
      if ((p & 3u) == 0u) w = r64_null_quad(ri, (u32)qpos);
      const int sh = 12 * (int)(p & 3u);
      w = (w & ~(0xFFFull << sh)) | (cl << sh);
      if ((++p & 3u) == 0u) ent[qpos++] = w;
    }
  }
  if (cur_ct != 0xFFFFFFFFu && (p & 3u)) ent[qpos++] = w;
}

// tiles with an odd number of quads end in a null quad (row 4095: s = 0)
__global__ __launch_bounds__(256) void k_r64_pad(const i64* __restrict__ rq_prefix, const i64* __restrict__ tile_unit, u32 RT, u32 ntile,
                                                 unsigned long long* __restrict__ ent) {
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ntile) return;
  const i64 quads = rq_prefix[(size_t)(t + 1) * RT] - rq_prefix[(size_t)t * RT];
  if (quads & 1) { const i64 qp = tile_unit[t] * 2 + quads; ent[qp] = r64_null_quad(4095u, (u32)qp); }
}

// ---- q64: row-streamed quads (k_sweep_q64): column strips only, rows ascending inside a strip, a quad =
// row:(64 - 4 CB) | 4 x col:CB with the row relative to the first row of its 64-unit block -------------------------------
// quads of both strip widths in one pass over the rows (the ingest picks the fewer bytes): out[0] CB = 12, out[1] CB = 13
__global__ __launch_bounds__(256) void k_q64_totals(const i64* __restrict__ rowptr, const u32* __restrict__ col, const float* __restrict__ val,
                                                    i64 n, u32 CT12, u32 CT13, unsigned long long* __restrict__ out) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  unsigned long long q12 = 0, q13 = 0;
  if (i < n) {
    u32 c12 = 0xFFFFFFFFu, c13 = 0xFFFFFFFFu, r12 = 0, r13 = 0;
    for (i64 a = rowptr[i] + 1; a < rowptr[i + 1]; a++) {                  // entry 0 is the bias
      const u32 c = col[a], v = (u32)val[a];
      const u32 t12 = c / CT12, t13 = c / CT13;
      if (t12 != c12) { q12 += (r12 + 3u) / 4u; c12 = t12; r12 = 0; }
      if (t13 != c13) { q13 += (r13 + 3u) / 4u; c13 = t13; r13 = 0; }
      r12 += v; r13 += v;
    }
    q12 += (r12 + 3u) / 4u; q13 += (r13 + 3u) / 4u;
  }
  for (int o = 32; o > 0; o >>= 1) { q12 += __shfl_xor(q12, o, 64); q13 += __shfl_xor(q13, o, 64); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], q12); atomicAdd(&out[1], q13); }
}

// quads of every (row, strip) run -> rq[strip * n + row]
__global__ __launch_bounds__(256) void k_q64_count(const i64* __restrict__ rowptr, const u32* __restrict__ col, const float* __restrict__ val,
                                                   i64 n, u32 CT, u32* __restrict__ rq) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  u32 cur = 0xFFFFFFFFu, run = 0;
  for (i64 a = rowptr[i] + 1; a < rowptr[i + 1]; a++) {
    const u32 t = col[a] / CT;
    if (t != cur) { if (cur != 0xFFFFFFFFu) rq[(size_t)cur * n + i] = (run + 3u) / 4u; cur = t; run = 0; }
    run += (u32)val[a];
  }
  if (cur != 0xFFFFFFFFu) rq[(size_t)cur * n + i] = (run + 3u) / 4u;
}

// 16-byte units of every strip, rounded up to whole 64-unit blocks (a wavefront never straddles two strips)
__global__ __launch_bounds__(256) void k_q64_strip_units(const i64* __restrict__ rq_prefix, i64 n, u32 nct, u32* __restrict__ units) {
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nct) return;
  const i64 quads = rq_prefix[(size_t)(t + 1) * n] - rq_prefix[(size_t)t * n];
  units[t] = (u32)(((quads + 1) / 2 + 63) / 64 * 64);
}

__device__ __forceinline__ unsigned long long q64_null_quad(int CB, u32 pos) {
  const unsigned long long d = ((1ull << CB) - 64ull) + (pos & 63u);          // row field 0: the block's first row
  return d | (d << CB) | (d << (2 * CB)) | (d << (3 * CB));
}

__global__ __launch_bounds__(256) void k_q64_fill_null(unsigned long long* __restrict__ ent, i64 nquads, int CB) {
  const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
  if (q < nquads) ent[q] = q64_null_quad(CB, (u32)q);
}

// first row of every 64-unit (128-quad) block: the quad at a block's first position names it (rows ascend inside a
// strip, strips are whole blocks); blocks of padding only keep row 0
__global__ __launch_bounds__(256) void k_q64_blkrow(const i64* __restrict__ rowptr, const u32* __restrict__ col, const float* __restrict__ val,
                                                    i64 n, u32 CT, const i64* __restrict__ rq_prefix, const i64* __restrict__ strip_unit,
                                                    u32* __restrict__ blkrow) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  u32 cur = 0xFFFFFFFFu;
  for (i64 a = rowptr[i] + 1; a < rowptr[i + 1]; a++) {
    const u32 t = col[a] / CT;
    if (t == cur) continue;
    cur = t;
    const i64 qs = strip_unit[t] * 2 + (rq_prefix[(size_t)t * n + i] - rq_prefix[(size_t)t * n]);
    const i64 qe = qs + (rq_prefix[(size_t)t * n + i + 1] - rq_prefix[(size_t)t * n + i]);
    for (i64 b = (qs + 127) & ~(i64)127; b < qe; b += 128) blkrow[b >> 7] = (u32)i;
  }
}

// the quads themselves; every quad is written by exactly one thread.  A row further than the row field reaches from its
// block's first row (only possible when thousands of consecutive rows are absent from a strip) raises `overflow`: the
// ingest then falls back to the tiled formats.
__global__ __launch_bounds__(256) void k_q64_fill(const i64* __restrict__ rowptr, const u32* __restrict__ col, const float* __restrict__ val,
                                                  i64 n, u32 CT, int CB, const i64* __restrict__ rq_prefix, const i64* __restrict__ strip_unit,
                                                  const u32* __restrict__ blkrow, unsigned long long* __restrict__ ent, u32* __restrict__ overflow) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int rsh = 4 * CB;
  const u32 rmax = (1u << (64 - rsh)) - 1u;
  u32 cur = 0xFFFFFFFFu, p = 0;
  i64 qpos = 0;
  unsigned long long w = 0;
  auto start_quad = [&]() {
    const u32 rel = (u32)i - blkrow[qpos >> 7];
    if (rel > rmax) atomicOr(overflow, 1u);
    w = q64_null_quad(CB, (u32)qpos) | ((unsigned long long)(rel & rmax) << rsh);
  };
  for (i64 a = rowptr[i] + 1; a < rowptr[i + 1]; a++) {
    const u32 c = col[a];
    const u32 t = c / CT;
    if (t != cur) {
      if (cur != 0xFFFFFFFFu && (p & 3u)) ent[qpos++] = w;
      qpos = strip_unit[t] * 2 + (rq_prefix[(size_t)t * n + i] - rq_prefix[(size_t)t * n]);
      cur = t; p = 0;
    }
    const unsigned long long cl = (unsigned long long)(c - t * CT);
    for (u32 v = (u32)val[a]; v > 0u; v--) {
      if ((p & 3u) == 0u) start_quad();
      const int sh = CB * (int)(p & 3u);
      w = (w & ~(((1ull << CB) - 1ull) << sh)) | (cl << sh);
      if ((++p & 3u) == 0u) ent[qpos++] = w;
    }
  }
  if (cur != 0xFFFFFFFFu && (p & 3u)) ent[qpos++] = w;
}

// Bank-aware column order.  The sweep's k-th LDS atomic of a step adds to column slot k of every lane's unit.  An 8-byte
// LDS atomic is serviced like an 8-byte store: the wave-instruction's 64 addresses go through in four groups of 16
// consecutive lanes, the bank of byte address a is (a / 4) mod 32, so an accumulator covers one of 16 bank pairs,
// (column mod 16); a group takes one LDS cycle per lane on its fullest bank pair.  Columns in ascending order leave
// that to chance (16 lanes on 16 bank pairs: 3.1 on the fullest, ~12.3 cycles per instruction); with at most two lanes
// per bank pair in every group the instruction costs what its address and data transfer costs anyway (8 - 9.6 cycles,
// tools/lds_probe2.hip).  Which slot of its quad a column sits in carries no meaning (all four get the row's S), nor
// which of a unit's two quads when both name the same row: this pass permutes the columns inside every lane's unit so
// that no bank pair is wanted by three or more lanes of a 16-lane group in the same slot.  One wavefront per 64-unit
// block, a parallel local search: count, every lane on an over-subscribed (slot, residue) proposes a swap with one of
// its own slots where both columns find room, half of the lanes (a hash of position and round) act per round, a few
// rounds.  Deterministic; the sums are integers, so the sweep's result does not depend on the order.
__global__ __launch_bounds__(64) void k_q64_bank_order(uint4* __restrict__ ent, int CB, int rounds) {
  __shared__ u32 cnt[4 * 8 * 16];
  const u32 lane = threadIdx.x;
  const size_t pos = (size_t)blockIdx.x * 64 + lane;
  const uint4 e = ent[pos];
  const unsigned long long qa = (unsigned long long)e.x | ((unsigned long long)e.y << 32);
  const unsigned long long qb = (unsigned long long)e.z | ((unsigned long long)e.w << 32);
  const int rsh = 4 * CB;
  const u32 CM = (1u << CB) - 1u;
  u32 c[8];
#pragma unroll
  for (int i = 0; i < 4; i++) { c[i] = (u32)(qa >> (CB * i)) & CM; c[4 + i] = (u32)(qb >> (CB * i)) & CM; }
  const bool same = (qa >> rsh) == (qb >> rsh);
  u32* my = cnt + (lane >> 4) * 128u;
  bool changed = false;
  for (int it = 0; it < rounds; it++) {
    for (u32 i = lane; i < 512u; i += 64u) cnt[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; s++) atomicAdd(&my[s * 16 + (c[s] & 15u)], 1u);
    __syncthreads();
    u32 hsh = (u32)pos * 2654435761u + (u32)it * 0x9E3779B9u;
    hsh ^= hsh >> 15; hsh *= 0x2C1B3C6Du; hsh ^= hsh >> 12;
    const bool act = (hsh & 1u) != 0u;
    bool bad = false, moved = false;
#pragma unroll
    for (int s = 0; s < 8; s++) {
      const u32 r = c[s] & 15u;
      if (my[s * 16 + r] >= 3u) {
        bad = true;
        if (act && !moved) {
#pragma unroll
          for (int s2 = 0; s2 < 8; s2++) {
            if (s2 == s) continue;
            if (!same && (s2 >> 2) != (s >> 2)) continue;
            const u32 r2 = c[s2] & 15u;
            if (!moved && r2 != r && my[s2 * 16 + r] < 2u && my[s * 16 + r2] < 2u) {
              const u32 t = c[s]; c[s] = c[s2]; c[s2] = t;
              moved = true;
            }
          }
        }
      }
    }
    changed |= moved;
    if (!__any((int)bad)) break;
    __syncthreads();
  }
  if (changed) {
    unsigned long long na = (qa >> rsh) << rsh, nb = (qb >> rsh) << rsh;
#pragma unroll
    for (int i = 0; i < 4; i++) { na |= (unsigned long long)c[i] << (CB * i); nb |= (unsigned long long)c[4 + i] << (CB * i); }
    ent[pos] = make_uint4((u32)na, (u32)(na >> 32), (u32)nb, (u32)(nb >> 32));
  }
}

// the stream pre-filled with null unit entries (row 4095, column 0): bytes FF 0F 00 repeating
__global__ __launch_bounds__(256) void k_fill_u24_null(u32* __restrict__ w, i64 nwords) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i >= nwords) return;
  const u32 r = (u32)(i % 3);
  w[i] = r == 0 ? 0xFF000FFFu : (r == 1 ? 0x0FFF000Fu : 0x000FFF00u);
}

// ---- small helpers -----------------------------------------------------------------------------------------
template <typename T>
static int dalloc(klr_ctx* ctx, T** p, size_t count) { return dev_alloc(ctx, (void**)p, std::max<size_t>(count, 1) * sizeof(T)); }

struct TmpPool {
  std::vector<void*> v;
  ~TmpPool() { for (void* p : v) (void)hipFree(p); }
  template <typename T> int get(klr_ctx* ctx, T** p, size_t count) {
    int rc = dalloc(ctx, p, count);
    if (rc == KLR_OK) v.push_back((void*)*p);
    return rc;
  }
};

static int exscan64(klr_ctx* ctx, const u32* in, i64 n, i64* out) { return launch_exscan_u32(ctx, in, n, out); }

int build_pieces_and_upload(klr_ctx* ctx, klr_mat* m, const std::vector<u64>& toff4, i64 nrt, i64 nct, bool rt_major);

// pieces of a q64 stream: the 64-unit blocks of strips [t0, t1) are dealt to nwg workgroups in equal shares, a share
// snapped to a strip boundary when one lies within a few percent of it; a share is cut into pieces where the strip
// changes.  Host work only.  piece: 4 u32 per piece {strip, first unit, end unit, 0}, ids local to the call; wgp: first
// piece of every workgroup (nwg + 1); rec: the 48-byte workgroup records of the kernel (piece ids local: the caller adds
// its offset); slds_rows > 0: every workgroup's rows fit LDS beside its accumulators (SLDS), the most rows of one.
struct Q64Cuts { std::vector<u32> piece, wgp, rec; u32 npiece = 0, slds_rows = 0; };
static void q64_cuts(klr_ctx* ctx, const klr_mat* m, const std::vector<u64>& sb, i64 t0, i64 t1, u32 nwg, int per_cu, Q64Cuts* out) {
  const std::vector<u32>& h_blkrow = m->h_q64_blkrow;
  const i64 n_rows = m->n;
  auto h_blkrow_at = [&](u32 b) -> u32 {
    const i64 r = b < h_blkrow.size() ? (i64)h_blkrow[b] : 0;
    return (u32)std::max<i64>(0, std::min<i64>(r, n_rows - 1024));       // the warming load reads S[r .. r + 1024)
  };
  const u64 blk0 = sb[(size_t)t0], blk1 = sb[(size_t)t1];
  const u64 total_blk = blk1 - blk0;
  // The kernel ends when its slowest workgroup does.  Equal shares of the blocks, with two corrections: a cut within
  // 4 % of a share of a strip boundary is snapped to it (a piece is 2^CB x 8 bytes written and read back: not worth a
  // sliver of a strip), and a share that still crosses a boundary pays for its second piece -- a second vector zeroed
  // and written out, its wavefronts' pipelines started a second time: 1 - 2 us, tools/q64_phases.py -- with
  // KLR_Q64_PIECE_COST blocks (default 40), which all shares carry in proportion (first pass: how many cross).
  // KLR_Q64_CUT=snap leaves the crossing shares whole (A/B).
  std::vector<u64> cut((size_t)nwg + 1);
  {
    const char* pc = ctx_opt(ctx, "KLR_Q64_PIECE_COST");
    const u64 piece_cost = pc ? (u64)std::max(0, atoi(pc)) : 40;
    const char* cs = ctx_opt(ctx, "KLR_Q64_CUT");
    const bool plain = cs && !strcmp(cs, "snap");
    const char* st = ctx_opt(ctx, "KLR_Q64_SNAP_TOL");
    const double tolf = st ? atof(st) : 0.04;
    u32 crossings = 0;
    for (int pass = plain ? 1 : 0; pass < 2; pass++) {
      const double extra = pass && !plain ? (double)piece_cost * (double)crossings / (double)std::max<u64>(total_blk, 1) : 0.0;
      i64 t = t0;
      cut[0] = blk0;
      u32 crossed = 0;
      for (u32 w = 1; w <= nwg; w++) {
        const u64 prev = cut[w - 1];
        const u64 share = (u64)((double)(blk1 - prev) * (1.0 + extra) / (double)(nwg - w + 1));
        const u64 ideal = std::min<u64>(prev + share, blk1);
        const double tol = tolf * (double)share;
        while (t < t1 && sb[(size_t)t + 1] <= ideal) t++;
        u64 c = ideal;
        const u64 lo = sb[(size_t)std::min(t, t1)], hi = t < t1 ? sb[(size_t)t + 1] : blk1;
        if (lo > prev && (double)(ideal - lo) <= tol && (ideal - lo) <= (hi - ideal)) c = lo;
        else if ((double)(hi - ideal) <= tol) c = hi;
        if (lo > prev && c > lo) {
          crossed++;
          if (pass && !plain && c > lo + 16) c -= std::min<u64>(piece_cost, c - lo - 16);
        }
        cut[w] = std::max(c, prev);
      }
      crossings = crossed;
    }
    cut[nwg] = blk1;
  }
  std::vector<u32>& piece = out->piece;
  std::vector<u32>& wgp = out->wgp;
  piece.clear(); wgp.assign((size_t)nwg + 1, 0u);
  {
    i64 t = t0;
    for (u32 w = 0; w < nwg; w++) {
      wgp[w] = (u32)(piece.size() / 4);
      u64 q = cut[w];
      const u64 q1 = cut[w + 1];
      while (q < q1) {
        while (sb[(size_t)t + 1] <= q) t++;
        const u64 e = std::min<u64>(q1, sb[(size_t)t + 1]);
        piece.push_back((u32)t); piece.push_back((u32)(q * 64)); piece.push_back((u32)(e * 64)); piece.push_back(0u);
        q = e;
      }
    }
    wgp[nwg] = (u32)(piece.size() / 4);
  }
  out->npiece = (u32)(piece.size() / 4);
  if (ctx_opt(ctx, "KLR_Q64_DEBUG")) {
    fprintf(stderr, "q64 cuts: strips %lld..%lld, %llu blocks, %u workgroups, %u pieces\n", (long long)t0, (long long)t1, (unsigned long long)total_blk, nwg, out->npiece);
    for (u32 w = 0; w < std::min<u32>(nwg, 10u); w++) {
      fprintf(stderr, "  wg %u:", w);
      for (u32 q = wgp[w]; q < wgp[w + 1]; q++) fprintf(stderr, " [strip %u: blocks %u..%u of %llu..%llu]", piece[(size_t)q * 4], piece[(size_t)q * 4 + 1] / 64,
                                                        piece[(size_t)q * 4 + 2] / 64, (unsigned long long)sb[piece[(size_t)q * 4]], (unsigned long long)sb[(size_t)piece[(size_t)q * 4] + 1]);
      fprintf(stderr, "\n");
    }
  }
  // per workgroup (48 bytes): {first piece, end piece, first unit, end unit of the first piece} {first row of its first
  // block, -, -, -} {first row, rows of its first piece; first row, rows of its second piece}.  The last four let the
  // sweep keep the S values of the workgroup's rows in LDS (SLDS) when EVERY workgroup has at most two pieces and its
  // rows fit beside the two accumulator regions: dense strips (configs[2]: ~6.7k rows per workgroup).
  std::vector<u32>& rec = out->rec;
  rec.assign((size_t)nwg * 12, 0u);
  const int CBm = m->fmt == FMT_Q64C13 ? 13 : 12;
  const size_t lds_rows_cap = (((size_t)159 << 10) - 2 * ((size_t)8 << CBm)) / 8;
  bool slds = per_cu == 1 && !(ctx_opt(ctx, "KLR_Q64_SLDS") && atoi(ctx_opt(ctx, "KLR_Q64_SLDS")) == 0);
  u32 most_rows = 0;
  for (u32 w = 0; w < nwg; w++) {
    rec[(size_t)w * 12] = wgp[w]; rec[(size_t)w * 12 + 1] = wgp[w + 1];
    if (wgp[w] < wgp[w + 1]) {
      rec[(size_t)w * 12 + 2] = piece[(size_t)wgp[w] * 4 + 1]; rec[(size_t)w * 12 + 3] = piece[(size_t)wgp[w] * 4 + 2];
      rec[(size_t)w * 12 + 4] = h_blkrow_at(piece[(size_t)wgp[w] * 4 + 1] / 64);
    }
    if (wgp[w + 1] - wgp[w] > 2) slds = false;
    u32 rows = 0;
    for (u32 q = wgp[w]; q < wgp[w + 1] && q < wgp[w] + 2; q++) {
      const u32 t = piece[(size_t)q * 4], b_first = piece[(size_t)q * 4 + 1] / 64, b_end = piece[(size_t)q * 4 + 2] / 64;
      const i64 lo = (i64)h_blkrow[b_first];
      // rows ascend inside a strip: the piece's last row is at most the first row of the block after it
      const i64 hi = (u64)b_end >= sb[(size_t)t + 1] ? n_rows : std::min<i64>(n_rows, (i64)h_blkrow[b_end] + 1);
      const i64 nr = std::max<i64>(hi - lo, 0);
      if (nr > (i64)lds_rows_cap) { slds = false; break; }
      rec[(size_t)w * 12 + 8 + 2 * (q - wgp[w])] = (u32)lo; rec[(size_t)w * 12 + 9 + 2 * (q - wgp[w])] = (u32)nr;
      rows += (u32)nr;
    }
    if (rows > lds_rows_cap) slds = false;
    most_rows = std::max(most_rows, rows);
  }
  out->slds_rows = slds ? std::max<u32>(most_rows, 1u) : 0u;
}

// pieces grouped by strip for the finishing kernel: ctptr (nct + 1), ctlist (npiece), most pieces of one strip
static u32 q64_group_pieces(const std::vector<u32>& piece, u32 npiece, i64 nct, std::vector<u32>& ctptr, std::vector<u32>& ctlist) {
  ctptr.assign((size_t)nct + 1, 0u); ctlist.assign(std::max<u32>(npiece, 1), 0u);
  for (u32 p = 0; p < npiece; p++) ctptr[(size_t)piece[(size_t)p * 4] + 1]++;
  u32 most = 0;
  for (i64 c = 0; c < nct; c++) most = std::max(most, ctptr[(size_t)c + 1]);
  for (i64 c = 0; c < nct; c++) ctptr[(size_t)c + 1] += ctptr[(size_t)c];
  std::vector<u32> cur(ctptr.begin(), ctptr.end() - 1);
  for (u32 p = 0; p < npiece; p++) ctlist[cur[(size_t)piece[(size_t)p * 4]]++] = p;
  return most;
}

static int q64_pieces_and_upload(klr_ctx* ctx, klr_mat* m, const std::vector<i64>& strip_unit, i64 nct) {
  // first row of a block, read back from the device table (one entry per workgroup; rows + 1024 must stay inside S)
  m->h_q64_blkrow.resize((size_t)(strip_unit[(size_t)nct] / 64));
  KLR_HIP(ctx, hipMemcpy(m->h_q64_blkrow.data(), m->t_blkrow, m->h_q64_blkrow.size() * 4, hipMemcpyDeviceToHost));
  // one workgroup per CU (a piece vector written per (strip, workgroup): half as many as with two), KLR_Q64_WGS=2 for two
  const int per_cu = ctx_opt(ctx, "KLR_Q64_WGS") ? std::max(1, std::min(2, atoi(ctx_opt(ctx, "KLR_Q64_WGS")))) : 1;
  const u32 nwg = (u32)std::max(2, per_cu * ctx->n_cu);
  m->h_q64_sb.resize((size_t)nct + 1);
  for (i64 t = 0; t <= nct; t++) m->h_q64_sb[(size_t)t] = (u64)strip_unit[(size_t)t] / 64;
  Q64Cuts c;
  q64_cuts(ctx, m, m->h_q64_sb, 0, nct, nwg, per_cu, &c);
  const u32 npiece = c.npiece;
  std::vector<u32> ctptr, ctlist;
  m->max_ct_pieces = q64_group_pieces(c.piece, npiece, nct, ctptr, ctlist);
  KLR_TRY(dev_alloc(ctx, (void**)&m->t_piece, (size_t)std::max<u32>(npiece, 1) * 16));
  KLR_TRY(dev_alloc(ctx, (void**)&m->t_wgpiece, (size_t)nwg * 48));
  KLR_TRY(dev_alloc(ctx, (void**)&m->t_ctptr, ((size_t)nct + 1) * 4));
  KLR_TRY(dev_alloc(ctx, (void**)&m->t_ctlist, (size_t)std::max<u32>(npiece, 1) * 4));
  KLR_HIP(ctx, hipMemcpy(m->t_piece, c.piece.data(), (size_t)npiece * 16, hipMemcpyHostToDevice));
  m->q64_slds_rows = c.slds_rows;
  KLR_HIP(ctx, hipMemcpy(m->t_wgpiece, c.rec.data(), (size_t)nwg * 48, hipMemcpyHostToDevice));
  KLR_HIP(ctx, hipMemcpy(m->t_ctptr, ctptr.data(), ((size_t)nct + 1) * 4, hipMemcpyHostToDevice));
  KLR_HIP(ctx, hipMemcpy(m->t_ctlist, ctlist.data(), (size_t)npiece * 4, hipMemcpyHostToDevice));
  m->rt_major = false; m->tiled = true; m->nrt = 1; m->nct = (u32)nct; m->nwg = nwg; m->npiece = npiece;
  m->total4 = strip_unit[(size_t)nct];
  return KLR_OK;
}

void q64_free_ranges(klr_mat* m) {
  if (!m->qr) return;
  for (Q64Range& r : m->qr->r) if (r.t_wgpiece) (void)hipFree(r.t_wgpiece);
  if (m->qr->t_piece) (void)hipFree(m->qr->t_piece);
  if (m->qr->t_ctptr) (void)hipFree(m->qr->t_ctptr);
  if (m->qr->t_ctlist) (void)hipFree(m->qr->t_ctlist);
  delete m->qr;
  m->qr = nullptr;
}

// The strip-range layout: G contiguous ranges of strips with about equal stream length (a range boundary is a strip
// boundary), every range cut for all workgroups by the same rule as the whole stream.  Piece ids run through the ranges.
int q64_build_ranges(klr_ctx* ctx, klr_mat* m, int G) {
  if (m->qr && m->qr->G == G) return KLR_OK;
  q64_free_ranges(m);
  const i64 nct = m->nct;
  if (!(m->fmt == FMT_Q64C12 || m->fmt == FMT_Q64C13) || m->h_q64_sb.size() != (size_t)nct + 1 || G < 2 || nct < 2 * G) return KLR_OK;
  const std::vector<u64>& sb = m->h_q64_sb;
  const int per_cu = (int)std::max<u32>(1, m->nwg / (u32)std::max(1, ctx->n_cu));
  Q64Ranges* qr = new Q64Ranges();
  qr->G = G;
  std::vector<u32> piece_all;
  i64 t_prev = 0;
  int rc = KLR_OK;
  for (int g = 0; g < G && rc == KLR_OK; g++) {
    // the strip boundary nearest to (g + 1) / G of the stream (at least one strip per range)
    i64 t_end = nct;
    if (g + 1 < G) {
      const u64 want = sb[(size_t)nct] * (u64)(g + 1) / (u64)G;
      t_end = t_prev + 1;
      while (t_end < nct - (G - 1 - g) && sb[(size_t)t_end] < want) t_end++;
      if (t_end > t_prev + 1 && want - sb[(size_t)t_end - 1] < sb[(size_t)t_end] - want) t_end--;
    }
    Q64Cuts c;
    q64_cuts(ctx, m, sb, t_prev, t_end, m->nwg, per_cu, &c);
    Q64Range r;
    r.strip0 = (u32)t_prev; r.strip1 = (u32)t_end;
    r.col0 = (i64)t_prev * m->tile_c; r.col1 = std::min<i64>(m->dim, (i64)t_end * m->tile_c);
    r.piece0 = (u32)(piece_all.size() / 4); r.npiece = c.npiece; r.nwg = m->nwg; r.slds_rows = c.slds_rows;
    for (u32 w = 0; w < m->nwg; w++) { c.rec[(size_t)w * 12] += r.piece0; c.rec[(size_t)w * 12 + 1] += r.piece0; }
    piece_all.insert(piece_all.end(), c.piece.begin(), c.piece.end());
    if ((rc = dev_alloc(ctx, (void**)&r.t_wgpiece, (size_t)m->nwg * 48)) == KLR_OK &&
        hipMemcpy(r.t_wgpiece, c.rec.data(), (size_t)m->nwg * 48, hipMemcpyHostToDevice) != hipSuccess)
      rc = klr_fail(ctx, KLR_EHIP, "q64_build_ranges: upload failed");
    qr->r.push_back(r);
    t_prev = t_end;
  }
  m->qr = qr;
  if (rc != KLR_OK) { q64_free_ranges(m); return rc; }
  qr->npiece = (u32)(piece_all.size() / 4);
  std::vector<u32> ctptr, ctlist;
  qr->max_ct_pieces = q64_group_pieces(piece_all, qr->npiece, nct, ctptr, ctlist);
  if ((rc = dev_alloc(ctx, (void**)&qr->t_piece, (size_t)std::max<u32>(qr->npiece, 1) * 16)) != KLR_OK ||
      (rc = dev_alloc(ctx, (void**)&qr->t_ctptr, ((size_t)nct + 1) * 4)) != KLR_OK ||
      (rc = dev_alloc(ctx, (void**)&qr->t_ctlist, (size_t)std::max<u32>(qr->npiece, 1) * 4)) != KLR_OK) { q64_free_ranges(m); return rc; }
  if (hipMemcpy(qr->t_piece, piece_all.data(), (size_t)qr->npiece * 16, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(qr->t_ctptr, ctptr.data(), ((size_t)nct + 1) * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(qr->t_ctlist, ctlist.data(), (size_t)qr->npiece * 4, hipMemcpyHostToDevice) != hipSuccess) {
    q64_free_ranges(m);
    return klr_fail(ctx, KLR_EHIP, "q64_build_ranges: upload failed");
  }
  return KLR_OK;
}

// The row-streamed q64 stream (k_sweep_q64) from the device CSR of a count matrix: strips of 4032 (CB = 12) or 8128
// (CB = 13) columns, whichever gives the fewer bytes (stream + the piece vectors the sweep writes); no sort -- a thread
// per row walks its column-sorted entries.  KLR_ENTRY_FORMAT=q64c12|q64c13 forces a width, r64|u24|u32 the tiled
// formats.  *built = false: the caller goes on with the tiled formats.
static int build_q64(klr_ctx* ctx, klr_mat* m, TmpPool& tmp, bool* built) {
  *built = false;
  const char* ef = ctx_opt(ctx, "KLR_ENTRY_FORMAT");
  if (ef && (!strcmp(ef, "r64") || !strcmp(ef, "u24") || !strcmp(ef, "u32"))) return KLR_OK;
  hipStream_t st = ctx->stream;
  const i64 n = m->n, dim = m->dim, nnz = m->nnz;
  const float* val = (const float*)m->val;
  auto geom = [&](int CB, i64* nct, u32* CT) {
    const i64 W = ((i64)1 << CB) - 64;
    *nct = std::max<i64>(1, ceil_div(dim, W));
    *CT = (u32)std::max<i64>(1, ceil_div(dim, *nct));
    *nct = ceil_div(dim, (i64)*CT);
  };
  i64 nct12, nct13; u32 CT12, CT13;
  geom(12, &nct12, &CT12); geom(13, &nct13, &CT13);
  unsigned long long* d_tot; KLR_TRY(tmp.get(ctx, &d_tot, 2));
  KLR_HIP(ctx, hipMemsetAsync(d_tot, 0, 16, st));
  hipLaunchKernelGGL(k_q64_totals, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, m->rowptr, m->col, val, n, CT12, CT13, d_tot);
  unsigned long long h_tot[2];
  KLR_TRY(d2h_sync(ctx, h_tot, d_tot, 16));
  const u32 nwg = (u32)std::max(2, 2 * ctx->n_cu);
  const double bytes12 = 8.0 * (double)h_tot[0] + (double)(nwg + nct12) * 4096.0 * 8.0;
  const double bytes13 = 8.0 * (double)h_tot[1] + (double)(nwg + nct13) * 8192.0 * 8.0;
  int CB = bytes13 < bytes12 ? 13 : 12;
  if (ef && !strcmp(ef, "q64c12")) CB = 12;
  if (ef && !strcmp(ef, "q64c13")) CB = 13;
  const i64 nct = CB == 13 ? nct13 : nct12;
  const u32 CT = CB == 13 ? CT13 : CT12;
  const u64 quads = CB == 13 ? h_tot[1] : h_tot[0];
  if (quads == 0 || quads / 2 + (u64)nct * 64 >= 0x0FFFFF00ull) return KLR_OK;       // the sweep addresses the stream with 32-bit byte offsets
  if ((u64)nct * (u64)n >= ((u64)1 << 32)) return KLR_OK;                              // the (strip, row) table is scanned with 32-bit counts
  {
    // the (strip, row) count table and its prefix are 12 bytes x strips x rows of scratch: when that does not fit beside
    // the stream itself the tiled formats are built instead (not an error)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && ((size_t)nct * (size_t)n * 12 + quads * 8) > free_b / 2) return KLR_OK;
  }
  u32 *rq, *sunits; i64 *rq_prefix, *strip_unit;
  KLR_TRY(tmp.get(ctx, &rq, (size_t)nct * n + 1)); KLR_TRY(tmp.get(ctx, &rq_prefix, (size_t)nct * n + 2));
  KLR_TRY(tmp.get(ctx, &sunits, (size_t)nct + 1)); KLR_TRY(tmp.get(ctx, &strip_unit, (size_t)nct + 2));
  KLR_HIP(ctx, hipMemsetAsync(rq, 0, ((size_t)nct * n + 1) * 4, st));
  hipLaunchKernelGGL(k_q64_count, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, m->rowptr, m->col, val, n, CT, rq);
  KLR_TRY(exscan64(ctx, rq, nct * n, rq_prefix));
  hipLaunchKernelGGL(k_q64_strip_units, dim3((unsigned)ceil_div(nct, 256)), dim3(256), 0, st, rq_prefix, n, (u32)nct, sunits);
  KLR_TRY(exscan64(ctx, sunits, nct, strip_unit));
  std::vector<i64> h_su((size_t)nct + 1);
  KLR_TRY(d2h_sync(ctx, h_su.data(), strip_unit, (size_t)(nct + 1) * 8));
  const u64 units = (u64)h_su[(size_t)nct];
  if (units == 0 || units >= 0x0FFFFF00ull) return KLR_OK;
  unsigned long long* ent64; u32 *blkrow, *d_flag;
  KLR_TRY(dalloc(ctx, (u32**)&ent64, (size_t)units * 4));
  KLR_TRY(dalloc(ctx, &blkrow, (size_t)(units / 64)));
  KLR_TRY(tmp.get(ctx, &d_flag, 1));
  KLR_HIP(ctx, hipMemsetAsync(d_flag, 0, 4, st));
  KLR_HIP(ctx, hipMemsetAsync(blkrow, 0, (size_t)(units / 64) * 4, st));
  hipLaunchKernelGGL(k_q64_fill_null, dim3((unsigned)ceil_div((i64)units * 2, 256)), dim3(256), 0, st, ent64, (i64)units * 2, CB);
  hipLaunchKernelGGL(k_q64_blkrow, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, m->rowptr, m->col, val, n, CT, rq_prefix, strip_unit, blkrow);
  hipLaunchKernelGGL(k_q64_fill, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, m->rowptr, m->col, val, n, CT, CB, rq_prefix, strip_unit,
                     (const u32*)blkrow, ent64, d_flag);
  {
    const char* bo = ctx_opt(ctx, "KLR_Q64_BANK");             // rounds of the bank-aware column order, 0 = columns stay ascending
    const int rounds = bo ? std::max(0, std::min(64, atoi(bo))) : 16;
    if (rounds > 0) hipLaunchKernelGGL(k_q64_bank_order, dim3((unsigned)(units / 64)), dim3(64), 0, st, (uint4*)ent64, CB, rounds);
  }
  KLR_HIP(ctx, hipGetLastError());
  u32 h_flag = 0;
  KLR_TRY(d2h_sync(ctx, &h_flag, d_flag, 4));
  if (h_flag) {                                          // a row field overflowed: very sparse strips -- tiled formats instead
    (void)hipFree(ent64); (void)hipFree(blkrow);
    return KLR_OK;
  }
  m->t_ent = (u32*)ent64; m->t_blkrow = blkrow;
  m->fmt = CB == 13 ? FMT_Q64C13 : FMT_Q64C12;
  m->tile_r = (u32)std::min<i64>(n, 0xFFFFFFFFll); m->tile_c = CT;
  unsigned long long* d_t1; KLR_TRY(tmp.get(ctx, &d_t1, 2));
  KLR_HIP(ctx, hipMemsetAsync(d_t1, 0, 16, st));
  hipLaunchKernelGGL(k_entry_totals, dim3((unsigned)std::min<i64>(ceil_div(nnz, 256), 2048)), dim3(256), 0, st, m->col, val, nnz, d_t1);
  unsigned long long h_t1[2];
  KLR_TRY(d2h_sync(ctx, h_t1, d_t1, 16));
  m->stored_entries = (i64)h_t1[0];
  KLR_TRY(q64_pieces_and_upload(ctx, m, h_su, nct));
  *built = true;
  return KLR_OK;
}

// CSC + tiles + host metadata from a device CSR (rowptr i64[n+1], col u32, val f32 counts, erow u32 row of entry)
static int finish_from_device_csr(klr_ctx* ctx, klr_mat* m, const u32* erow, TmpPool& tmp) {
  hipStream_t st = ctx->stream;
  const i64 n = m->n, dim = m->dim, nnz = m->nnz;
  const float* val = (const float*)m->val;
  // ---- column lengths / sums -> host metadata
  u32* collen; unsigned long long* colsum;
  KLR_TRY(tmp.get(ctx, &collen, (size_t)dim)); KLR_TRY(tmp.get(ctx, &colsum, (size_t)dim));
  KLR_HIP(ctx, hipMemsetAsync(collen, 0, (size_t)dim * 4, st));
  KLR_HIP(ctx, hipMemsetAsync(colsum, 0, (size_t)dim * 8, st));
  if (nnz) hipLaunchKernelGGL(k_col_hist, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, st, m->col, val, nnz, collen, colsum);
  m->h_collen.assign((size_t)dim, 0);
  std::vector<unsigned long long> h_sum((size_t)dim);
  KLR_HIP(ctx, hipMemcpyAsync(m->h_collen.data(), collen, (size_t)dim * 4, hipMemcpyDeviceToHost, st));
  KLR_TRY(d2h_sync(ctx, h_sum.data(), colsum, (size_t)dim * 8));
  m->h_colnz = m->h_collen;                                    // counts are never stored as zero
  m->h_colptr.assign((size_t)dim + 1, 0);
  std::vector<i64> h_colstart((size_t)dim + 1, 0);
  double amax = 0.0;
  for (i64 j = 0; j < dim; j++) {
    h_colstart[(size_t)j + 1] = h_colstart[(size_t)j] + m->h_collen[(size_t)j];
    m->h_colptr[(size_t)j + 1] = m->h_colptr[(size_t)j] + (i64)((m->h_collen[(size_t)j] + 3u) & ~3u);
    if (j > 0) amax = std::max(amax, (double)h_sum[(size_t)j]);
  }
  m->nnz_pad = m->h_colptr[(size_t)dim];
  m->col_abs_max = amax;
  // ---- CSC: stable sort of the entries by column
  u32 *kin, *kout, *vin, *vout;
  KLR_TRY(tmp.get(ctx, &kout, (size_t)nnz)); KLR_TRY(tmp.get(ctx, &vin, (size_t)nnz)); KLR_TRY(tmp.get(ctx, &vout, (size_t)nnz));
  kin = m->col;
  if (nnz) hipLaunchKernelGGL(k_iota, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, st, vin, nnz);
  int bits = 1; while (((i64)1 << bits) < dim) bits++;
  size_t tsz = 0;
  KLR_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tsz, kin, kout, vin, vout, (size_t)nnz, 0, bits, st));
  void* tstore; KLR_TRY(tmp.get(ctx, (char**)&tstore, tsz));
  if (nnz) KLR_HIP(ctx, rocprim::radix_sort_pairs(tstore, tsz, kin, kout, vin, vout, (size_t)nnz, 0, bits, st));
  i64 *d_colstart, *d_colptr;
  KLR_TRY(tmp.get(ctx, &d_colstart, (size_t)dim + 1)); KLR_TRY(tmp.get(ctx, &d_colptr, (size_t)dim + 1));
  KLR_HIP(ctx, hipMemcpyAsync(d_colstart, h_colstart.data(), (size_t)(dim + 1) * 8, hipMemcpyHostToDevice, st));
  KLR_HIP(ctx, hipMemcpyAsync(d_colptr, m->h_colptr.data(), (size_t)(dim + 1) * 8, hipMemcpyHostToDevice, st));
  KLR_TRY(dalloc(ctx, &m->crow, (size_t)m->nnz_pad));
  KLR_TRY(dev_alloc(ctx, &m->cval, (size_t)std::max<i64>(m->nnz_pad, 1) * 4));
  KLR_HIP(ctx, hipMemsetAsync(m->crow, 0, (size_t)std::max<i64>(m->nnz_pad, 1) * 4, st));
  KLR_HIP(ctx, hipMemsetAsync(m->cval, 0, (size_t)std::max<i64>(m->nnz_pad, 1) * 4, st));
  if (nnz) hipLaunchKernelGGL(k_csc_scatter, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, st, kout, vout, nnz, d_colstart, d_colptr,
                              erow, val, m->crow, (float*)m->cval);
  KLR_HIP(ctx, hipGetLastError());
  KLR_HIP(ctx, hipStreamSynchronize(st));
  KLR_TRY(mat_build_chunks_and_upload(ctx, m));
  // ---- packed integer stream of the full-width sweep
  if (n < 2048 || nnz < (1 << 20)) return KLR_OK;
  {
    bool built = false;
    KLR_TRY(build_q64(ctx, m, tmp, &built));
    if (built) return KLR_OK;
  }
  i64 nrt, nct;
  tile_geometry(n, dim, (u32)std::max(2, 2 * ctx->n_cu), &m->tile_r, &m->tile_c, &nrt, &nct);
  const i64 ntile = nrt * nct;
  if (ntile >= ((i64)1 << 31)) return KLR_OK;
  const char* ord = ctx_opt(ctx, "KLR_TILE_ORDER");
  bool rt_major = (double)nnz / (double)ntile >= 32768.0 && ntile > 2 * (i64)ctx->n_cu;      // one tile per workgroup: column-tile major
  if (ord && ord[0] == 'r') rt_major = true;
  if (ord && ord[0] == 'c') rt_major = false;
  // entry format: 3-byte unit entries when that is fewer bytes, else 4-byte entries with an 8-bit count
  bool u24;
  {
    unsigned long long* d_tot; KLR_TRY(tmp.get(ctx, &d_tot, 2));
    KLR_HIP(ctx, hipMemsetAsync(d_tot, 0, 16, st));
    hipLaunchKernelGGL(k_entry_totals, dim3((unsigned)std::min<i64>(ceil_div(nnz, 256), 2048)), dim3(256), 0, st, m->col, val, nnz, d_tot);
    unsigned long long h_tot[2];
    KLR_TRY(d2h_sync(ctx, h_tot, d_tot, 16));
    u24 = 3ull * h_tot[0] <= 4ull * h_tot[1];
    if (const char* ef = ctx_opt(ctx, "KLR_ENTRY_FORMAT")) { if (!strcmp(ef, "u24")) u24 = true; else if (!strcmp(ef, "u32")) u24 = false; }
    if (h_tot[0] >= (1ull << 31)) u24 = false;        // unit entries only while the stream stays below 6 GiB
    // r64 (8-byte quads row:12 | 4 x col:12, every (row, tile) run padded to 4 columns): needs no sort at all
    const char* ef = ctx_opt(ctx, "KLR_ENTRY_FORMAT");
    const bool forced_other = ef && (!strcmp(ef, "u24") || !strcmp(ef, "u32"));
    if (h_tot[0] < (1ull << 31) && !forced_other && (u64)ntile * m->tile_r < ((u64)1 << 31)) {
      const u32 RT = m->tile_r;
      u32 *rq, *tunits; i64 *rq_prefix, *tile_unit;
      KLR_TRY(tmp.get(ctx, &rq, (size_t)ntile * RT + 1)); KLR_TRY(tmp.get(ctx, &rq_prefix, (size_t)ntile * RT + 2));
      KLR_TRY(tmp.get(ctx, &tunits, (size_t)ntile + 1)); KLR_TRY(tmp.get(ctx, &tile_unit, (size_t)ntile + 2));
      KLR_HIP(ctx, hipMemsetAsync(rq, 0, ((size_t)ntile * RT + 1) * 4, st));
      hipLaunchKernelGGL(k_r64_count, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, m->rowptr, m->col, val, n, RT, m->tile_c, (u32)nrt,
                         (u32)nct, rt_major ? 1 : 0, rq);
      KLR_TRY(exscan64(ctx, rq, (i64)ntile * RT, rq_prefix));
      hipLaunchKernelGGL(k_r64_tile_units, dim3((unsigned)ceil_div(ntile, 256)), dim3(256), 0, st, rq_prefix, RT, (u32)ntile, tunits);
      KLR_TRY(exscan64(ctx, tunits, ntile, tile_unit));
      std::vector<i64> h_tu((size_t)ntile + 1);
      KLR_TRY(d2h_sync(ctx, h_tu.data(), tile_unit, (size_t)(ntile + 1) * 8));
      const u64 units_r = (u64)h_tu[(size_t)ntile];
      const u64 bytes_r = 16 * units_r, bytes_other = u24 ? 12 * ((h_tot[0] + 3) / 4) : 16 * ((h_tot[1] + 3) / 4);   // (per-tile padding of the others ignored)
      const bool force_r = ef && !strcmp(ef, "r64");
      if (units_r > 0 && units_r < 0xFFFFFFF0ull && (bytes_r <= bytes_other || force_r)) {
        KLR_TRY(dalloc(ctx, &m->t_ent, (size_t)units_r * 4));
        unsigned long long* ent64 = reinterpret_cast<unsigned long long*>(m->t_ent);
        hipLaunchKernelGGL(k_r64_fill, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, m->rowptr, m->col, val, n, RT, m->tile_c, (u32)nrt,
                           (u32)nct, rt_major ? 1 : 0, rq_prefix, tile_unit, ent64);
        hipLaunchKernelGGL(k_r64_pad, dim3((unsigned)ceil_div(ntile, 256)), dim3(256), 0, st, rq_prefix, tile_unit, RT, (u32)ntile, ent64);
        KLR_HIP(ctx, hipGetLastError());
        KLR_HIP(ctx, hipStreamSynchronize(st));
        m->fmt = FMT_R64;
        m->stored_entries = (i64)h_tot[0];
        std::vector<u64> toff4((size_t)ntile + 1);
        for (i64 t = 0; t <= ntile; t++) toff4[(size_t)t] = (u64)h_tu[(size_t)t];
        return build_pieces_and_upload(ctx, m, toff4, nrt, nct, rt_major);
      }
    }
  }
  u32 *tkey, *ne, *tile_ne, *tile_q4, *skey, *sne;
  KLR_TRY(tmp.get(ctx, &tkey, (size_t)nnz)); KLR_TRY(tmp.get(ctx, &ne, (size_t)nnz)); KLR_TRY(tmp.get(ctx, &skey, (size_t)nnz));
  KLR_TRY(tmp.get(ctx, &sne, (size_t)nnz));
  KLR_TRY(tmp.get(ctx, &tile_ne, (size_t)ntile + 1)); KLR_TRY(tmp.get(ctx, &tile_q4, (size_t)ntile + 1));
  KLR_HIP(ctx, hipMemsetAsync(tile_ne, 0, (size_t)(ntile + 1) * 4, st));
  hipLaunchKernelGGL(k_tile_keys, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, st, m->col, val, erow, nnz, (u32)nrt, (u32)nct,
                     m->tile_r, m->tile_c, rt_major ? 1 : 0, (u32)ntile, u24 ? 1 : 0, tkey, ne, tile_ne);
  int tbits = 1; while (((i64)1 << tbits) < ntile + 1) tbits++;
  size_t tsz2 = 0;
  KLR_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tsz2, tkey, skey, vin, vout, (size_t)nnz, 0, tbits, st));
  void* tstore2; KLR_TRY(tmp.get(ctx, (char**)&tstore2, tsz2));
  KLR_HIP(ctx, rocprim::radix_sort_pairs(tstore2, tsz2, tkey, skey, vin, vout, (size_t)nnz, 0, tbits, st));   // vin is still iota
  hipLaunchKernelGGL(k_gather_u32, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, st, ne, vout, nnz, sne);
  hipLaunchKernelGGL(k_tile_quads, dim3((unsigned)ceil_div(ntile, 256)), dim3(256), 0, st, tile_ne, (u32)ntile, tile_q4);
  i64 *ne_prefix, *tile_first, *tile_q;
  KLR_TRY(tmp.get(ctx, &ne_prefix, (size_t)nnz + 1)); KLR_TRY(tmp.get(ctx, &tile_first, (size_t)ntile + 1));
  KLR_TRY(tmp.get(ctx, &tile_q, (size_t)ntile + 1));
  KLR_TRY(exscan64(ctx, sne, nnz, ne_prefix));
  KLR_TRY(exscan64(ctx, tile_ne, ntile, tile_first));
  KLR_TRY(exscan64(ctx, tile_q4, ntile, tile_q));
  std::vector<i64> h_tq((size_t)ntile + 1);
  KLR_TRY(d2h_sync(ctx, h_tq.data(), tile_q, (size_t)(ntile + 1) * 8));
  const i64 total4 = h_tq[(size_t)ntile];
  if (total4 <= 0 || (u64)total4 * 4ull >= 0xFFFFFFF0ull) return KLR_OK;
  i64 h_stored = 0;
  KLR_TRY(d2h_sync(ctx, &h_stored, ne_prefix + nnz, 8));
  KLR_TRY(dalloc(ctx, &m->t_ent, (size_t)total4 * 4));                 // 16 bytes per quad (12 used by unit entries)
  if (u24) hipLaunchKernelGGL(k_fill_u24_null, dim3((unsigned)ceil_div(total4 * 3, 256)), dim3(256), 0, st, m->t_ent, total4 * 3);
  else KLR_HIP(ctx, hipMemsetAsync(m->t_ent, 0, (size_t)total4 * 16, st));
  m->fmt = u24 ? FMT_U24 : FMT_U32;
  hipLaunchKernelGGL(k_tile_scatter, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, st, skey, vout, nnz, (u32)ntile, ne_prefix,
                     tile_first, tile_q, m->col, val, erow, m->tile_r, m->tile_c, u24 ? 1 : 0, m->t_ent);
  KLR_HIP(ctx, hipGetLastError());
  KLR_HIP(ctx, hipStreamSynchronize(st));
  std::vector<u64> toff4((size_t)ntile + 1);
  for (i64 t = 0; t <= ntile; t++) toff4[(size_t)t] = (u64)h_tq[(size_t)t];
  m->stored_entries = h_stored;
  return build_pieces_and_upload(ctx, m, toff4, nrt, nct, rt_major);
}

// row id of every stored entry (one wavefront per row)
__global__ __launch_bounds__(256) void k_row_ids(const i64* __restrict__ rowptr, i64 n, u32* __restrict__ erow) {
  const i64 i = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  for (i64 a = rowptr[i] + (threadIdx.x & 63); a < rowptr[i + 1]; a += 64) erow[a] = (u32)i;
}

// CSC, tiles and host metadata of a count matrix whose narrow CSR (rowptr, u32 col, f32 val) and labels are already
// on the device: klr_mat_from_csr's route for large count data
int device_finish_csr(klr_ctx* ctx, klr_mat* m) {
  TmpPool tmp;
  u32* erow;
  KLR_TRY(tmp.get(ctx, &erow, (size_t)std::max<i64>(m->nnz, 1)));
  if (m->n) hipLaunchKernelGGL(k_row_ids, dim3((unsigned)ceil_div(m->n, 4)), dim3(256), 0, ctx->stream, m->rowptr, m->n, erow);
  KLR_HIP(ctx, hipGetLastError());
  KLR_TRY(finish_from_device_csr(ctx, m, erow, tmp));
  KLR_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return KLR_OK;
}

extern "C" int klr_mat_from_sequences(klr_ctx* ctx, const char* seqs, const int64_t* seq_off, int64_t n, int M, int N,
                                      int revcomp, int binarize, const uint8_t* labels, int64_t n_total, klr_mat** out) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, out != nullptr);
  *out = nullptr;
  klr_kmer_spec spec;
  memset(&spec, 0, sizeof(spec));
  spec.M = M; spec.N = N; spec.revcomp = revcomp ? 1 : 0; spec.alphabet = KLR_ALPHABET_NUCLEOTIDE;
  klr_kmer_classes* kc = nullptr;
  char eb[256] = "";
  if (klr_kmer_classes_create(&spec, &kc, eb, sizeof(eb)) != 0) return klr_fail(ctx, KLR_EINVAL, "%s", eb);
  const int rc = klr_mat_from_sequences_classes(ctx, seqs, seq_off, n, kc, binarize, labels, n_total, out);
  klr_kmer_classes_destroy(kc);
  return rc;
}

extern "C" int klr_mat_from_sequences_classes(klr_ctx* ctx, const char* seqs, const int64_t* seq_off, int64_t n,
                                              const klr_kmer_classes* classes, int binarize, const uint8_t* labels,
                                              int64_t n_total, klr_mat** out) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, out != nullptr);
  *out = nullptr;
  KLR_ARG(ctx, seqs != nullptr && seq_off != nullptr && labels != nullptr && classes != nullptr && n >= 1 && seq_off[0] == 0);
  const klr_kmer_spec* spec = klr_kmer_classes_spec(classes);
  const int M = spec->M, N = spec->N;
  const int gapped = spec->alphabet == KLR_ALPHABET_IUPAC ? 2 : (spec->alphabet == KLR_ALPHABET_GAPPED ? 1 : 0);   // window_variants' alpha
  if (n_total == 0) n_total = n;
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  TmpPool tmp;
  const i64 nbytes = seq_off[n];
  // ---- class tables (built on the host, klr_kmers.cpp): code -> class id, one table per length
  const uint32_t* h_map; const uint64_t* h_mapoff; int base;
  klr_kmer_classes_tables(classes, &h_map, &h_mapoff, &base);
  const u64 tot_codes = h_mapoff[(size_t)(N - M + 1)];
  u32* maps; unsigned long long* d_mapoff;
  KLR_TRY(tmp.get(ctx, &maps, (size_t)tot_codes)); KLR_TRY(tmp.get(ctx, &d_mapoff, (size_t)(N - M + 2)));
  KLR_HIP(ctx, hipMemcpyAsync(maps, h_map, (size_t)tot_codes * 4, hipMemcpyHostToDevice, st));
  KLR_HIP(ctx, hipMemcpyAsync(d_mapoff, h_mapoff, (size_t)(N - M + 2) * 8, hipMemcpyHostToDevice, st));
  const i64 dim = klr_kmer_classes_count(classes) + 1;
  // ---- sequences, windows
  unsigned char* d_seqs; i64* d_seqoff; u32* wcount; i64* woff; u32* d_ovf;
  KLR_TRY(tmp.get(ctx, &d_seqs, (size_t)nbytes)); KLR_TRY(tmp.get(ctx, &d_seqoff, (size_t)n + 1));
  KLR_TRY(tmp.get(ctx, &wcount, (size_t)n)); KLR_TRY(tmp.get(ctx, &woff, (size_t)n + 1)); KLR_TRY(tmp.get(ctx, &d_ovf, 1));
  KLR_HIP(ctx, hipMemsetAsync(d_ovf, 0, 4, st));
  KLR_HIP(ctx, hipMemcpyAsync(d_seqs, seqs, (size_t)nbytes, hipMemcpyHostToDevice, st));
  KLR_HIP(ctx, hipMemcpyAsync(d_seqoff, seq_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_row_windows, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, d_seqoff, n, M, N, gapped, wcount, d_ovf);
  KLR_TRY(exscan64(ctx, wcount, n, woff));
  i64 nwin = 0;
  KLR_TRY(d2h_sync(ctx, &nwin, woff + n, 8));
  u32 h_ovf = 0;
  KLR_TRY(d2h_sync(ctx, &h_ovf, d_ovf, 4));
  if (h_ovf || nwin >= (i64)0xFFFFFFF0ll)
    return klr_fail(ctx, KLR_EINVAL, "too many k-mer candidates (%lld) for one matrix: count fewer sequences at a time or bound --max-ambiguous",
                    (long long)nwin);
  u32 *ids, *ids_sorted, *head, *rowheads;
  KLR_TRY(tmp.get(ctx, &ids, (size_t)nwin)); KLR_TRY(tmp.get(ctx, &ids_sorted, (size_t)nwin));
  KLR_TRY(tmp.get(ctx, &head, (size_t)nwin)); KLR_TRY(tmp.get(ctx, &rowheads, (size_t)n));
  hipLaunchKernelGGL(k_emit_ids, dim3((unsigned)n), dim3(256), 0, st, d_seqs, d_seqoff, woff, n, M, N, gapped, maps, d_mapoff, ids);
  // ---- per-row sort (segments = rows)
  {
    size_t tsz = 0;
    KLR_HIP(ctx, rocprim::segmented_radix_sort_keys(nullptr, tsz, ids, ids_sorted, (unsigned)nwin, (unsigned)n, woff, woff + 1, 0, 32, st));
    void* ts; KLR_TRY(tmp.get(ctx, (char**)&ts, tsz));
    if (nwin) KLR_HIP(ctx, rocprim::segmented_radix_sort_keys(ts, tsz, ids, ids_sorted, (unsigned)nwin, (unsigned)n, woff, woff + 1, 0, 32, st));
  }
  hipLaunchKernelGGL(k_mark_heads, dim3((unsigned)n), dim3(256), 0, st, ids_sorted, woff, n, head, rowheads);
  // ---- CSR
  klr_mat* m = new klr_mat();
  m->ctx = ctx; m->device = ctx->device; m->n = n; m->dim = dim; m->n_total = n_total; m->n_total_all = n_total;
  m->val_f64 = false;
  m->h_labels.assign(labels, labels + n);
  for (i64 i = 0; i < n; i++) if (labels[i]) m->n1_local++;
  int rc = KLR_OK;
  do {
    if ((rc = dalloc(ctx, &m->rowptr, (size_t)n + 1))) break;
    if ((rc = exscan64(ctx, rowheads, n, m->rowptr))) break;
    i64 nnz = 0;
    if ((rc = d2h_sync(ctx, &nnz, m->rowptr + n, 8))) break;
    m->nnz = nnz;
    m->avg_row = (double)nnz / (double)n;
    if ((rc = dalloc(ctx, &m->col, (size_t)nnz))) break;
    if ((rc = dev_alloc(ctx, &m->val, (size_t)std::max<i64>(nnz, 1) * 4))) break;
    if ((rc = dalloc(ctx, &m->labels, (size_t)n))) break;
    u32* erow;
    if ((rc = tmp.get(ctx, &erow, (size_t)nnz))) break;
    hipLaunchKernelGGL(k_write_csr, dim3((unsigned)n), dim3(256), 0, st, ids_sorted, head, woff, m->rowptr, n, binarize, m->col,
                       (float*)m->val, erow);
    if (hipMemcpyAsync(m->labels, labels, (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess) { rc = klr_fail(ctx, KLR_EHIP, "label upload failed"); break; }
    if (hipGetLastError() != hipSuccess) { rc = klr_fail(ctx, KLR_EHIP, "k-mer counting kernels failed"); break; }
    rc = finish_from_device_csr(ctx, m, erow, tmp);
  } while (0);
  if (rc == KLR_OK && hipStreamSynchronize(st) != hipSuccess) rc = klr_fail(ctx, KLR_EHIP, "device ingest failed");
  if (rc != KLR_OK) { std::string keep = ctx->err; mat_free(m); ctx->err = keep; return rc; }
  *out = m;
  return KLR_OK;
}

// =================================================================================================================
// Co-occurrence features.  The reference evaluates the pair features on the fly inside LinearPdf /
// Gradient (kmerLr_logistic_regression.go:69-85, 183-195: every pair j1 < j2 of stored entries of a row contributes
// v[j1]*v[j2] at parameter CoeffIndex.Ind2Sub(i[j1]-1, i[j2]-1), kmerLr_coefficients_index.go:35-42) and writes
// them out only for selected columns (featureSelection.Data, kmerLr_feature_selection.go:325-330).  Here the pair
// matrix of a (pre-selected) base matrix is written out once, on the device, in CoeffIndex column order; after that
// it is an ordinary data matrix for the sweep, the selection, the column sub-select and the solver.
// =================================================================================================================
__device__ __forceinline__ i64 d_ind2sub(i64 n, i64 k1, i64 k2) {
  return k1 == k2 ? k1 + 1 : n + (n * (n - 1) / 2) - (n - k1) * ((n - k1) - 1) / 2 + k2 - k1;
}

__global__ __launch_bounds__(256) void k_cooc_count(const i64* __restrict__ rowptr, i64 n, u32* __restrict__ len, u32* __restrict__ too_long) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const i64 q = rowptr[i + 1] - rowptr[i] - 1;
  const i64 l = 1 + q + q * (q - 1) / 2;
  if (l >= ((i64)1 << 31)) { atomicOr(too_long, 1u); len[i] = 0; return; }
  len[i] = (u32)l;
}

// one wavefront per row: singles copied, then for every p1 the lanes take the p2 > p1
template <typename VT>
__global__ __launch_bounds__(256) void k_cooc_fill(const i64* __restrict__ rowptr, const u32* __restrict__ col, const VT* __restrict__ val,
                                                   i64 n, i64 nb, const i64* __restrict__ orow, u32* __restrict__ ocol,
                                                   double* __restrict__ oval, u32* __restrict__ erow, u32* __restrict__ not_count) {
  const i64 i = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const i64 b = rowptr[i], q1 = rowptr[i + 1] - b;            // incl. bias
  const i64 o = orow[i];
  u32 bad = 0;
  for (i64 p = lane; p < q1; p += 64) {
    const double v = (double)val[b + p];
    ocol[o + p] = col[b + p]; oval[o + p] = v; erow[o + p] = (u32)i;
    if (!(v >= 0.0 && v <= 16777216.0 && v == floor(v))) bad = 1;
  }
  const i64 q = q1 - 1;
  for (i64 p1 = 0; p1 + 1 < q; p1++) {
    const u32 c1 = col[b + 1 + p1];
    const double v1 = (double)val[b + 1 + p1];
    const i64 base = o + q1 + p1 * (2 * q - p1 - 1) / 2;       // pairs (p1, *) follow those of all p < p1
    for (i64 p2 = p1 + 1 + lane; p2 < q; p2 += 64) {
      const double v = v1 * (double)val[b + 1 + p2];
      const i64 pos = base + (p2 - p1 - 1);
      ocol[pos] = (u32)d_ind2sub(nb, (i64)c1 - 1, (i64)col[b + 1 + p2] - 1);
      oval[pos] = v; erow[pos] = (u32)i;
      if (!(v >= 0.0 && v <= 16777216.0 && v == floor(v))) bad = 1;
    }
  }
  if (bad) atomicOr(not_count, 1u);
}

__global__ __launch_bounds__(256) void k_f64_to_f32(const double* __restrict__ a, i64 n, float* __restrict__ b) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = (float)a[i];
}

extern "C" int64_t klr_coeff_dim(int64_t n) { return (n + 1) * n / 2 + 1; }
extern "C" int64_t klr_coeff_ind2sub(int64_t n, int64_t k1, int64_t k2) {
  return k1 == k2 ? k1 + 1 : n + (n * (n - 1) / 2) - (n - k1) * ((n - k1) - 1) / 2 + k2 - k1;
}
extern "C" void klr_coeff_sub2ind(int64_t n, int64_t i, int64_t* k1, int64_t* k2) {
  if (i < n) { if (k1) *k1 = i; if (k2) *k2 = i; return; }
  i -= n;
  const int64_t a = n - 2 - (int64_t)std::floor(std::sqrt((double)(-8 * i + 4 * n * (n - 1) - 7)) / 2.0 - 0.5);
  if (k1) *k1 = a;
  if (k2) *k2 = i + a + 1 - n * (n - 1) / 2 + (n - a) * ((n - a) - 1) / 2;
}

extern "C" int klr_mat_expand_cooccurrence(klr_ctx* ctx, const klr_mat* base, klr_mat** out) {
  if (!ctx) return klr_fail(nullptr, KLR_EINVAL, "ctx is NULL");
  KLR_ARG(ctx, base != nullptr && out != nullptr);
  *out = nullptr;
  const i64 n = base->n, nb = base->dim - 1;
  KLR_ARG(ctx, nb >= 1 && nb < 65536);                              // Dim = nb (nb+1)/2 + 1 must stay below 2^31
  const i64 dim = klr_coeff_dim(nb);
  KLR_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  TmpPool tmp;
  u32 *len, *flags;
  KLR_TRY(tmp.get(ctx, &len, (size_t)std::max<i64>(n, 1))); KLR_TRY(tmp.get(ctx, &flags, 2));
  KLR_HIP(ctx, hipMemsetAsync(flags, 0, 8, st));
  if (n) hipLaunchKernelGGL(k_cooc_count, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, base->rowptr, n, len, flags);
  klr_mat* m = new klr_mat();
  m->ctx = ctx; m->device = ctx->device; m->n = n; m->dim = dim; m->n_total = base->n_total; m->n_total_all = base->n_total_all;
  m->n1_local = base->n1_local; m->h_labels = base->h_labels; m->val_f64 = true;
  int rc = KLR_OK;
  do {
    if ((rc = dalloc(ctx, &m->rowptr, (size_t)n + 1))) break;
    if ((rc = exscan64(ctx, len, n, m->rowptr))) break;
    i64 nnz = 0;
    if ((rc = d2h_sync(ctx, &nnz, m->rowptr + n, 8))) break;
    u32 hflag = 0;
    if ((rc = d2h_sync(ctx, &hflag, flags, 4))) break;
    if (hflag || nnz >= ((i64)1 << 32)) { rc = klr_fail(ctx, KLR_EINVAL, "co-occurrence matrix too large (%lld entries): pre-select fewer features", (long long)nnz); break; }
    m->nnz = nnz;
    m->avg_row = n ? (double)nnz / (double)n : 0.0;
    double* dval; u32* erow;
    if ((rc = dalloc(ctx, &m->col, (size_t)nnz))) break;
    if ((rc = tmp.get(ctx, &dval, (size_t)nnz))) break;
    if ((rc = tmp.get(ctx, &erow, (size_t)nnz))) break;
    if ((rc = dalloc(ctx, &m->labels, (size_t)std::max<i64>(n, 1)))) break;
    if (n) {
      const dim3 grid((unsigned)ceil_div(n, 4));
      if (base->val_f64) hipLaunchKernelGGL(k_cooc_fill<double>, grid, dim3(256), 0, st, base->rowptr, base->col, (const double*)base->val, n, nb, m->rowptr, m->col, dval, erow, flags + 1);
      else               hipLaunchKernelGGL(k_cooc_fill<float>, grid, dim3(256), 0, st, base->rowptr, base->col, (const float*)base->val, n, nb, m->rowptr, m->col, dval, erow, flags + 1);
      if (hipMemcpyAsync(m->labels, base->labels, (size_t)n, hipMemcpyDeviceToDevice, st) != hipSuccess) { rc = klr_fail(ctx, KLR_EHIP, "label copy failed"); break; }
      if (base->active) {
        if (hipMalloc((void**)&m->active, (size_t)n) != hipSuccess) { rc = klr_fail(ctx, KLR_EHIP, "out of device memory"); break; }
        if (hipMemcpyAsync(m->active, base->active, (size_t)n, hipMemcpyDeviceToDevice, st) != hipSuccess) { rc = klr_fail(ctx, KLR_EHIP, "mask copy failed"); break; }
      }
    }
    if (hipGetLastError() != hipSuccess) { rc = klr_fail(ctx, KLR_EHIP, "co-occurrence kernels failed"); break; }
    u32 not_count = 0;
    if ((rc = d2h_sync(ctx, &not_count, flags + 1, 4))) break;
    if (!not_count) {
      // products of counts: the device ingest finishes CSC and tiles
      m->val_f64 = false;
      if ((rc = dev_alloc(ctx, &m->val, (size_t)std::max<i64>(nnz, 1) * 4))) break;
      if (nnz) hipLaunchKernelGGL(k_f64_to_f32, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, st, dval, nnz, (float*)m->val);
      rc = finish_from_device_csr(ctx, m, erow, tmp);
    } else {
      // real-valued scores (small, dense inputs): the host ingest builds the layouts from a copy of the CSR
      std::vector<i64> h_rp((size_t)n + 1), h_col64((size_t)nnz);
      std::vector<u32> h_col((size_t)nnz);
      std::vector<double> h_val((size_t)nnz);
      if ((rc = d2h_sync(ctx, h_rp.data(), m->rowptr, (size_t)(n + 1) * 8))) break;
      if (nnz && (rc = d2h_sync(ctx, h_col.data(), m->col, (size_t)nnz * 4))) break;
      if (nnz && (rc = d2h_sync(ctx, h_val.data(), dval, (size_t)nnz * 8))) break;
      for (i64 a = 0; a < nnz; a++) h_col64[(size_t)a] = h_col[(size_t)a];
      klr_mat* hm = nullptr;
      rc = klr_mat_from_csr(ctx, n, dim, h_rp.data(), h_col64.data(), h_val.data(), base->h_labels.data(), base->n_total_all, &hm);
      if (rc != KLR_OK) break;
      if (base->active) {
        std::vector<uint8_t> h_act((size_t)n);
        if ((rc = d2h_sync(ctx, h_act.data(), base->active, (size_t)n))) { mat_free(hm); break; }
        if ((rc = klr_mat_set_active_rows(ctx, hm, h_act.data(), base->n_total))) { mat_free(hm); break; }
      }
      mat_free(m);
      m = hm;
    }
  } while (0);
  if (rc == KLR_OK && hipStreamSynchronize(st) != hipSuccess) rc = klr_fail(ctx, KLR_EHIP, "co-occurrence expansion failed");
  if (rc != KLR_OK) { std::string keep = ctx->err; mat_free(m); ctx->err = keep; return rc; }
  *out = m;
  return KLR_OK;
}
