// Dev tool: LDS cost of the sweep's inner step (4 x ds_read_b64 of s + 4 x ds_add_u64 into g per lane) for different
// arrangements of the (row, column) pairs inside a wavefront.  No global traffic: pure LDS pipe time.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/lds_probe tools/lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned int u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ u32 hash32(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// PAT: 0 random columns, rows in runs of 60 (today's row-major order)
//      1 columns: bank pair = lane & 31 (conflict free per 32-lane half), rows in runs
//      2 columns: bank pair distinct inside every 16-lane group only, rows in runs
//      3 random columns, random rows
//      4 conflict-free columns, random rows
//      5 all lanes the same column (same-address atomics)
//      6 reads only (random columns unused), rows in runs
//      7 atomics only, random columns
template <int PAT>
__global__ __launch_bounds__(1024) void k_lds(int iters, unsigned long long* out) {
  extern __shared__ unsigned long long lds[];
  unsigned long long* sS = lds;
  unsigned long long* sG = lds + 4096;
  const u32 tid = threadIdx.x, lane = tid & 63;
  for (u32 i = tid; i < 4096; i += 1024) { sS[i] = i * 0x9E3779B97F4A7C15ull; sG[i] = 0; }
  __syncthreads();
  unsigned long long acc = 0;
  for (int it = 0; it < iters; it++) {
    u32 row[4], col[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 pos = ((u32)it * 1024u + tid) * 4u + k;                 // entry position in the stream
      const u32 h = hash32(pos * 2654435761u + 12345u);
      row[k] = (PAT == 3 || PAT == 4) ? (h >> 8) % 3334u : (pos / 60u) % 3334u;
      if (PAT == 0 || PAT == 3 || PAT == 7 || PAT == 6 || PAT >= 8) col[k] = h % 3856u;
      else if (PAT == 1 || PAT == 4) col[k] = ((h % 120u) * 32u + (lane & 31u)) % 3856u;      // bank pair = lane & 31
      else if (PAT == 2) col[k] = ((h % 120u) * 32u + ((lane & 15u) * 2u + (h >> 20 & 1u))) % 3856u;
      else col[k] = 7u;
    }
    if (PAT >= 13 && PAT <= 15) {
      // q64 shape: 2 quads per lane and trip, 8 s reads + 2 atomics.  Runs of 50 rows per column; the lane's rows:
      //   13 ascending rows of the run in order (row-sorted runs: what a column-major tile gives)
      //   14 the run's rows ordered by (row mod 32): consecutive positions walk through the banks
      //   15 random rows
      unsigned long long sum[2];
#pragma unroll
      for (int h2 = 0; h2 < 2; h2++) {
        unsigned long long sacc = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const u32 pos = ((u32)it * 1024u + tid) * 8u + h2 * 4 + k;       // position in the stream
          const u32 run = pos / 50u, pr = pos % 50u;
          const u32 hr = hash32(run * 2654435761u + 777u);
          u32 r;
          if (PAT == 13) r = (pr * 66u + (hash32(pos) % 60u) + hr % 66u) % 3334u;                       // ascending, gaps ~66
          else if (PAT == 14) r = (((pr * 32u) / 50u + hr) % 32u + 32u * (hash32(pos) % 104u)) % 3334u;  // bank = position-ordered
          else r = hash32(pos * 7u + 3u) % 3334u;
          sacc += sS[r];
        }
        sum[h2] = sacc;
      }
      const u32 posq = ((u32)it * 1024u + tid) * 2u;
      atomicAdd(&sG[hash32(posq / 12u) % 3856u], sum[0]);
      atomicAdd(&sG[hash32((posq + 1) / 12u) % 3856u], sum[1]);
    } else if (PAT == 8) {                                   // 32-bit atomics, random columns
      u32* g32 = (u32*)sG;
      atomicAdd(&g32[col[0]], 1u); atomicAdd(&g32[col[1]], 2u); atomicAdd(&g32[col[2]], 3u); atomicAdd(&g32[col[3]], 4u);
    } else if (PAT == 9) {                            // 64-bit atomics, half of the lanes active
      if (lane & 1) { atomicAdd(&sG[col[0]], 1ull); atomicAdd(&sG[col[1]], 2ull); atomicAdd(&sG[col[2]], 3ull); atomicAdd(&sG[col[3]], 4ull); }
    } else if (PAT == 10) {                           // 64-bit atomics, a quarter of the lanes active
      if ((lane & 3) == 0) { atomicAdd(&sG[col[0]], 1ull); atomicAdd(&sG[col[1]], 2ull); atomicAdd(&sG[col[2]], 3ull); atomicAdd(&sG[col[3]], 4ull); }
    } else if (PAT == 11) {                           // two 32-bit atomics per entry (lo, hi halves)
      u32* g32 = (u32*)sG;
#pragma unroll
      for (int k = 0; k < 4; k++) { atomicAdd(&g32[2 * col[k]], row[k]); atomicAdd(&g32[2 * col[k] + 1], 1u); }
    } else if (PAT == 12) {                           // f64 atomics
      double* gd = (double*)sG;
      atomicAdd(&gd[col[0]], 1.0); atomicAdd(&gd[col[1]], 2.0); atomicAdd(&gd[col[2]], 3.0); atomicAdd(&gd[col[3]], 4.0);
    } else if (PAT != 7) {
      unsigned long long s0 = sS[row[0]], s1 = sS[row[1]], s2 = sS[row[2]], s3 = sS[row[3]];
      if (PAT == 6) { acc += s0 + s1 + s2 + s3; continue; }
      atomicAdd(&sG[col[0]], s0); atomicAdd(&sG[col[1]], s1); atomicAdd(&sG[col[2]], s2); atomicAdd(&sG[col[3]], s3);
    } else {
      atomicAdd(&sG[col[0]], 1ull); atomicAdd(&sG[col[1]], 2ull); atomicAdd(&sG[col[2]], 3ull); atomicAdd(&sG[col[3]], 4ull);
    }
  }
  __syncthreads();
  acc += sG[tid];
  if (acc == 0x123456789ull) out[0] = acc;
}

template <int PAT>
static void run(const char* name, unsigned long long* out, int ncu) {
  CK(hipFuncSetAttribute((const void*)k_lds<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 400;
  std::vector<float> t;
  for (int r = 0; r < 8; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_lds<PAT>), dim3(2 * ncu), dim3(1024), 65536, 0, iters, out);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  const double us = t[t.size() / 2] * 1e3;
  // per CU: 2 workgroups x 16 waves x iters wave-iterations
  const double ns_per_wave_iter = us * 1e3 / (32.0 * iters);
  printf("%-58s %8.1f us  %6.2f ns per wave-iteration per CU  (%5.1f clk @2.4GHz; budget for 6.4 TB/s of 3-byte entries: 30.7 ns)\n", name, us,
         ns_per_wave_iter, ns_per_wave_iter * 2.4);
}

int main() {
  unsigned long long* out; CK(hipMalloc(&out, 64));
  int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  run<0>("0 random columns, rows in runs (today)", out, ncu);
  run<1>("1 column bank pair = lane & 31, rows in runs", out, ncu);
  run<2>("2 column banks distinct per 16 lanes, rows in runs", out, ncu);
  run<3>("3 random columns, random rows", out, ncu);
  run<4>("4 conflict-free columns, random rows", out, ncu);
  run<5>("5 one column for all lanes", out, ncu);
  run<6>("6 reads only, rows in runs", out, ncu);
  run<7>("7 atomics only, random columns", out, ncu);
  run<8>("8 32-bit atomics only, random columns", out, ncu);
  run<9>("9 64-bit atomics, half of the lanes active", out, ncu);
  run<10>("10 64-bit atomics, a quarter of the lanes active", out, ncu);
  run<11>("11 two 32-bit atomics per entry", out, ncu);
  run<12>("12 f64 atomics, random columns", out, ncu);
  run<13>("13 q64: 8 reads (row-sorted runs) + 2 atomics [8 entries]", out, ncu);
  run<14>("14 q64: 8 reads (bank-ordered runs) + 2 atomics [8 entries]", out, ncu);
  run<15>("15 q64: 8 reads (random rows) + 2 atomics [8 entries]", out, ncu);
  return 0;
}
