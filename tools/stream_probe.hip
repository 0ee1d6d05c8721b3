// Dev tool: read-bandwidth probe for the shapes the sweep kernel could stream its entry array in.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe tools/stream_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: every workgroup owns one contiguous chunk (the sweep kernel's shape)
// MODE 1: grid-stride, the whole grid reads one moving window
template <int THREADS, int MODE, bool NT, int LDSB>
__global__ __launch_bounds__(THREADS) void k_read(const u32x4* __restrict__ src, size_t nquad, unsigned long long* __restrict__ out) {
  extern __shared__ unsigned long long lds[];
  if (LDSB) lds[threadIdx.x] = 0;
  unsigned long long acc = 0;
  if (MODE == 0) {
    const size_t per = (nquad + gridDim.x - 1) / gridDim.x;
    const size_t q0 = per * blockIdx.x, q1 = q0 + per < nquad ? q0 + per : nquad;
    for (size_t q = q0 + threadIdx.x; q < q1; q += THREADS) {
      const u32x4 v = NT ? __builtin_nontemporal_load(src + q) : src[q];
      acc += v.x + v.y + v.z + v.w;
    }
  } else {
    for (size_t q = (size_t)blockIdx.x * THREADS + threadIdx.x; q < nquad; q += (size_t)gridDim.x * THREADS) {
      const u32x4 v = NT ? __builtin_nontemporal_load(src + q) : src[q];
      acc += v.x + v.y + v.z + v.w;
    }
  }
  if (acc == 0x123456789ull) out[0] = acc + (LDSB ? lds[0] : 0);
}

// the sweep kernel's inner loop on a flat chunk: UNR loads per trip (compiler's choice if 0), optional consume
template <int THREADS, int UNR, bool CONSUME>
__global__ __launch_bounds__(THREADS) void k_like(const u32x4* __restrict__ src, size_t nquad, unsigned long long* __restrict__ out) {
  extern __shared__ unsigned long long lds[];
  unsigned long long* sS = lds;
  unsigned long long* sG = lds + 4096;
  for (int i = threadIdx.x; i < 4096; i += THREADS) { sS[i] = i * 0x9E3779B97F4A7C15ull; sG[i] = 0; }
  __syncthreads();
  unsigned long long acc = 0;
  const size_t per = (nquad + gridDim.x - 1) / gridDim.x;
  const size_t q0 = per * blockIdx.x, q1 = q0 + per < nquad ? q0 + per : nquad;
  auto body = [&](size_t q) {
    const u32x4 v = __builtin_nontemporal_load(src + q);
    if (CONSUME) {
      const u32 e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 row = e[k] & 4095u, col = (e[k] >> 12) & 4095u, c = e[k] >> 24;
        atomicAdd(&sG[col], sS[row] * (unsigned long long)c);
      }
    } else acc += v.x + v.y + v.z + v.w;
  };
  if (UNR == 1) {
#pragma unroll 1
    for (size_t q = q0 + threadIdx.x; q < q1; q += THREADS) body(q);
  } else {
    for (size_t q = q0 + threadIdx.x; q < q1; q += THREADS) body(q);
  }
  __syncthreads();
  if (CONSUME) acc = sG[threadIdx.x];
  if (acc == 0x123456789ull) out[0] = acc;
}

template <int THREADS, int UNR, bool CONSUME>
static void run_like(const char* name, const u32x4* src, size_t nquad, unsigned long long* out, int grid) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)k_like<THREADS, UNR, CONSUME>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  std::vector<float> t;
  for (int r = 0; r < 12; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_like<THREADS, UNR, CONSUME>), dim3(grid), dim3(THREADS), 65536, 0, src, nquad, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  const double us = t[t.size() / 2] * 1e3;
  printf("%-44s grid %6d  median %8.2f us  %6.3f TB/s\n", name, grid, us, (double)nquad * 16 / us * 1e-6);
}

// 12 bytes per lane (one dwordx3), every workgroup one contiguous chunk: the unit-entry sweep's access shape;
// run under `rocprofv3 --pmc FETCH_SIZE` it calibrates the counter for this width (known byte count)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_read12(const u32* __restrict__ src, size_t ntrip, unsigned long long* __restrict__ out) {
  unsigned long long acc = 0;
  const size_t per = (ntrip + gridDim.x - 1) / gridDim.x;
  const size_t q0 = per * blockIdx.x, q1 = q0 + per < ntrip ? q0 + per : ntrip;
  for (size_t q = q0 + threadIdx.x; q < q1; q += THREADS) {
    const u32* p = src + 3 * q;
    const u32 a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 1), c = __builtin_nontemporal_load(p + 2);
    acc += a + b + c;
  }
  if (acc == 0x123456789ull) out[0] = acc;
}

static void run12(const u32x4* src, size_t bytes, unsigned long long* out, int grid) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t ntrip = bytes / 12;
  std::vector<float> t;
  for (int r = 0; r < 12; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_read12<1024>), dim3(grid), dim3(1024), 0, 0, (const u32*)src, ntrip, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  const double us = t[t.size() / 2] * 1e3;
  printf("%-44s grid %6d  median %8.2f us  %6.3f TB/s  (%zu bytes per launch)\n", "chunk 1024thr nt 12 B/lane (dwordx3)", grid, us,
         (double)ntrip * 12 / us * 1e-6, ntrip * 12);
}

template <int THREADS, int MODE, bool NT, int LDSB>
static void run(const char* name, const u32x4* src, size_t nquad, unsigned long long* out, int grid) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  if (LDSB) CK(hipFuncSetAttribute((const void*)k_read<THREADS, MODE, NT, LDSB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
  std::vector<float> t;
  for (int r = 0; r < 12; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_read<THREADS, MODE, NT, LDSB>), dim3(grid), dim3(THREADS), LDSB, 0, src, nquad, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  const double us = t[t.size() / 2] * 1e3;
  printf("%-44s grid %6d  median %8.2f us  %6.3f TB/s\n", name, grid, us, (double)nquad * 16 / us * 1e-6);
}

__global__ void k_fill(u32* p, size_t nw, u32 mode) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long x = i * 0x9E3779B97F4A7C15ull + 12345; x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
    p[i] = mode == 1 ? (u32)x : ((1u << 24) | ((u32)x & 0xFFF000u) | (u32)((i >> 6) & 0xFFFu));   // 1: random, 2: entry-like
  }
}

int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)393457488;
  const size_t nquad = bytes / 16;
  u32x4* src; unsigned long long* out;
  CK(hipMalloc(&src, nquad * 16)); CK(hipMalloc(&out, 64));
  const int fill = argc > 2 ? atoi(argv[2]) : 0;
  if (fill == 0) CK(hipMemset(src, 1, nquad * 16));
  else hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (u32*)src, nquad * 4, (u32)fill);
  CK(hipDeviceSynchronize());
  printf("fill mode %d\n", fill);
  int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  printf("bytes %zu  CUs %d\n", bytes, ncu);
  run12(src, nquad * 16, out, 2 * ncu);
  if (argc > 3 && argv[3][0] == 'c') return 0;           // calibration run: the 12-byte kernel only
  run_like<1024, 1, false>("like: unroll 1, stream only", src, nquad, out, 2 * ncu);
  run_like<1024, 0, false>("like: default unroll, stream only", src, nquad, out, 2 * ncu);
  run_like<1024, 1, true>("like: unroll 1, consume", src, nquad, out, 2 * ncu);
  run_like<1024, 0, true>("like: default unroll, consume", src, nquad, out, 2 * ncu);
  if (argc > 3) return 0;
  run<1024, 0, true, 65536>("chunk 1024thr nt lds64k (sweep shape)", src, nquad, out, 2 * ncu);
  run<1024, 0, false, 65536>("chunk 1024thr lds64k", src, nquad, out, 2 * ncu);
  run<1024, 1, true, 65536>("stride 1024thr nt lds64k", src, nquad, out, 2 * ncu);
  run<1024, 1, false, 65536>("stride 1024thr lds64k", src, nquad, out, 2 * ncu);
  run<1024, 0, true, 0>("chunk 1024thr nt", src, nquad, out, 2 * ncu);
  run<1024, 1, true, 0>("stride 1024thr nt", src, nquad, out, 2 * ncu);
  run<256, 0, true, 0>("chunk 256thr nt", src, nquad, out, 8 * ncu);
  run<256, 1, true, 0>("stride 256thr nt", src, nquad, out, 8 * ncu);
  run<256, 1, false, 0>("stride 256thr", src, nquad, out, 8 * ncu);
  run<256, 1, true, 0>("stride 256thr nt", src, nquad, out, 16 * ncu);
  run<256, 1, true, 0>("stride 256thr nt", src, nquad, out, 32 * ncu);
  run<256, 0, true, 0>("chunk 256thr nt", src, nquad, out, 32 * ncu);
  run<256, 0, true, 0>("chunk 256thr nt", src, nquad, out, 128 * ncu);
  run<256, 0, true, 0>("chunk 256thr nt (1 quad/thread)", src, nquad, out, (int)((nquad + 255) / 256));
  run<512, 0, true, 32768>("chunk 512thr nt lds32k", src, nquad, out, 4 * ncu);
  run<512, 0, true, 32768>("chunk 512thr nt lds32k", src, nquad, out, 16 * ncu);
  run<1024, 0, true, 65536>("chunk 1024thr nt lds64k", src, nquad, out, 8 * ncu);
  run<1024, 0, true, 65536>("chunk 1024thr nt lds64k", src, nquad, out, 32 * ncu);
  return 0;
}
