// Dev tool (round 2): cost of one LDS wave-instruction of the kinds a sweep kernel could use, as a function of the address
// pattern and of the LDS range it spreads over.  Pure LDS: no global traffic in the timed loop.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/lds_probe2 tools/lds_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned int u32;
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ u32 hash32(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// OP: 0 ds_add_u64, 1 ds_add_u32, 2 ds_add_f64, 3 ds_read_b64, 4 ds_read_b32, 5 two ds_read_b32 (hi / lo arrays), 6 ds_add_rtn_u64,
//     7 ds_max_u64, 8 ds_write_b64
template <int OP, int pattern>
__global__ __launch_bounds__(1024) void k_probe(int iters, u32 slots /* 8-byte slots the addresses spread over */, u64* out) {
  extern __shared__ u64 lds[];
  const u32 tid = threadIdx.x, lane = tid & 63;
  for (u32 i = tid; i < slots; i += 1024) lds[i] = i;
  __syncthreads();
  u64 acc = 0;
  u32 seed = hash32(tid * 2654435761u + blockIdx.x);
  for (int it = 0; it < iters; it++) {
    u32 idx[8];
    // cheap indices (an LCG per lane, power-of-two ranges): the loop must be bound by the LDS pipe, not by the index arithmetic
#pragma unroll
    for (int k = 0; k < 8; k++) {
      seed = seed * 1664525u + 1013904223u;
      const u32 h = seed >> 10;
      if (pattern == 0) idx[k] = h & (slots - 1u);                                  // random
      else if (pattern == 1) idx[k] = ((h << 6) | lane) & (slots - 1u);               // lane l -> bank pair l: conflict free
      else if (pattern == 2) idx[k] = ((h << 5) | (lane & 31u)) & (slots - 1u);       // conflict free per half wave
      else if (pattern <= 5) {                                                        // every bank pair hit by m = 2, 3, 4 lanes of the half wave
        const u32 m = (u32)pattern - 1u, l31 = lane & 31u;
        idx[k] = ((h << 5) | (m == 2u ? (l31 & ~1u) : (m == 3u ? (l31 / 3u) * 3u : (l31 & ~3u)))) & (slots - 1u);
      } else if (pattern == 6) {                                                      // one bank pair hit by 4 lanes, the others by one
        const u32 l31 = lane & 31u;
        idx[k] = ((h << 5) | (l31 < 4u ? 0u : l31)) & (slots - 1u);
      } else {                                                                        // 4 lanes of a 16-lane group share (slot mod 16), 2 and 2 share (slot mod 32)
        const u32 l15 = lane & 15u;
        idx[k] = ((h << 5) | ((l15 >> 2) + 16u * (l15 & 1u))) & (slots - 1u);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (OP == 0) atomicAdd(&lds[idx[k]], (u64)(it + k));
      else if (OP == 1) atomicAdd(&((u32*)lds)[idx[k] * 2u], (u32)(it + k));
      else if (OP == 2) atomicAdd(&((double*)lds)[idx[k]], 1.0);
      else if (OP == 3) acc += lds[idx[k]];
      else if (OP == 4) acc += ((u32*)lds)[idx[k] * 2u];
      else if (OP == 5) acc += (u64)((u32*)lds)[idx[k]] + ((u64)((u32*)lds)[slots + idx[k]] << 32);
      else if (OP == 6) acc += atomicAdd(&lds[idx[k]], (u64)(it + k));
      else if (OP == 7) atomicMax(&lds[idx[k]], (u64)(it + k));
      else lds[idx[k]] = (u64)(it + k) + acc;
    }
  }
  __syncthreads();
  acc += lds[tid % slots];
  if (acc == 0x123456789ull) out[0] = acc;
}

template <int OP, int pattern>
static void run(const char* name, u64* out, int ncu, u32 slots, int wgs_per_cu) {
  const size_t lds_bytes = (size_t)slots * 8;
  CK(hipFuncSetAttribute((const void*)k_probe<OP, pattern>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 300;
  std::vector<float> t;
  for (int r = 0; r < 6; r++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_probe<OP, pattern>), dim3(wgs_per_cu * ncu), dim3(1024), lds_bytes, 0, iters, slots, out);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  const double us = t[t.size() / 2] * 1e3;
  const double instr_per_cu = (double)wgs_per_cu * 16.0 * iters * 8.0;         // wave-instructions per CU
  printf("%-34s %6u KiB %s %d wg/cu : %8.1f us  %6.2f clk per wave-instruction @2.4GHz\n", name, (u32)(lds_bytes >> 10),
         pattern == 0 ? "random   " : (pattern == 1 ? "bank=lane" : (pattern == 2 ? "bank=l&31" : (pattern == 3 ? "all x2   " :
         (pattern == 4 ? "all x3   " : (pattern == 5 ? "all x4   " : (pattern == 6 ? "one x4   " : "x4mod16  ")))))), wgs_per_cu, us, us * 1e-6 * 2.4e9 / instr_per_cu);
}

int main() {
  u64* out; CK(hipMalloc(&out, 64));
  int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  for (int w = 1; w <= 2; w++) {
    const u32 sizes[] = {4096u, 8192u, 16384u};
    for (u32 s : sizes) {
      if (w == 2 && s > 8192u) continue;
      run<0, 0>("ds_add_u64", out, ncu, s, w);
      run<0, 1>("ds_add_u64", out, ncu, s, w);
      run<0, 2>("ds_add_u64", out, ncu, s, w);
    }
    run<0, 3>("ds_add_u64", out, ncu, 4096u, w);
    run<0, 4>("ds_add_u64", out, ncu, 4096u, w);
    run<0, 5>("ds_add_u64", out, ncu, 4096u, w);
    run<0, 6>("ds_add_u64", out, ncu, 4096u, w);
    run<0, 7>("ds_add_u64", out, ncu, 4096u, w);
    run<1, 0>("ds_add_u32", out, ncu, 4096u, w);
    run<1, 1>("ds_add_u32", out, ncu, 4096u, w);
    run<2, 0>("ds_add_f64", out, ncu, 4096u, w);
    run<6, 0>("ds_add_rtn_u64", out, ncu, 4096u, w);
    run<7, 0>("ds_max_u64", out, ncu, 4096u, w);
    run<8, 0>("ds_write_b64", out, ncu, 4096u, w);
    run<3, 0>("ds_read_b64", out, ncu, 4096u, w);
    run<3, 1>("ds_read_b64", out, ncu, 4096u, w);
    run<3, 5>("ds_read_b64", out, ncu, 4096u, w);
    run<4, 0>("ds_read_b32", out, ncu, 4096u, w);
    run<4, 1>("ds_read_b32", out, ncu, 4096u, w);
    run<5, 0>("2 x ds_read_b32 (hi, lo arrays)", out, ncu, 4096u, w);
  }
  return 0;
}
