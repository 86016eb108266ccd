// Micro-benchmark: sustained L2 -> LDS ingest per CU with global_load_lds_dwordx4 (LDS-DMA), every CU
// streaming the SAME L2-resident buffer (the weight-stream pattern of csrc/block.hip).
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/glds_bench.hip -o /tmp/glds_bench && /tmp/glds_bench
// Variants: waves per workgroup (4 / 8 / 16), ring slots of 32 KiB (units in flight = slots - 1), with / without
// a barrier per unit, rotated start offset per workgroup, plain global_load_dwordx4 into registers for comparison.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void glds16(const unsigned char* sbase, int voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// UNIT bytes per step, NW waves, SLOTS ring slots; every wave issues UNIT / NW / 1024 instructions per unit
// the K-unit source pattern of csrc/block.hip (4 issuing waves): glds i of wave q covers 8 rows of 512 B, taking the
// 128-byte K tile q of each, 16-byte chunks XOR-swizzled
struct Tab { unsigned long long u[96]; };
__device__ __forceinline__ void glds16x4(const unsigned char* sbase, int o0, int o1, int o2, int o3, unsigned d0) {
  unsigned keep;
  const unsigned d1 = d0 + 1024, d2 = d0 + 2048, d3 = d0 + 3072;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
               "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\ts_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(sbase), "s"(d0), "s"(d1), "s"(d2), "s"(d3) : "memory");
}
// feat bits: 1 = unit addresses from a kernel-argument table, 2 = LDS code word written by wave 0 before / read by the
// loaders after every barrier, 4 = glds16x4 form, 8 = code read BEFORE the issue
template <int SLOTS>
__global__ __launch_bounds__(512) void kfeat_kernel(const unsigned char* __restrict__ w, const Tab tab, int nunits, int* sink, int feat) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  volatile int* sync = (volatile int*)(smem + 4 * 32768);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int UNIT = 32768;
  if (wave < 4) {
    for (int g = 0; g < nunits; ++g) {
      if (feat & 2) {
        if (threadIdx.x == 0) sync[g & 1] = 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    return;
  }
  const int q = wave & 3;
  const int gc = (lane & 7) ^ (lane >> 3);
  int kofs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) kofs[i] = (8 * i + (lane >> 3)) * 512 + q * 128 + gc * 16;
  int acc = 0;
  auto issue = [&](int g) {
    const unsigned char* base = (feat & 1) ? (const unsigned char*)tab.u[g] : w + (size_t)g * UNIT;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)base), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)base >> 32));
    base = (const unsigned char*)(((unsigned long long)hi << 32) | lo);
    const unsigned dst = (unsigned)((g % SLOTS) * UNIT + q * 8192);
    if (feat & 4) {
      glds16x4(base, kofs[0], kofs[1], kofs[2], kofs[3], dst);
      glds16x4(base, kofs[4], kofs[5], kofs[6], kofs[7], dst + 4096);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) glds16(base, kofs[i], dst + i * 1024);
    }
  };
  for (int g = 0; g < SLOTS - 1 && g < nunits; ++g) issue(g);
  for (int g = 0; g < nunits; ++g) {
    if (g + SLOTS - 2 < nunits) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if ((feat & 10) == 10) acc += __builtin_amdgcn_readfirstlane(sync[g & 1]);
    if (g + SLOTS - 1 < nunits) issue(g + SLOTS - 1);
    if ((feat & 10) == 2) acc += __builtin_amdgcn_readfirstlane(sync[g & 1]);
  }
  if (acc == 0x12345) *sink = 1;
}

template <int SLOTS>
__global__ __launch_bounds__(512) void kpat_kernel(const unsigned char* __restrict__ w, int nunits, int* sink, int prio) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int UNIT = 32768;
  if (wave < 4) {  // stand-ins for the compute waves: barriers only
    for (int g = 0; g < nunits; ++g) __builtin_amdgcn_s_barrier();
    return;
  }
  if (prio) __builtin_amdgcn_s_setprio(3);
  const int q = wave & 3;
  const int gc = (lane & 7) ^ (lane >> 3);
  int kofs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) kofs[i] = (8 * i + (lane >> 3)) * 512 + q * 128 + gc * 16;
  auto issue = [&](int g) {
    const unsigned char* base = w + (size_t)g * UNIT;
#pragma unroll
    for (int i = 0; i < 8; ++i) glds16(base, kofs[i], (unsigned)((g % SLOTS) * UNIT + q * 8192 + i * 1024));
  };
  for (int g = 0; g < SLOTS - 1 && g < nunits; ++g) issue(g);
  for (int g = 0; g < nunits; ++g) {
    if (g + SLOTS - 2 < nunits) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g + SLOTS - 1 < nunits) issue(g + SLOTS - 1);
  }
  if (smem[threadIdx.x] == 77 && smem[threadIdx.x + 1000] == 78) *sink = 1;
}

template <int NW, int SLOTS, int UNIT, bool BARRIER, bool ROTATE>
__global__ __launch_bounds__(NW * 64) void stream_kernel(const unsigned char* __restrict__ w, int nunits, int* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int PER = UNIT / NW / 1024;
  const int rot = ROTATE ? (blockIdx.x * 7) % nunits : 0;
  auto issue = [&](int g) {
    int u = g + rot;
    u = u >= nunits ? u - nunits : u;
    const unsigned char* base = w + (size_t)u * UNIT;
#pragma unroll
    for (int i = 0; i < PER; ++i)
      glds16(base, (wave * PER + i) * 1024 + lane * 16, (unsigned)((g % SLOTS) * UNIT + (wave * PER + i) * 1024));
  };
  for (int g = 0; g < SLOTS - 1 && g < nunits; ++g) issue(g);
  int acc = 0;
  for (int g = 0; g < nunits; ++g) {
    // wait for unit g: SLOTS - 2 later units may be in flight
    if (g + SLOTS - 2 < nunits) {
      if (SLOTS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (SLOTS == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
      else if (SLOTS == 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * PER) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (BARRIER) __builtin_amdgcn_s_barrier();
    if (g + SLOTS - 1 < nunits) issue(g + SLOTS - 1);
    acc += *(const int*)(smem + (g % SLOTS) * UNIT + threadIdx.x * 4);  // touch the unit
  }
  if (acc == 0x12345678) *sink = acc;
}

// plain loads into registers (no LDS): 16 B per lane per instruction, DEPTH instructions in flight per wave
template <int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void reg_kernel(const uint4* __restrict__ w, int nvec_per_wg, int* sink) {
  const int tid = threadIdx.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const int per_iter = NW * 64 * DEPTH;
  for (int i = 0; i < nvec_per_wg; i += per_iter) {
    uint4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = w[i + d * NW * 64 + tid];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) *sink = 1;
}

template <typename F>
float time_us(F launch, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

template <int NW, int SLOTS, int UNIT, bool BARRIER, bool ROTATE>
void run(const unsigned char* w, size_t bytes, int* sink, int grid) {
  const int nunits = (int)(bytes / UNIT);
  const int lds = SLOTS * UNIT;
  CK(hipFuncSetAttribute((const void*)stream_kernel<NW, SLOTS, UNIT, BARRIER, ROTATE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  float us = time_us([&] { hipLaunchKernelGGL((stream_kernel<NW, SLOTS, UNIT, BARRIER, ROTATE>), dim3(grid), dim3(NW * 64), lds, 0, w, nunits, sink); }, 20);
  printf("glds  waves %2d  slots %d x %2d KiB  barrier %d  rotate %d  grid %3d : %8.2f us  %6.1f GB/s per CU  (%.2f us per 32 KiB)\n", NW, SLOTS,
         UNIT / 1024, (int)BARRIER, (int)ROTATE, grid, us, bytes / us / 1e3, us / (bytes / 32768.0));
}

__global__ void thrash_kernel(const uint4* __restrict__ p, size_t n, int* sink) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) *sink = 1;
}

// the same stream with the L2s emptied before every launch (64 MiB of other data read in between): the weights
// then come from the Infinity Cache, as in the real encoder where every launch streams different weights
template <int NW, int SLOTS, int UNIT, bool BARRIER, bool ROTATE>
void run_cold(const unsigned char* w, size_t bytes, const uint4* other, size_t other_n, int* sink, int grid) {
  const int nunits = (int)(bytes / UNIT);
  const int lds = SLOTS * UNIT;
  CK(hipFuncSetAttribute((const void*)stream_kernel<NW, SLOTS, UNIT, BARRIER, ROTATE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  float t_both = time_us([&] {
    hipLaunchKernelGGL(thrash_kernel, dim3(1024), dim3(256), 0, 0, other, other_n, sink);
    hipLaunchKernelGGL((stream_kernel<NW, SLOTS, UNIT, BARRIER, ROTATE>), dim3(grid), dim3(NW * 64), lds, 0, w, nunits, sink); }, 10);
  float t_thr = time_us([&] { hipLaunchKernelGGL(thrash_kernel, dim3(1024), dim3(256), 0, 0, other, other_n, sink); }, 10);
  const float us = t_both - t_thr;
  printf("glds COLD L2  waves %2d  slots %d x %2d KiB  grid %3d : %8.2f us  %6.1f GB/s per CU  (%.2f us per 32 KiB)\n", NW, SLOTS,
         UNIT / 1024, grid, us, bytes / us / 1e3, us / (bytes / 32768.0));
}

int main(int argc, char** argv) {
  const size_t bytes = 2560 * 1024;  // the weights of one block<D|A> launch
  unsigned char* w; int* sink;
  CK(hipMalloc(&w, bytes + (1 << 20))); CK(hipMalloc(&sink, 4));
  CK(hipMemset(w, 1, bytes + (1 << 20)));
  {
    uint4* other; const size_t other_n = (64u << 20) / 16;
    CK(hipMalloc(&other, other_n * 16)); CK(hipMemset(other, 2, other_n * 16));
    run_cold<4, 4, 32768, true, false>(w, bytes, other, other_n, sink, 256);
    run_cold<8, 4, 32768, true, false>(w, bytes, other, other_n, sink, 256);
    run_cold<8, 5, 32768, true, false>(w, bytes, other, other_n, sink, 256);
    run_cold<8, 3, 32768, true, false>(w, bytes, other, other_n, sink, 256);
  }
  for (int feat : {0, 1, 2, 4, 7, 10, 15}) {
    const int nunits = (int)(bytes / 32768), lds = 4 * 32768 + 64;
    Tab tab;
    for (int g = 0; g < 96; ++g) tab.u[g] = (unsigned long long)(w + (size_t)(g % nunits) * 32768);
    CK(hipFuncSetAttribute((const void*)kfeat_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    float us = time_us([&] { hipLaunchKernelGGL((kfeat_kernel<4>), dim3(256), dim3(512), lds, 0, w, tab, nunits, sink, feat); }, 20);
    printf("glds K-unit pattern + features %2d : %8.2f us  (%.2f us per 32 KiB)\n", feat, us, us / (bytes / 32768.0));
  }
  for (int prio : {0, 1}) {
    const int nunits = (int)(bytes / 32768), lds = 4 * 32768;
    CK(hipFuncSetAttribute((const void*)kpat_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    float us = time_us([&] { hipLaunchKernelGGL((kpat_kernel<4>), dim3(256), dim3(512), lds, 0, w, nunits, sink, prio); }, 20);
    printf("glds K-unit pattern, 4 loader + 4 idle waves, prio %d : %8.2f us  %6.1f GB/s per CU  (%.2f us per 32 KiB)\n", prio, us, bytes / us / 1e3,
           us / (bytes / 32768.0));
  }
  for (int grid : {256}) {
    run<4, 4, 32768, true, false>(w, bytes, sink, grid);
    run<8, 4, 32768, true, false>(w, bytes, sink, grid);
    run<16, 4, 32768, true, false>(w, bytes, sink, grid);
    run<4, 4, 32768, false, false>(w, bytes, sink, grid);
    run<8, 4, 32768, false, false>(w, bytes, sink, grid);
    run<8, 4, 32768, true, true>(w, bytes, sink, grid);
    run<8, 3, 32768, true, false>(w, bytes, sink, grid);
    run<8, 2, 32768, true, false>(w, bytes, sink, grid);
    run<8, 5, 16384, true, false>(w, bytes, sink, grid);
    run<8, 5, 32768, true, false>(w, bytes, sink, grid);
    {
      const int nvec = (int)(bytes / 16);
      float us = time_us([&] { hipLaunchKernelGGL((reg_kernel<8, 8>), dim3(grid), dim3(512), 0, 0, (const uint4*)w, nvec, sink); }, 20);
      printf("regs  waves  8  depth 8 x 8 KiB                            grid %3d : %8.2f us  %6.1f GB/s per CU\n", grid, us, bytes / us / 1e3);
      us = time_us([&] { hipLaunchKernelGGL((reg_kernel<4, 16>), dim3(grid), dim3(256), 0, 0, (const uint4*)w, nvec, sink); }, 20);
      printf("regs  waves  4  depth 16 x 4 KiB                           grid %3d : %8.2f us  %6.1f GB/s per CU\n", grid, us, bytes / us / 1e3);
    }
  }
  return 0;
}
