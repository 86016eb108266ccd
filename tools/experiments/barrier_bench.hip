// How long is one s_barrier interval for a workgroup of NW waves (1 workgroup per CU, 160 KiB LDS)?
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/barrier_bench.hip -o tools/experiments/barrier_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

template <int NW, int WORK>
__global__ __launch_bounds__(NW * 64) void bar_kernel(long long* out, int iters) {
  extern __shared__ unsigned char smem[];
  float* s = (float*)smem;
  float acc = threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (WORK == 1) {  // a little LDS traffic before every barrier (what an interval of block.hip does at least)
      s[threadIdx.x] = acc;
      acc += s[(threadIdx.x + 64) % (NW * 64)];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)acc; }
}

template <int NW, int WORK>
int run(long long* d, int lds) {
  const int iters = 2000;
  CK(hipFuncSetAttribute((const void*)bar_kernel<NW, WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL((bar_kernel<NW, WORK>), dim3(256), dim3(NW * 64), lds, 0, d, iters);
  long long h[2];
  CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  printf("waves %2d  lds %6d  work %d : %6.1f cycles per barrier interval\n", NW, lds, WORK, (double)h[0] / iters);
  return 0;
}

int main() {
  long long* d;
  CK(hipMalloc(&d, 16));
  for (int lds : {4096, 162816}) {
    run<4, 0>(d, lds); run<8, 0>(d, lds); run<16, 0>(d, lds);
    run<4, 1>(d, lds); run<8, 1>(d, lds); run<16, 1>(d, lds);
  }
  return 0;
}
