// Micro-benchmark: weight fragments straight from L2 into MFMA operand registers (no LDS, no loader waves, no
// per-unit barrier) for the row-block kernels of csrc/block.hip.  One workgroup per CU, 4 waves; wave nf owns rows
// nf*16 .. +15 of every 64-row x 256-deep bf16 "K unit" (8 fragments of 16 bytes per lane) and multiplies them with
// 32 resident activation rows (16 MFMAs per unit), with the loads of unit u + P issued while unit u computes.
// Prints cycles per unit for pipeline depths P = 2, 3, 4, with and without a Swish-like epilogue every other unit.
//   hipcc --offload-arch=gfx950 -O3 direct_frag_bench.hip -o direct_frag_bench && ./direct_frag_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int P, bool EPI, bool PACKED>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ w, int units_total, int iters,
                                         long long* __restrict__ cycles, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63, nf = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  bf16x8 act[2][8];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) act[mi][ks][e] = (__bf16)(0.01f * ((lane + ks + e + mi) & 7));
  // lane's byte offset inside a unit.  PACKED = 0: row-major weights, row (nf*16 + lr) of 512 B, k-step ks at + ks*64,
  // lane group lg at + lg*16 (a wave-wide load touches 16 rows x 64 B).  PACKED = 1: fragment-major, the 64 lanes of a
  // load read 1 KiB contiguous: [nf][ks][lane][16 B].
  const unsigned lane_off = PACKED ? (nf * 8 * 64 + lane) * 16 : (nf * 16 + lr) * 512 + lg * 16;
  const unsigned ks_stride = PACKED ? 1024 : 64;
  bf16x8 wf[P][8];
  auto load = [&](int slot, int u) {
    const unsigned char* base = w + (size_t)(u % units_total) * 32768 + lane_off;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) wf[slot][ks] = *(const bf16x8*)(base + ks * ks_stride);
  };
  f32x4 tot[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // different workgroups start at different units, like row blocks that drift apart
  const int u0 = (blockIdx.x * 7) % units_total;
#pragma unroll
  for (int p = 0; p < P; ++p) load(p, u0 + p);
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it += P) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      f32x4 c[2][2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) c[mi][0] = c[mi][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          c[mi][ks & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[p][ks], act[mi][ks], c[mi][ks & 1], 0, 0, 0);
      load(p, u0 + it + p + P);  // refill this slot for P units later
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        f32x4 h = c[mi][0] + c[mi][1];
        if (EPI && (p & 1)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = h[r] * __builtin_amdgcn_rcpf(1.0f + __expf(-h[r]));
        }
        tot[mi] += h;
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 256 + threadIdx.x] = tot[0][0] + tot[1][1] + (float)wf[0][0][0];
}

template <int P, bool EPI, bool PACKED>
void run(const unsigned char* w, int units, long long* cyc, float* sink) {
  const int iters = 1200 / P * P;
  hipLaunchKernelGGL((k<P, EPI, PACKED>), dim3(256), dim3(256), 0, 0, w, units, iters, cyc, sink);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<P, EPI, PACKED>), dim3(256), dim3(256), 0, 0, w, units, iters, cyc, sink);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto v : h) avg += v;
  avg /= 256;
  printf("packed=%d P=%d epilogue=%d: %.0f cycles per unit (s_memtime), %.3f us per unit wall, %.1f GB/s per CU\n", (int)PACKED, P,
         (int)EPI, avg / iters, ms * 1e3 / iters, 32768.0 / (ms * 1e-3 / iters) / 1e9);
}

int main() {
  const int units = 80;  // 2.5 MB: the weight stream of block<D|A>
  unsigned char* w;
  long long* cyc;
  float* sink;
  hipMalloc((void**)&w, (size_t)units * 32768);
  hipMemset(w, 0x3c, (size_t)units * 32768);
  hipMalloc((void**)&cyc, 256 * sizeof(long long));
  hipMalloc((void**)&sink, 256 * 256 * sizeof(float));
  run<2, false, false>(w, units, cyc, sink);
  run<4, false, false>(w, units, cyc, sink);
  run<4, true, false>(w, units, cyc, sink);
  run<2, false, true>(w, units, cyc, sink);
  run<3, false, true>(w, units, cyc, sink);
  run<4, false, true>(w, units, cyc, sink);
  run<6, false, true>(w, units, cyc, sink);
  run<2, true, true>(w, units, cyc, sink);
  run<4, true, true>(w, units, cyc, sink);
  run<6, true, true>(w, units, cyc, sink);
  return 0;
}
