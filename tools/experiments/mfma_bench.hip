// Cycles per v_mfma_f32_16x16x32_bf16 for the dependency patterns of csrc/block.hip (one wave per SIMD, 4 waves per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_bench.hip -o tools/experiments/mfma_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

// VARIANT 0: 4 accumulator chains, operands in registers; 1: + 8 ds_read_b128 per 16 MFMAs (results consumed next round);
// 2: variant 1 + one s_barrier per 16 MFMAs; 3: 8 accumulators (the W2 pattern); 4: variant 0 with 2 chains only
template <int VARIANT, int NW>
__global__ __launch_bounds__(NW * 64) void k(long long* out, int iters, float* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[32768];
  const int lane = threadIdx.x & 63;
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(0.001f * (lane + i + e)); b[i][e] = (__bf16)(0.002f * (lane ^ (i + e))); }
  for (int i = threadIdx.x; i < 8192; i += NW * 64) ((float*)smem)[i] = 0.001f * i;
  __syncthreads();
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    bf16x8 w[8];
    if (VARIANT == 1 || VARIANT == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) w[i] = *(const bf16x8*)(smem + ((threadIdx.x * 16 + i * 4096 + it * 64) & 32752));
    }
    if (VARIANT == 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (VARIANT == 3) {
        c[ks] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], b[0], c[ks], 0, 0, 0);
        c[ks] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], b[1], c[ks], 0, 0, 0);
      } else if (VARIANT == 4) {
        c[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], b[ks], c[0], 0, 0, 0);
        c[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], b[(ks + 1) & 7], c[1], 0, 0, 0);
      } else {
        c[(ks & 1) * 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], b[ks], c[(ks & 1) * 2], 0, 0, 0);
        c[(ks & 1) * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], b[(ks + 1) & 7], c[(ks & 1) * 2 + 1], 0, 0, 0);
      }
    }
    if (VARIANT == 1 || VARIANT == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = w[i];
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (s == 12345.678f) *sink = s;
}

// the interval of csrc/block.hip: barrier, 8 fragment reads of the NEXT unit into the other register set, 16 MFMAs from
// the current set (no register copies: two intervals per iteration).  EPI = 1 adds a Swish epilogue (8 exp + 8 rcp,
// bias read, 2 ds_write_b64) to every other interval.
template <int NW, int EPI>
__global__ __launch_bounds__(NW * 64) void pipe(long long* out, int iters, float* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  const int lane = threadIdx.x & 63;
  bf16x8 b[8], wa[8], wb[8];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) { b[i][e] = (__bf16)(0.002f * (lane ^ (i + e))); wa[i][e] = (__bf16)0.01f; }
  for (int i = threadIdx.x; i < 16384; i += NW * 64) ((float*)smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  f32x4 c[4];
  for (int i = 0; i < 4; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float acc = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it += 2) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) wb[i] = *(const bf16x8*)(smem + ((threadIdx.x * 16 + i * 4096 + it * 64) & 65520));
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      c[(ks & 1) * 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks], b[ks], c[(ks & 1) * 2], 0, 0, 0);
      c[(ks & 1) * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks], b[(ks + 1) & 7], c[(ks & 1) * 2 + 1], 0, 0, 0);
    }
    if (EPI) {
      const float4 bb = *(const float4*)(smem + (lane & 15) * 16);
      f32x4 h0 = c[0] + c[2], h1 = c[1] + c[3];
      float v[8] = {h0[0] + bb.x, h0[1] + bb.y, h0[2] + bb.z, h0[3] + bb.w, h1[0] + bb.x, h1[1] + bb.y, h1[2] + bb.z, h1[3] + bb.w};
      unsigned short pk[8];
      for (int e = 0; e < 8; ++e) { const float sw = v[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[e])); __bf16 hb = (__bf16)sw; pk[e] = *(unsigned short*)&hb; }
      *(uint2*)(smem + 32768 + threadIdx.x * 8) = make_uint2(pk[0] | (pk[1] << 16), pk[2] | (pk[3] << 16));
      *(uint2*)(smem + 40960 + threadIdx.x * 8) = make_uint2(pk[4] | (pk[5] << 16), pk[6] | (pk[7] << 16));
      for (int i = 0; i < 4; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) wa[i] = *(const bf16x8*)(smem + ((threadIdx.x * 16 + i * 4096 + it * 64 + 2048) & 65520));
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      c[(ks & 1) * 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ks], b[ks], c[(ks & 1) * 2], 0, 0, 0);
      c[(ks & 1) * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ks], b[(ks + 1) & 7], c[(ks & 1) * 2 + 1], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = acc;
  for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (s == 12345.678f) *sink = s;
}
template <int NW, int EPI>
int run_pipe(long long* d, float* sink, const char* what) {
  const int iters = 4000;
  hipLaunchKernelGGL((pipe<NW, EPI>), dim3(256), dim3(NW * 64), 0, 0, d, iters, sink);
  long long h;
  CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
  printf("%-64s waves/CU %2d : %6.1f cycles per interval\n", what, NW, (double)h / iters);
  return 0;
}

template <int VARIANT, int NW>
int run(long long* d, float* sink, const char* what) {
  const int iters = 4000;
  hipLaunchKernelGGL((k<VARIANT, NW>), dim3(256), dim3(NW * 64), 0, 0, d, iters, sink);
  long long h;
  CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
  printf("%-64s waves/CU %2d : %6.1f cycles per 16 MFMAs (%.1f per MFMA)\n", what, NW, (double)h / iters, (double)h / iters / 16);
  return 0;
}

int main() {
  long long* d; float* sink;
  CK(hipMalloc(&d, 16)); CK(hipMalloc(&sink, 4));
  run<0, 4>(d, sink, "4 chains, registers only");
  run<4, 4>(d, sink, "2 chains, registers only");
  run<3, 4>(d, sink, "8 accumulators x 2 (W2 pattern)");
  run<1, 4>(d, sink, "4 chains + 8 ds_read_b128 for the next round");
  run<2, 4>(d, sink, "4 chains + 8 ds_read_b128 + barrier");
  run_pipe<4, 0>(d, sink, "pipelined interval: barrier | 8 reads (next) | 16 MFMA (current)");
  run_pipe<4, 1>(d, sink, "pipelined interval + Swish epilogue every other interval");
  run_pipe<8, 0>(d, sink, "pipelined interval: barrier | 8 reads (next) | 16 MFMA (current)");
  run_pipe<8, 1>(d, sink, "pipelined interval + Swish epilogue every other interval");
  run<0, 8>(d, sink, "4 chains, registers only");
  run<2, 8>(d, sink, "4 chains + 8 ds_read_b128 + barrier");
  return 0;
}
