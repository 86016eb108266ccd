// LayerNorm over the f32 residual stream, one wave per row (transformer/layer_norm.py:12-42,
// eps 1e-12 passed by the caller).  Bandwidth-bound: each row is read once, kept in registers
// (d/64 values per lane), reduced with wave shuffles, and written in the GEMM input dtype.
#include <stdint.h>

#include "em_common.h"

namespace {

template <int NV>
__device__ __forceinline__ void ln_row(float v[NV], const float* __restrict__ g,
                                       const float* __restrict__ b, int lane, int d, float eps) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) s += v[j];
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    float c = v[j] - mean;
    q += c * c;
  }
  const float var = wave_sum(q) / (float)d;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    int c = lane + 64 * j;
    v[j] = (v[j] - mean) * rstd * g[c] + b[c];
  }
}

// MODE 0: out = LN(x; g1,b1)  (optional f32 copy)
// MODE 1: x <- LN(x; g1,b1) in place, out = LN(x; g2,b2)
template <typename T, int NV, int MODE>
__global__ __launch_bounds__(256) void layernorm_kernel(float* __restrict__ x,
                                                        const float* __restrict__ g1,
                                                        const float* __restrict__ b1,
                                                        const float* __restrict__ g2,
                                                        const float* __restrict__ b2, int M, int d,
                                                        float eps, T* __restrict__ out,
                                                        float* __restrict__ out_f32) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float* xr = x + (size_t)row * d;
  float v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = xr[lane + 64 * j];
  ln_row<NV>(v, g1, b1, lane, d, eps);
  if (MODE >= 1) {
#pragma unroll
    for (int j = 0; j < NV; ++j) xr[lane + 64 * j] = v[j];
    if (MODE == 2) return;  // in place only
    ln_row<NV>(v, g2, b2, lane, d, eps);
  }
  T* o = out + (size_t)row * d;
#pragma unroll
  for (int j = 0; j < NV; ++j) o[lane + 64 * j] = from_f32<T>(v[j]);
  if (out_f32) {
    float* of = out_f32 + (size_t)row * d;
#pragma unroll
    for (int j = 0; j < NV; ++j) of[lane + 64 * j] = v[j];
  }
}

// Vectorised form for d % 256 == 0: lane l owns the float4 chunks l, l+64, ... of the row
// (16-byte loads, 8/16-byte stores).
template <int NV4>
__device__ __forceinline__ void ln_row4(float4 v[NV4], const float* __restrict__ g,
                                        const float* __restrict__ b, int lane, int d, float eps) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV4; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV4; ++j) {
    const float a = v[j].x - mean, bq = v[j].y - mean, c = v[j].z - mean, e = v[j].w - mean;
    q += (a * a + bq * bq) + (c * c + e * e);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
  for (int j = 0; j < NV4; ++j) {
    const float4 g4 = *(const float4*)(g + (j * 64 + lane) * 4);
    const float4 b4 = *(const float4*)(b + (j * 64 + lane) * 4);
    v[j].x = (v[j].x - mean) * rstd * g4.x + b4.x;
    v[j].y = (v[j].y - mean) * rstd * g4.y + b4.y;
    v[j].z = (v[j].z - mean) * rstd * g4.z + b4.z;
    v[j].w = (v[j].w - mean) * rstd * g4.w + b4.w;
  }
}

template <typename T, int NV4, int MODE>
__global__ __launch_bounds__(256) void layernorm4_kernel(float* __restrict__ x,
                                                         const float* __restrict__ g1,
                                                         const float* __restrict__ b1,
                                                         const float* __restrict__ g2,
                                                         const float* __restrict__ b2, int M, int d,
                                                         float eps, T* __restrict__ out,
                                                         float* __restrict__ out_f32) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float4* xr = (float4*)(x + (size_t)row * d);
  float4 v[NV4];
#pragma unroll
  for (int j = 0; j < NV4; ++j) v[j] = xr[j * 64 + lane];
  ln_row4<NV4>(v, g1, b1, lane, d, eps);
  if (MODE >= 1) {
#pragma unroll
    for (int j = 0; j < NV4; ++j) xr[j * 64 + lane] = v[j];
    if (MODE == 2) return;  // in place only
    ln_row4<NV4>(v, g2, b2, lane, d, eps);
  }
  T* o = out + (size_t)row * d;
#pragma unroll
  for (int j = 0; j < NV4; ++j) {
    if (sizeof(T) == 2) {
      bf16x4 pk = {(bf16)v[j].x, (bf16)v[j].y, (bf16)v[j].z, (bf16)v[j].w};
      *(bf16x4*)(o + (j * 64 + lane) * 4) = pk;
    } else {
      *(float4*)(o + (j * 64 + lane) * 4) = v[j];
    }
  }
  if (out_f32) {
    float4* of = (float4*)(out_f32 + (size_t)row * d);
#pragma unroll
    for (int j = 0; j < NV4; ++j) of[j * 64 + lane] = v[j];
  }
}

template <typename T, int MODE>
int launch_ln(float* x, const float* g1, const float* b1, const float* g2, const float* b2, int M,
              int d, float eps, T* out, float* out_f32, hipStream_t s) {
  dim3 grid(em_cdiv(M, 4)), block(256);
  if (d % 256 == 0 && d <= 1024) {
#define EM_LN4_CASE(NV4)                                                                          \
  case NV4:                                                                                       \
    hipLaunchKernelGGL((layernorm4_kernel<T, NV4, MODE>), grid, block, 0, s, x, g1, b1, g2, b2, M, \
                       d, eps, out, out_f32);                                                     \
    break;
    switch (d / 256) { EM_LN4_CASE(1) EM_LN4_CASE(2) EM_LN4_CASE(3) EM_LN4_CASE(4) }
#undef EM_LN4_CASE
    EM_CHECK_LAUNCH();
    return EM_OK;
  }
#define EM_LN_CASE(NV)                                                                          \
  case NV:                                                                                      \
    hipLaunchKernelGGL((layernorm_kernel<T, NV, MODE>), grid, block, 0, s, x, g1, b1, g2, b2, M, \
                       d, eps, out, out_f32);                                                   \
    break;
  switch (d / 64) {
    EM_LN_CASE(1) EM_LN_CASE(2) EM_LN_CASE(4) EM_LN_CASE(6) EM_LN_CASE(8) EM_LN_CASE(12)
    EM_LN_CASE(16)
    default: return EM_ERR_UNSUPPORTED;
  }
#undef EM_LN_CASE
  EM_CHECK_LAUNCH();
  return EM_OK;
}


// LayerNorm of an act-dtype matrix with arbitrary row strides (the gate half of the cgMLP hidden
// activation, cgmlp.py:63-64: channels [n, 2n) of a [M][2n] matrix).  One wave per row, 16-byte
// loads / stores (8 bf16 or 4 f32 per lane per access), values in registers, two-pass statistics.
constexpr int NVA = 48;  // values per lane: n <= 64 * NVA
template <typename T>
__global__ __launch_bounds__(256) void layernorm_act_kernel(const T* __restrict__ x, int ldx,
                                                            const float* __restrict__ g,
                                                            const float* __restrict__ b, int M, int n,
                                                            float eps, T* __restrict__ out, int ldo) {
  constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int NJ = NVA / EPC;             // chunks per lane
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const T* xr = x + (size_t)row * ldx;
  const int nchunk = n / EPC;
  float v[NJ][EPC];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c < nchunk) {
      const uint4 raw = *(const uint4*)(xr + (size_t)c * EPC);
      const T* e = (const T*)&raw;
#pragma unroll
      for (int k = 0; k < EPC; ++k) {
        v[j][k] = to_f32(e[k]);
        s += v[j][k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < EPC; ++k) v[j][k] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)n;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    if (lane + 64 * j < nchunk) {
#pragma unroll
      for (int k = 0; k < EPC; ++k) {
        const float c = v[j][k] - mean;
        q += c * c;
      }
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)n + eps);
  T* o = out + (size_t)row * ldo;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c >= nchunk) continue;
    __attribute__((aligned(16))) T r[EPC];
#pragma unroll
    for (int k = 0; k < EPC; ++k) {
      const int ch = c * EPC + k;
      r[k] = from_f32<T>((v[j][k] - mean) * rstd * g[ch] + b[ch]);
    }
    *(uint4*)(o + (size_t)c * EPC) = *(const uint4*)r;
  }
}

}  // namespace

extern "C" int em_layernorm(int dtype, const float* x, const float* g, const float* b, int32_t M,
                            int32_t d, float eps, void* out, float* out_f32, void* stream) {
  if (M <= 0 || d <= 0 || d % 64 != 0) return d % 64 ? EM_ERR_UNSUPPORTED : EM_ERR_BAD_ARG;
  if (dtype == EM_F32)
    return launch_ln<float, 0>((float*)x, g, b, nullptr, nullptr, M, d, eps, (float*)out, out_f32,
                               (hipStream_t)stream);
  if (dtype == EM_BF16)
    return launch_ln<bf16, 0>((float*)x, g, b, nullptr, nullptr, M, d, eps, (bf16*)out, out_f32,
                              (hipStream_t)stream);
  return EM_ERR_BAD_ARG;
}

extern "C" int em_layernorm2(int dtype, float* x, const float* g1, const float* b1,
                             const float* g2, const float* b2, int32_t M, int32_t d, float eps,
                             void* out, float* out_f32, void* stream) {
  if (M <= 0 || d <= 0 || d % 64 != 0) return d % 64 ? EM_ERR_UNSUPPORTED : EM_ERR_BAD_ARG;
  if (dtype == EM_F32)
    return launch_ln<float, 1>(x, g1, b1, g2, b2, M, d, eps, (float*)out, out_f32,
                               (hipStream_t)stream);
  if (dtype == EM_BF16)
    return launch_ln<bf16, 1>(x, g1, b1, g2, b2, M, d, eps, (bf16*)out, out_f32,
                              (hipStream_t)stream);
  return EM_ERR_BAD_ARG;
}

extern "C" int em_layernorm_inplace_f32(float* x, const float* g, const float* b, int32_t M,
                                        int32_t d, float eps, void* stream) {
  if (M <= 0 || d <= 0 || d % 64 != 0) return d % 64 ? EM_ERR_UNSUPPORTED : EM_ERR_BAD_ARG;
  return launch_ln<float, 2>(x, g, b, nullptr, nullptr, M, d, eps, (float*)nullptr, nullptr,
                             (hipStream_t)stream);
}

// (mean, 1/sqrt(var + eps)) of every row of an act-dtype matrix with a row stride: the statistics half of
// layernorm_act_kernel, for consumers that normalise on load (conv.hip LNIN).
template <typename T>
__global__ __launch_bounds__(256) void row_stats_kernel(const T* __restrict__ x, int ldx, int M, int n, float eps,
                                                        float2* __restrict__ stats) {
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int NJ = NVA / EPC;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const T* xr = x + (size_t)row * ldx;
  const int nchunk = n / EPC;
  float v[NJ][EPC];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c < nchunk) {
      const uint4 raw = *(const uint4*)(xr + (size_t)c * EPC);
      const T* e = (const T*)&raw;
#pragma unroll
      for (int k = 0; k < EPC; ++k) {
        v[j][k] = to_f32(e[k]);
        s += v[j][k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < EPC; ++k) v[j][k] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)n;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    if (lane + 64 * j < nchunk) {
#pragma unroll
      for (int k = 0; k < EPC; ++k) {
        const float c = v[j][k] - mean;
        q += c * c;
      }
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)n + eps);
  if (lane == 0) stats[row] = make_float2(mean, rstd);
}

extern "C" int em_row_stats(int dtype, const void* x, int32_t ldx, int32_t M, int32_t n, float eps, float* stats,
                            void* stream) {
  if (!x || !stats || M <= 0 || n <= 0 || ldx < n) return EM_ERR_BAD_ARG;
  if (n > 64 * NVA) return EM_ERR_UNSUPPORTED;
  const int epc = dtype == EM_BF16 ? 8 : 4;
  if (n % epc || ldx % epc || ((uintptr_t)x & 15)) return EM_ERR_UNSUPPORTED;
  dim3 grid(em_cdiv(M, 4));
  if (dtype == EM_F32)
    hipLaunchKernelGGL(row_stats_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, M, n,
                       eps, (float2*)stats);
  else if (dtype == EM_BF16)
    hipLaunchKernelGGL(row_stats_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, M, n,
                       eps, (float2*)stats);
  else
    return EM_ERR_BAD_ARG;
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_layernorm_act(int dtype, const void* x, int32_t ldx, const float* g, const float* b,
                                int32_t M, int32_t n, float eps, void* out, int32_t ldo, void* stream) {
  if (!x || !g || !b || !out || M <= 0 || n <= 0 || ldx < n || ldo < n) return EM_ERR_BAD_ARG;
  if (n > 64 * NVA) return EM_ERR_UNSUPPORTED;
  const int epc = dtype == EM_BF16 ? 8 : 4;  // 16-byte accesses: channel counts, strides and bases aligned
  if (n % epc || ldx % epc || ldo % epc || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return EM_ERR_UNSUPPORTED;
  dim3 grid(em_cdiv(M, 4));
  if (dtype == EM_F32)
    hipLaunchKernelGGL(layernorm_act_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       ldx, g, b, M, n, eps, (float*)out, ldo);
  else if (dtype == EM_BF16)
    hipLaunchKernelGGL(layernorm_act_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx,
                       g, b, M, n, eps, (bf16*)out, ldo);
  else
    return EM_ERR_BAD_ARG;
  EM_CHECK_LAUNCH();
  return EM_OK;
}
