// Depthwise Conv1d(k, zero pad (k-1)/2) + eval-mode BatchNorm1d + Swish of the Conformer
// convolution module (espnet2/legacy/nets/pytorch_backend/conformer/convolution.py:72-75).
// The BatchNorm affine (running stats) is folded into the depthwise weights/bias by the caller.
// Bandwidth-bound; channel-last (B, T, d) so every access is coalesced over channels.  No padding
// mask is applied (the reference lets padded frames leak into their neighbours; parity needs it).
#include "em_common.h"

namespace {

constexpr int TT = 16;   // outputs per thread along time
constexpr int KMAX = 31;

template <typename T, int KW>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ x,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ bias, int Tn, int d,
                                                     T* __restrict__ y) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  const int t0 = blockIdx.y * TT, b = blockIdx.z;
  constexpr int HALF = (KW - 1) / 2;
  float wk[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) wk[k] = w[c * KW + k];
  float win[TT + KW - 1];
  const T* xb = x + (size_t)b * Tn * d + c;
#pragma unroll
  for (int i = 0; i < TT + KW - 1; ++i) {
    int t = t0 - HALF + i;
    win[i] = (t >= 0 && t < Tn) ? to_f32(xb[(size_t)t * d]) : 0.f;
  }
  const float bc = bias[c];
  T* yb = y + (size_t)b * Tn * d + c;
#pragma unroll
  for (int o = 0; o < TT; ++o) {
    if (t0 + o >= Tn) break;
    float acc = bc;
#pragma unroll
    for (int k = 0; k < KW; ++k) acc = fmaf(wk[k], win[o + k], acc);
    yb[(size_t)(t0 + o) * d] = from_f32<T>(swishf_(acc));
  }
}

template <typename T>
int launch_dw(const void* x, const float* w, const float* b, int B, int Tn, int d, int k, void* y,
              hipStream_t s) {
  dim3 grid(em_cdiv(d, 256), em_cdiv(Tn, TT), B);
#define EM_DW_CASE(KW)                                                                      \
  case KW:                                                                                  \
    hipLaunchKernelGGL((dwconv_kernel<T, KW>), grid, dim3(256), 0, s, (const T*)x, w, b, Tn, \
                       d, (T*)y);                                                           \
    break;
  switch (k) {
    EM_DW_CASE(3) EM_DW_CASE(7) EM_DW_CASE(15) EM_DW_CASE(31)
    default: return EM_ERR_UNSUPPORTED;
  }
#undef EM_DW_CASE
  EM_CHECK_LAUNCH();
  return EM_OK;
}

}  // namespace

extern "C" int em_dwconv_bn_swish(int dtype, const void* x, const float* w, const float* b,
                                  int32_t B, int32_t T, int32_t d, int32_t k, void* y,
                                  void* stream) {
  if (B <= 0 || T <= 0 || d <= 0 || k > KMAX) return EM_ERR_BAD_ARG;
  if (dtype == EM_F32) return launch_dw<float>(x, w, b, B, T, d, k, y, (hipStream_t)stream);
  if (dtype == EM_BF16) return launch_dw<bf16>(x, w, b, B, T, d, k, y, (hipStream_t)stream);
  return EM_ERR_BAD_ARG;
}
