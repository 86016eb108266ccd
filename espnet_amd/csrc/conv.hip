// Depthwise Conv1d(k, zero pad (k-1)/2) + eval-mode BatchNorm1d + Swish of the Conformer
// convolution module (espnet2/legacy/nets/pytorch_backend/conformer/convolution.py:72-75).
// The BatchNorm affine (running stats) is folded into the depthwise weights/bias by the caller.
// Bandwidth-bound; channel-last (B, T, d) so every access is coalesced over channels.  Without
// `tlens` no padding mask is applied (the reference lets the padded frames of a batch leak into
// their neighbours; batched parity needs it); with `tlens` frames t >= tlens[b] read as zero, which
// is what the utterance sees when it is decoded alone (asr_inference.py: one utterance per call).
#include "em_common.h"

namespace {

constexpr int TT = 16;  // outputs per workgroup along time
constexpr int KMAX = 31;

// One workgroup = TT output frames x 256 channels of one utterance.  The (TT + KW - 1) x 256 input
// tile is staged in LDS with 16-byte coalesced loads (zero rows outside [0, T)); thread c then
// walks its channel's column once, feeding each value to the (up to KW) outputs it contributes to
// -- every tap weight and every accumulator stays in registers, each input is read from LDS once.
template <typename T, int KW>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ x,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ bias,
                                                     const int* __restrict__ tlens, int Tn, int d,
                                                     T* __restrict__ y) {
  constexpr int HALF = (KW - 1) / 2;
  constexpr int ROWS = TT + KW - 1;
  constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int CPR = 256 / EPC;            // chunks per 256-channel row
  __shared__ __attribute__((aligned(16))) T tile[ROWS][256];
  const int c0 = blockIdx.x * 256, t0 = blockIdx.y * TT, b = blockIdx.z;
  const int tid = threadIdx.x;
  const T* xb = x + (size_t)b * Tn * d;
  const int Tv = tlens ? (tlens[b] < Tn ? tlens[b] : Tn) : Tn;
  constexpr int NLD = (ROWS * CPR + 255) / 256;
  uint4 stage[NLD];
#pragma unroll
  for (int it = 0; it < NLD; ++it) {  // all global loads in flight before the first LDS store
    const int q = tid + it * 256;
    const int r = q / CPR, ch = q - r * CPR;
    const int t = t0 - HALF + r;
    stage[it] = make_uint4(0u, 0u, 0u, 0u);
    if (q < ROWS * CPR && t >= 0 && t < Tv && c0 + ch * EPC < d)
      stage[it] = *(const uint4*)(xb + (size_t)t * d + c0 + ch * EPC);
  }
#pragma unroll
  for (int it = 0; it < NLD; ++it) {
    const int q = tid + it * 256;
    const int r = q / CPR, ch = q - r * CPR;
    if (q < ROWS * CPR) *(uint4*)(&tile[r][ch * EPC]) = stage[it];
  }
  __syncthreads();
  const int c = c0 + tid;
  if (c >= d) return;
  float wk[KW];
#pragma unroll
  for (int k = 0; k < KW; ++k) wk[k] = w[(size_t)k * d + c];  // tap-major: coalesced
  const float bc = bias[c];
  float acc[TT];
#pragma unroll
  for (int o = 0; o < TT; ++o) acc[o] = bc;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const float v = to_f32(tile[r][tid]);
#pragma unroll
    for (int o = 0; o < TT; ++o)
      if (r - o >= 0 && r - o < KW) acc[o] = fmaf(wk[r - o], v, acc[o]);
  }
  T* yb = y + (size_t)b * Tn * d + c;
#pragma unroll
  for (int o = 0; o < TT; ++o)
    if (t0 + o < Tn) yb[(size_t)(t0 + o) * d] = from_f32<T>(swishf_(acc[o]));
}

template <typename T>
int launch_dw(const void* x, const float* w, const float* b, const int* tlens, int B, int Tn, int d,
              int k, void* y, hipStream_t s) {
  if (d % (16 / (int)sizeof(T)) != 0) return EM_ERR_UNSUPPORTED;
  dim3 grid(em_cdiv(d, 256), em_cdiv(Tn, TT), B);
#define EM_DW_CASE(KW)                                                                      \
  case KW:                                                                                  \
    hipLaunchKernelGGL((dwconv_kernel<T, KW>), grid, dim3(256), 0, s, (const T*)x, w, b, tlens, \
                       Tn, d, (T*)y);                                                       \
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
                                  const int32_t* tlens, int32_t B, int32_t T, int32_t d, int32_t k,
                                  void* y, void* stream) {
  if (B <= 0 || T <= 0 || d <= 0 || k > KMAX) return EM_ERR_BAD_ARG;
  if (dtype == EM_F32) return launch_dw<float>(x, w, b, tlens, B, T, d, k, y, (hipStream_t)stream);
  if (dtype == EM_BF16) return launch_dw<bf16>(x, w, b, tlens, B, T, d, k, y, (hipStream_t)stream);
  return EM_ERR_BAD_ARG;
}
