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
// MODE (EM_DW_*): what happens to conv(x) + bias before it is stored
//   SWISH  y = swish(.)            Conformer conv module (BatchNorm folded into w, b)
//   LINEAR y = .                   plain depthwise conv
//   GATE   y = gate * (.)          cgMLP CSGU with identity gate activation (cgmlp.py:61-79): gate is
//                                  the other half of the hidden activation, read with its own stride
//   SELFRES y = x + (.)            E-Branchformer merge: x_concat + depthwise_conv_fusion(x_concat)
//                                  (e_branchformer_encoder.py:166-170)
// x, gate and y carry their own row strides so halves of wider matrices are used in place.
// LNIN: the input rows are LayerNorm'ed on the way into LDS (cgmlp.py:63-64: x_g = norm(x_g) before the
// conv) from per-row (mean, rstd) computed by row_stats_kernel and the per-channel gain / bias; the
// normalised value is rounded to the operand dtype first, exactly as a stand-alone LayerNorm would have
// stored it, so the fusion changes no bit -- it only saves writing and re-reading the normalised half.
template <typename T, int KW, int MODE, bool LNIN>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ x, int ldx,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ bias,
                                                     const int* __restrict__ tlens, int Tn, int d,
                                                     const T* __restrict__ gate, int ldg,
                                                     T* __restrict__ y, int ldy,
                                                     const float2* __restrict__ stats,
                                                     const float* __restrict__ ln_g,
                                                     const float* __restrict__ ln_b) {
  constexpr int HALF = (KW - 1) / 2;
  constexpr int ROWS = TT + KW - 1;
  constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int CPR = 256 / EPC;            // chunks per 256-channel row
  __shared__ __attribute__((aligned(16))) T tile[ROWS][256];
  const int c0 = blockIdx.x * 256, t0 = blockIdx.y * TT, b = blockIdx.z;
  const int tid = threadIdx.x;
  const T* xb = x + (size_t)b * Tn * ldx;
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
      stage[it] = *(const uint4*)(xb + (size_t)t * ldx + c0 + ch * EPC);
  }
#pragma unroll
  for (int it = 0; it < NLD; ++it) {
    const int q = tid + it * 256;
    const int r = q / CPR, ch = q - r * CPR;
    if (q >= ROWS * CPR) continue;
    if (LNIN) {
      const int t = t0 - HALF + r, cc = c0 + ch * EPC;
      if (t >= 0 && t < Tv && cc < d) {  // rows outside the utterance stay zero: the conv pads LN's output
        const float2 ms = stats[(size_t)b * Tn + t];
        const T* e = (const T*)&stage[it];
        __attribute__((aligned(16))) T o[EPC];
#pragma unroll
        for (int k = 0; k < EPC; ++k) o[k] = from_f32<T>((to_f32(e[k]) - ms.x) * ms.y * ln_g[cc + k] + ln_b[cc + k]);
        stage[it] = *(const uint4*)o;
      }
    }
    *(uint4*)(&tile[r][ch * EPC]) = stage[it];
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
  T* yb = y + (size_t)b * Tn * ldy + c;
  const T* gb = MODE == EM_DW_GATE ? gate + (size_t)b * Tn * ldg + c : nullptr;
#pragma unroll
  for (int o = 0; o < TT; ++o) {
    if (t0 + o >= Tn) continue;
    float r = acc[o];
    if (MODE == EM_DW_SWISH) r = swishf_(r);
    if (MODE == EM_DW_GATE) r *= to_f32(gb[(size_t)(t0 + o) * ldg]);
    if (MODE == EM_DW_SELFRES) r += to_f32(tile[o + HALF][tid]);
    yb[(size_t)(t0 + o) * ldy] = from_f32<T>(r);
  }
}

template <typename T, int MODE, bool LNIN>
int launch_dw(const void* x, int ldx, const float* w, const float* b, const int* tlens, int B, int Tn, int d,
              int k, const void* gate, int ldg, void* y, int ldy, const float* stats, const float* ln_g,
              const float* ln_b, hipStream_t s) {
  constexpr int EPC = 16 / (int)sizeof(T);
  if (d % EPC != 0 || ldx % EPC != 0) return EM_ERR_UNSUPPORTED;  // 16-byte staged loads
  dim3 grid(em_cdiv(d, 256), em_cdiv(Tn, TT), B);
#define EM_DW_CASE(KW)                                                                               \
  case KW:                                                                                           \
    hipLaunchKernelGGL((dwconv_kernel<T, KW, MODE, LNIN>), grid, dim3(256), 0, s, (const T*)x, ldx, w, b, \
                       tlens, Tn, d, (const T*)gate, ldg, (T*)y, ldy, (const float2*)stats, ln_g, ln_b);  \
    break;
  switch (k) {
    EM_DW_CASE(3) EM_DW_CASE(7) EM_DW_CASE(15) EM_DW_CASE(31)
    default: return EM_ERR_UNSUPPORTED;
  }
#undef EM_DW_CASE
  EM_CHECK_LAUNCH();
  return EM_OK;
}

template <typename T>
int dispatch_dw(int mode, const void* x, int ldx, const float* w, const float* b, const int* tlens, int B,
                int Tn, int d, int k, const void* gate, int ldg, void* y, int ldy, const float* stats,
                const float* ln_g, const float* ln_b, hipStream_t s) {
  if (stats) {  // LayerNorm on the way in: the cgMLP gate path
    if (mode != EM_DW_GATE || !gate || ldg < d || !ln_g || !ln_b) return EM_ERR_BAD_ARG;
    return launch_dw<T, EM_DW_GATE, true>(x, ldx, w, b, tlens, B, Tn, d, k, gate, ldg, y, ldy, stats, ln_g, ln_b, s);
  }
  switch (mode) {
    case EM_DW_SWISH:
      return launch_dw<T, EM_DW_SWISH, false>(x, ldx, w, b, tlens, B, Tn, d, k, gate, ldg, y, ldy, nullptr, nullptr, nullptr, s);
    case EM_DW_LINEAR:
      return launch_dw<T, EM_DW_LINEAR, false>(x, ldx, w, b, tlens, B, Tn, d, k, gate, ldg, y, ldy, nullptr, nullptr, nullptr, s);
    case EM_DW_GATE:
      if (!gate || ldg < d) return EM_ERR_BAD_ARG;
      return launch_dw<T, EM_DW_GATE, false>(x, ldx, w, b, tlens, B, Tn, d, k, gate, ldg, y, ldy, nullptr, nullptr, nullptr, s);
    case EM_DW_SELFRES:
      return launch_dw<T, EM_DW_SELFRES, false>(x, ldx, w, b, tlens, B, Tn, d, k, gate, ldg, y, ldy, nullptr, nullptr, nullptr, s);
  }
  return EM_ERR_BAD_ARG;
}

int dwconv_any(int dtype, int mode, const void* x, int ldx, const float* w, const float* b, const int* tlens,
               int B, int T, int d, int k, const void* gate, int ldg, void* y, int ldy, const float* stats,
               const float* ln_g, const float* ln_b, void* stream) {
  if (!x || !w || !b || !y || B <= 0 || T <= 0 || d <= 0 || k > KMAX || ldx < d || ldy < d) return EM_ERR_BAD_ARG;
  if (dtype == EM_F32)
    return dispatch_dw<float>(mode, x, ldx, w, b, tlens, B, T, d, k, gate, ldg, y, ldy, stats, ln_g, ln_b,
                              (hipStream_t)stream);
  if (dtype == EM_BF16)
    return dispatch_dw<bf16>(mode, x, ldx, w, b, tlens, B, T, d, k, gate, ldg, y, ldy, stats, ln_g, ln_b,
                             (hipStream_t)stream);
  return EM_ERR_BAD_ARG;
}

}  // namespace

extern "C" int em_dwconv(int dtype, int mode, const void* x, int32_t ldx, const float* w, const float* b,
                         const int32_t* tlens, int32_t B, int32_t T, int32_t d, int32_t k, const void* gate,
                         int32_t ldg, void* y, int32_t ldy, void* stream) {
  return dwconv_any(dtype, mode, x, ldx, w, b, tlens, B, T, d, k, gate, ldg, y, ldy, nullptr, nullptr, nullptr, stream);
}

extern "C" int em_dwconv_ln_gate(int dtype, const void* x, int32_t ldx, const float* stats, const float* ln_g,
                                 const float* ln_b, const float* w, const float* b, const int32_t* tlens,
                                 int32_t B, int32_t T, int32_t d, int32_t k, const void* gate, int32_t ldg, void* y,
                                 int32_t ldy, void* stream) {
  if (!stats) return EM_ERR_BAD_ARG;
  return dwconv_any(dtype, EM_DW_GATE, x, ldx, w, b, tlens, B, T, d, k, gate, ldg, y, ldy, stats, ln_g, ln_b, stream);
}

extern "C" int em_dwconv_bn_swish(int dtype, const void* x, const float* w, const float* b,
                                  const int32_t* tlens, int32_t B, int32_t T, int32_t d, int32_t k,
                                  void* y, void* stream) {
  return em_dwconv(dtype, EM_DW_SWISH, x, d, w, b, tlens, B, T, d, k, nullptr, 0, y, d, stream);
}
