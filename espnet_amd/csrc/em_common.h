// espnet_amd — shared device helpers for the gfx950 (MI355X, CDNA4) kernels.
// Wave = 64 lanes everywhere; no other target is supported (no dual paths, no shims).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/espnet_amd.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define EM_WAVE 64

// ---- MFMA traits: 16x16 output tile, C/D layout col = lane&15, row = (lane>>4)*4 + reg.
// A operand: row = lane&15, k-slice (lane>>4); B operand: col = lane&15, k-slice (lane>>4).
// Both operands use the same (lane-group, element) -> k convention, so the contraction is
// correct for any hardware k numbering.
template <typename T>
struct Mma;

template <>
struct Mma<bf16> {
  static constexpr int K = 32;   // contraction depth of one instruction
  static constexpr int EPL = 8;  // elements per lane per operand
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ frag load(const bf16* p) { return *(const bf16x8*)p; }
};

template <>
struct Mma<float> {
  static constexpr int K = 4;
  static constexpr int EPL = 1;
  typedef float frag;
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ frag load(const float* p) { return *p; }
};

template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }  // v_cvt_pk_bf16_f32, RNE

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }

// sigmoid via the hardware exp2 / reciprocal approximations (v_exp_f32, v_rcp_f32: ~1 ulp each).
// The libm expf + IEEE division cost ~35 VALU instructions per element and made the Swish / GLU
// GEMM epilogues VALU-bound (+5.5 us on a 12 us GEMM); the result differs from the exact form by
// ~2e-7 relative, below the f32 summation-order noise of the GEMM it follows.
__device__ __forceinline__ float sigmoidf_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
// exact (erf) GELU = torch.nn.GELU() default (cgmlp.py:106-108).  erf by Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, below the f32 round-off of the surrounding GEMM) on the hardware exp: libm's
// erff costs ~3x as much in a store-bound epilogue.
__device__ __forceinline__ float geluf_(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- cross-lane moves on the DPP path (VALU rate, ~8 cycles) instead of ds_bpermute (an LDS-pipe round trip of
// 100+ cycles): a chain of __shfl_xor steps is a chain of dependent LDS round trips, and the search kernels' arg-max
// rounds were exactly that (profiles/r03f_search_kernel_stats.csv: pre-beam 30 us, selection 9.7 us).
// ctrl: quad_perm 0x00-0xff, row_ror:n 0x120+n, row_mirror 0x140, row_half_mirror 0x141 (rows = 16 lanes).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
constexpr int DPP_ROR4 = 0x124, DPP_ROR8 = 0x128;
// all-lanes maximum / minimum of a wave: four DPP steps inside the 16-lane rows, then the four row results through
// scalar registers.  Every lane returns the same value.
// max over the lanes l, l ^ 16 (resp. l ^ 32) through gfx950's register swaps: v_permlane16_swap exchanges the odd 16-lane
// rows of one operand with the even rows of the other, v_permlane32_swap the upper half of one with the lower half of the
// other - with both operands = x, every lane ends up holding its own value in one result and its partner's in the other.
// One VALU instruction instead of a ds_bpermute round trip through the LDS crossbar.
struct Pair2 {
  float a, b;  // a lane's own value and that of lane ^ 16 (resp. ^ 32), in an order that depends on the lane
};
__device__ __forceinline__ Pair2 wave_xor16_pair(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return {__uint_as_float(r[0]), __uint_as_float(r[1])};
}
__device__ __forceinline__ Pair2 wave_xor32_pair(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return {__uint_as_float(r[0]), __uint_as_float(r[1])};
}
__device__ __forceinline__ float wave_xor16_max(float x) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float wave_xor32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float wave_allmax_dpp(float v) {
  v = fmaxf(v, dpp_f32<DPP_XOR1>(v));
  v = fmaxf(v, dpp_f32<DPP_XOR2>(v));
  v = fmaxf(v, dpp_f32<DPP_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_f32<DPP_MIRROR>(v));
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ float wave_allsum_dpp(float v) {
  v += dpp_f32<DPP_XOR1>(v);
  v += dpp_f32<DPP_XOR2>(v);
  v += dpp_f32<DPP_HALF_MIRROR>(v);
  v += dpp_f32<DPP_MIRROR>(v);
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (a + b) + (c + d);
}
__device__ __forceinline__ int wave_allmin_dpp(int v) {
  v = min(v, dpp_i32<DPP_XOR1>(v));
  v = min(v, dpp_i32<DPP_XOR2>(v));
  v = min(v, dpp_i32<DPP_HALF_MIRROR>(v));
  v = min(v, dpp_i32<DPP_MIRROR>(v));
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return min(min(a, b), min(c, d));
}

#define EM_CHECK_LAUNCH()                                  \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return EM_ERR_LAUNCH;           \
  } while (0)

static inline int em_cdiv(int a, int b) { return (a + b - 1) / b; }

// Dynamic-LDS opt-in (hipFuncAttributeMaxDynamicSharedMemorySize) of ONE kernel: the attribute is per DEVICE, so the
// high-water mark is kept per device (a per-process flag would leave the second GPU of a process at the 64 KB default and
// its launches would fail).  Raised only when a launch needs more than the device has been given: no runtime API call
// on later launches, which keeps them legal inside a stream capture.  Concurrent first launches may both set it (idempotent).
struct EmLdsCap {
  int bytes[64];
};
static inline int em_raise_lds_cap(const void* fn, size_t bytes, EmLdsCap* cap) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return EM_ERR_LAUNCH;
  if ((int)bytes > __atomic_load_n(&cap->bytes[dev], __ATOMIC_ACQUIRE)) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return EM_ERR_LAUNCH;
    __atomic_store_n(&cap->bytes[dev], (int)bytes, __ATOMIC_RELEASE);
  }
  return EM_OK;
}

// per-launch timing of the MFMA kernel families (csrc/gemm.hip; bench.py's roofline leg)
bool em_prof_begin(void* stream);
void em_prof_end(void* stream, double flops, int tag);

// csrc/encoder.hip: do the rounds of M / 64 row-block workgroups (csrc/ffn_rows.hip) fill enough of this device's CUs?  `flags`:
// the encoder call's flags (EM_ENC_IN_FLIGHT: launches of other streams run beside a half-filled one)
bool em_rows_fill_ok(long M, int flags);

// csrc/decoder.hip (round 6): the source-attention memory of one decoder layer fragment-major (kv [B*T][2d] = K | V rows ->
// kf [B][heads][Tpad/16][dk/32][64][8], vf [B][heads][dk/16][Tpad/32][64][8], zero for keys >= T), and the label step's source
// attention (norm2 + query projection in its prologue, wq fragment-major) reading it: bit for bit em_dec_src_attention_lnq
int em_dec_pack_memory_frag_bf16(const void* kv, int B, int T, int Tpad, int d, int heads, void* kf, void* vf, void* stream);
int em_dec_src_attention_lnq_memfrag_bf16(const float* x, const float* g, const float* be, float eps, const void* wq_frag,
                                          const float* bq, const void* kf, const void* vf, const int32_t* klens, int B, int W, int d,
                                          int heads, int T, int Tpad, void* ctx, void* stream);

// csrc/gemm_mid.hip: mid_gemm on fragment-major bf16 operands (EM_EPI_RESID_F32; frag 1: A and W, 2: W only)
int em_gemm_mid_frag(int epilogue, int frag, const EmGemmArgs* p, void* stream);

// csrc/decoder.hip: self-attention over the union of a beam's ancestors (bf16, d_k = 64, W <= 16 consecutive rows per
// utterance, Lmax <= 512 with the key list inside 64 KiB of LDS); EM_ERR_UNSUPPORTED for other shapes (the caller keeps
// em_dec_self_attention)
int em_dec_self_attention_tree_bf16(const void* qkv, void* kc, void* vc, const int* anc, const int* anc_odd, int n, int d,
                                    int heads, int Lmax, int pos, const int* pos_dev, int W, void* ctx, void* stream);
