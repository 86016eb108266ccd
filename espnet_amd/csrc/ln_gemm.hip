// LayerNorm fused into the GEMM that consumes it, for the small-M launches of the decoder / LM step:
//     C[M][N] = epilogue( LN(x[M][K]; g, b, eps) * W[N][K]^T + bias )
// x is the f32 residual stream; K is the model dimension, so one workgroup can normalise its rows
// completely before the contraction.  In the beam-search step (M = B*W rows, ~75 launches of ~5 us,
// each at the launch-latency floor) every pre-norm LayerNorm (decoder_layer.py:96-98,119-121,150-152;
// transformer_decoder.py:226-227; encoder_layer.py of the LM) was its own launch; here it rides in the
// prologue of the projection that follows it.
//
// Workgroup = 32 rows x 64 columns, 4 waves.  Prologue: wave w normalises rows 8w .. 8w+7 (one row at
// a time, K/64 values per lane, two-pass statistics like norm.hip) and writes them in the operand
// dtype to LDS.  Main loop: wave w owns 16 columns and both 16-row tiles; A fragments come from LDS,
// W fragments straight from global memory (each W element is used by one wave only), 4 k-steps of
// loads in flight.  Every workgroup of a column strip re-normalises the same rows: 64 KB of L2 reads
// and ~2 us of ALU instead of a 5 us launch.
#include <stdlib.h>

#include "em_common.h"
#include "switches.h"

namespace {

constexpr int LG_BM = 32, LG_BN = 64;

// sum over the 16 lanes of a row, in every lane, on DPP (hipcc turns __shfl_xor(., 4) and (., 8) into ds_bpermute: four
// dependent LDS round trips per statistic)
__device__ __forceinline__ float row16_allsum(float v) {
  v += dpp_f32<DPP_XOR1>(v);
  v += dpp_f32<DPP_XOR2>(v);
  v += dpp_f32<DPP_HALF_MIRROR>(v);
  v += dpp_f32<DPP_MIRROR>(v);
  return v;
}
constexpr int LG_MAXV = 16;  // K <= 64 * LG_MAXV

// NV = float4 chunks per lane per row (K <= 64 * NV).  All loads of a wave's 8 rows are in flight before
// any reduction: loading row by row serialises the global round trips.
template <typename T, int EPI, int NV, bool EXACT, int RT>  // RT: 16-row tiles per workgroup (2: 32 rows, 1: 16 rows)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void ln_gemm_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ g,
                                                      const float* __restrict__ be, float eps,
                                                      const T* __restrict__ W,
                                                      const float* __restrict__ bias,
                                                      void* __restrict__ Cv, int M, int N, int K,
                                                      int ldc) {
  using MM = Mma<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lg_smem[];
  T* sA = (T*)lg_smem;
  const int LDA = K + 16 / (int)sizeof(T);  // padded row (elements): 16-byte aligned, conflict-free
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * LG_BN, m0 = blockIdx.y * 16 * RT;
  // EXACT: K == 64 * NV, so `j < nv` folds away.  With a runtime nv every row / gamma / beta load sits under a (uniform)
  // branch, hipcc follows each with s_waitcnt vmcnt(0) - its wait counting gives up at a branch - and the prologue
  // is a chain of dependent global round trips (tools/isa_waits.py: 3 before the statistics at K = 512; round 3).
  const int nv = EXACT ? NV : (K >> 6);

  // ---- the first U k-steps of this wave's W rows are requested before anything else: they do not depend
  // on the LayerNorm, so their latency overlaps the x loads and the statistics (one global round trip
  // for the whole kernel when K <= U * MM::K, i.e. K <= 512 in bf16)
  constexpr int U = 16;
  const int nsteps = (EXACT ? 64 * NV : K) / MM::K;
  int n = n0 + wave * 16 + lr;
  n = n < N ? n : N - 1;
  const T* wrow = W + (size_t)n * K + lg * MM::EPL;
  typename MM::frag fw[U];
#pragma unroll
  for (int u = 0; u < U; ++u) fw[u] = MM::load(wrow + (size_t)(u < nsteps ? u : nsteps - 1) * MM::K);
  const int ncol = n0 + wave * 16 + lr;
  const float bv = bias ? bias[ncol < N ? ncol : N - 1] : 0.f;  // requested with the weights, used after the last MFMA
  // ---- prologue: LayerNorm of rows m0 .. m0+31 into LDS.  16 lanes per row (4 rows per wave at a time,
  // 2 passes): float4 loads (256 contiguous bytes per row and instruction), statistics reduced over 16
  // lanes only (xor 1, 2, 4, 8: DPP row operations, no LDS crossbar), 4 operand-dtype values per LDS
  // store.  The first version used one row per wave with 64-lane reductions and cost 6.5 us of the
  // kernel's 11.7 (phase ablation during development; tools/ln_gemm_bench.py times the result).
  {
    const int grp = lane >> 4, li = lane & 15;
    float4 v[RT][NV];
#pragma unroll
    for (int ps = 0; ps < RT; ++ps) {
      int m = m0 + wave * 4 * RT + ps * 4 + grp;
      m = m < M ? m : M - 1;
      const float4* xr = (const float4*)(x + (size_t)m * K);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[ps][j] = j < nv ? xr[j * 16 + li] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 g4[NV], b4[NV];  // requested with the rows: not a second round trip after the statistics
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      g4[j] = j < nv ? *(const float4*)(g + (j * 16 + li) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      b4[j] = j < nv ? *(const float4*)(be + (j * 16 + li) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // every request of the kernel is out before the first reduction: a memory-clobbering (empty) asm is a line no load
    // may be moved across - left alone hipcc sinks the weight prefetch and the second pass's rows behind the first
    // pass's statistics to save registers, and the prologue becomes three dependent round trips
    asm volatile("" ::: "memory");
#pragma unroll
    for (int ps = 0; ps < RT; ++ps) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) s += (v[ps][j].x + v[ps][j].y) + (v[ps][j].z + v[ps][j].w);
      s = row16_allsum(s);
      const float mean = s / (float)K;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (j < nv) {
          const float a = v[ps][j].x - mean, b2 = v[ps][j].y - mean, c = v[ps][j].z - mean, d2 = v[ps][j].w - mean;
          q += (a * a + b2 * b2) + (c * c + d2 * d2);
        }
      q = row16_allsum(q);
      const float rstd = 1.0f / sqrtf(q / (float)K + eps);
      T* dst = sA + (size_t)(wave * 4 * RT + ps * 4 + grp) * LDA;
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (j < nv) {
          const int c0 = (j * 16 + li) * 4;
          __attribute__((aligned(16))) T o[4];
          o[0] = from_f32<T>((v[ps][j].x - mean) * rstd * g4[j].x + b4[j].x);
          o[1] = from_f32<T>((v[ps][j].y - mean) * rstd * g4[j].y + b4[j].y);
          o[2] = from_f32<T>((v[ps][j].z - mean) * rstd * g4[j].z + b4[j].z);
          o[3] = from_f32<T>((v[ps][j].w - mean) * rstd * g4[j].w + b4[j].w);
          if (sizeof(T) == 2) *(uint2*)(dst + c0) = *(const uint2*)o;
          else *(uint4*)(dst + c0) = *(const uint4*)o;
        }
    }
  }
  __syncthreads();

  // ---- contraction: wave w -> columns n0 + 16w .. +15, row tiles 0 and 1
  const T* a0 = sA + (size_t)lr * LDA + lg * MM::EPL;
  const T* a1 = a0 + (size_t)16 * LDA;
  f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < nsteps; s0 += U) {
    if (s0 > 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) fw[u] = MM::load(wrow + (size_t)(s0 + u < nsteps ? s0 + u : nsteps - 1) * MM::K);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (s0 + u < nsteps) {
        acc0 = MM::mma(MM::load(a0 + (size_t)(s0 + u) * MM::K), fw[u], acc0);
        if constexpr (RT == 2) acc1 = MM::mma(MM::load(a1 + (size_t)(s0 + u) * MM::K), fw[u], acc1);
      }
  }

  // ---- epilogue.  C/D layout: col = lr, row = lg*4 + r.  Stores through a raw buffer resource: an element past the
  // edge gets the offset ~0, which the hardware drops.  (Under `if (m < M)` branches hipcc put an s_waitcnt vmcnt(0) in
  // front of every one of the eight stores - eight dependent store round trips at the end of a 7 us kernel - and the
  // bias was loaded HERE, one more round trip after the last MFMA; tools/isa_waits.py, round 3.)
  constexpr unsigned ES = EPI == EM_EPI_STORE_F32 ? 4 : sizeof(T);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Cv, 0, (int)(unsigned)((size_t)M * ldc * ES), 0x00020000);
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const f32x4 a = i ? acc1 : acc0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + i * 16 + lg * 4 + r;
      float y = a[r] + bv;
      if (EPI == EM_EPI_RELU) y = fmaxf(y, 0.f);
      const unsigned off = (m < M && ncol < N) ? (unsigned)(((size_t)m * ldc + ncol) * ES) : 0xffffffffu;
      if constexpr (EPI == EM_EPI_STORE_F32 || sizeof(T) == 4) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rs, off, 0, 0);
      } else {
        const bf16 hv = (bf16)y;
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), rs, off, 0, 0);
      }
    }
  }
}

template <typename T, int EPI, int NV, bool EXACT, int RT>
int launch_ln_gemm_rt(const float* x, const float* g, const float* b, float eps, const void* W, const float* bias,
                      void* C, int M, int N, int K, int ldc, hipStream_t s) {
  const size_t lds = (size_t)16 * RT * (K + 16 / sizeof(T)) * sizeof(T);
  static EmLdsCap cap = {};  // per instantiation and device: raised once (not per launch: launches may be inside a hipGraph capture)
  if (em_raise_lds_cap((const void*)ln_gemm_kernel<T, EPI, NV, EXACT, RT>, 160 * 1024 - 1024, &cap) != EM_OK) return EM_ERR_LAUNCH;
  dim3 grid(em_cdiv(N, LG_BN), em_cdiv(M, 16 * RT));
  hipLaunchKernelGGL((ln_gemm_kernel<T, EPI, NV, EXACT, RT>), grid, dim3(256), lds, s, x, g, b, eps, (const T*)W, bias, C,
                     M, N, K, ldc);
  EM_CHECK_LAUNCH();
  return EM_OK;
}
template <typename T, int EPI, int NV, bool EXACT>
int launch_ln_gemm_nv(const float* x, const float* g, const float* b, float eps, const void* W, const float* bias,
                      void* C, int M, int N, int K, int ldc, hipStream_t s) {
  // 16-row workgroups where 32-row ones would leave most of the chip idle (developer switch: ESPNET_AMD_LNG_RT = 1 | 2)
  const int force = em_sw().lng_rt;
  const bool rt1 = force ? force == 1 : (long)em_cdiv(N, LG_BN) * em_cdiv(M, 32) < 144;  // (160 rows: the q / k / v and source-q projections 6.0 -> 4.8 us, the 2048-wide FFN projection 6.6 -> 7.3 us with 16-row workgroups: profiles/r03h, r03j)
  if (rt1) return launch_ln_gemm_rt<T, EPI, NV, EXACT, 1>(x, g, b, eps, W, bias, C, M, N, K, ldc, s);
  return launch_ln_gemm_rt<T, EPI, NV, EXACT, 2>(x, g, b, eps, W, bias, C, M, N, K, ldc, s);
}

template <typename T, int EPI>
int launch_ln_gemm(const float* x, const float* g, const float* b, float eps, const void* W, const float* bias,
                   void* C, int M, int N, int K, int ldc, hipStream_t s) {
  const int nv = K / 64;
  switch (nv) {  // the model widths in use get branch-free prologues
    case 4: return launch_ln_gemm_nv<T, EPI, 4, true>(x, g, b, eps, W, bias, C, M, N, K, ldc, s);
    case 8: return launch_ln_gemm_nv<T, EPI, 8, true>(x, g, b, eps, W, bias, C, M, N, K, ldc, s);
    case 16: return launch_ln_gemm_nv<T, EPI, 16, true>(x, g, b, eps, W, bias, C, M, N, K, ldc, s);
  }
  if (nv <= 1) return launch_ln_gemm_nv<T, EPI, 1, false>(x, g, b, eps, W, bias, C, M, N, K, ldc, s);
  if (nv <= 4) return launch_ln_gemm_nv<T, EPI, 4, false>(x, g, b, eps, W, bias, C, M, N, K, ldc, s);
  if (nv <= 8) return launch_ln_gemm_nv<T, EPI, 8, false>(x, g, b, eps, W, bias, C, M, N, K, ldc, s);
  return launch_ln_gemm_nv<T, EPI, LG_MAXV, false>(x, g, b, eps, W, bias, C, M, N, K, ldc, s);
}

template <typename T>
int dispatch_ln_gemm(int epi, const float* x, const float* g, const float* b, float eps, const void* W,
                     const float* bias, void* C, int M, int N, int K, int ldc, hipStream_t s) {
  switch (epi) {
    case EM_EPI_STORE: return launch_ln_gemm<T, EM_EPI_STORE>(x, g, b, eps, W, bias, C, M, N, K, ldc, s);
    case EM_EPI_RELU: return launch_ln_gemm<T, EM_EPI_RELU>(x, g, b, eps, W, bias, C, M, N, K, ldc, s);
    case EM_EPI_STORE_F32: return launch_ln_gemm<T, EM_EPI_STORE_F32>(x, g, b, eps, W, bias, C, M, N, K, ldc, s);
  }
  return EM_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int em_ln_gemm(int dtype, int epilogue, const float* x, const float* ln_g, const float* ln_b,
                          float eps, const void* W, const float* bias, void* C, int32_t M, int32_t N,
                          int32_t K, int32_t ldc, void* stream) {
  if (!x || !ln_g || !ln_b || !W || !C || M <= 0 || N <= 0 || K <= 0 || ldc < N) return EM_ERR_BAD_ARG;
  if (K % 64 != 0 || K > 64 * LG_MAXV) return EM_ERR_UNSUPPORTED;
  if ((size_t)M * ldc * 4 >= 0xffffffc0ull) return EM_ERR_UNSUPPORTED;  // 32-bit offsets of the raw-buffer epilogue
  if (dtype == EM_F32)
    return dispatch_ln_gemm<float>(epilogue, x, ln_g, ln_b, eps, W, bias, C, M, N, K, ldc, (hipStream_t)stream);
  if (dtype == EM_BF16)
    return dispatch_ln_gemm<bf16>(epilogue, x, ln_g, ln_b, eps, W, bias, C, M, N, K, ldc, (hipStream_t)stream);
  return EM_ERR_BAD_ARG;
}
