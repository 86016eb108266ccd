// Mid-size GEMM: C[M][N] = A[M][K] * W[N][K]^T for the row counts of a beam-search label step
// (M = B * beam = 160 ... 640) against a few hundred output columns, same fused epilogues as gemm.hip.
//
// Why a third GEMM kernel.  At M = 160, N = 512 the tiled kernel (gemm.hip, 64 x 128 tiles through an LDS
// pipeline with ONE K-step in flight) is 12 workgroups on a 256-CU chip, each walking its whole K extent as a
// chain of dependent L2 round trips: 5.6 us at K = 512 and 17-25 us at K = 2048, 18 such launches per label step
// = 171 of the step's 541 us (profiles/r03f_search_kernel_stats.csv).  The skinny kernel (gemm_skinny.hip, all rows
// x 16 columns per workgroup) re-streams all of A per workgroup and loses beyond M = 48.  Here
//   * a workgroup owns a 32 x 32 output tile (BMT x BNT fragments of 16 x 16) and the WHOLE K extent: the grid is
//     ceil(M / 32) x ceil(N / 32) workgroups (80 at 160 x 512) and there is no cross-workgroup reduction - results are
//     deterministic, no atomics, no fences;
//   * inside the workgroup K is split into 8 contiguous slices, one per wave; a wave requests ALL fragments of its
//     slice (16 bytes per lane straight from global memory into MFMA operand registers: nothing is reused, so no
//     LDS staging) before its first MFMA - one memory round trip per workgroup instead of K / 64 dependent ones;
//   * the 8 partial tiles meet once in LDS and the first BMT * BNT waves finish one fragment each.
// f32 (the parity mode) runs the same kernel: a 16-byte operand load then carries four consecutive k of a row, used
// as element s of four v_mfma_f32_16x16x4_f32 steps (A and W use the same k assignment, so any assignment contracts
// correctly).
#include <stdlib.h>

#include "em_common.h"
#include "switches.h"

namespace {

template <typename T>
struct Frag16;  // 16 bytes per lane of a 16-row operand: KS contraction indices per wave-wide load
template <>
struct Frag16<bf16> {
  static constexpr int KS = 32, EPL = 8;
  bf16x8 v;
  __device__ __forceinline__ void load(const bf16* p) { v = *(const bf16x8*)p; }
  __device__ __forceinline__ void keep(bool live) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 r = __builtin_bit_cast(u32x4, v);
    r = live ? r : (u32x4){0u, 0u, 0u, 0u};
    v = __builtin_bit_cast(bf16x8, r);
  }
  static __device__ __forceinline__ f32x4 mma(const Frag16& a, const Frag16& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
  }
};
template <>
struct Frag16<float> {
  static constexpr int KS = 16, EPL = 4;
  float4 v;
  __device__ __forceinline__ void load(const float* p) { v = *(const float4*)p; }
  __device__ __forceinline__ void keep(bool live) { v = live ? v : make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ f32x4 mma(const Frag16& a, const Frag16& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v.x, b.v.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v.y, b.v.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v.z, b.v.z, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a.v.w, b.v.w, c, 0, 0, 0);
  }
};

constexpr int NW = 8;  // waves per workgroup = K slices

// FRAG (round 6, bf16): both operands FRAGMENT-MAJOR - a [R][K] matrix as [R / 16][K / 32][lane][8 elements], the 16 rows x 32 k
// of one MFMA operand 1 KiB contiguous (espnet_amd.lib.pack_frag16; csrc/dec_ffn.hip writes its hidden activation that way) -
// so a wave-wide operand load is ONE contiguous KiB instead of 16 rows x 64 bytes: the K = 2048 projection of the decoder's
// feed-forward at 640 rows, 384 KiB of operands per workgroup, was 14.4 us from row-major operands.
template <typename T, int EPI, int BMT, int BNT, int U, int FRAG = 0>  // U: steps of a slice requested together; FRAG: 1 both operands fragment-major, 2 W only
__global__ __launch_bounds__(64 * NW) void mid_gemm_kernel(const T* __restrict__ A, const T* __restrict__ W,
                                                           void* __restrict__ Cv, const float* __restrict__ bias,
                                                           int M, int N, int K, int lda, int ldc, float scale) {
  using F = Frag16<T>;
  __shared__ f32x4 red[NW][BMT * BNT][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.y * 16 * BMT, n0 = blockIdx.x * 16 * BNT;
  const int nsteps = K / F::KS;
  const int per = (nsteps + NW - 1) / NW;
  const int s_lo = wave * per;
  const int s_hi = s_lo + per < nsteps ? s_lo + per : nsteps;
  // operand rows (clamped: rows / columns past the edge are computed from valid memory and never stored)
  const T* arow[BMT];
  const T* wrow[BNT];
#pragma unroll
  for (int i = 0; i < BMT; ++i) {
    if constexpr (FRAG == 1) {  // (M, N multiples of 16: whole fragments; a tile past the edge repeats the last fragment)
      int rf = (m0 >> 4) + i;
      rf = rf < (M >> 4) ? rf : (M >> 4) - 1;
      arow[i] = A + ((size_t)rf * nsteps * 64 + lane) * F::EPL;
    } else {
      int m = m0 + i * 16 + lr;
      m = m < M ? m : M - 1;
      arow[i] = A + (size_t)m * lda + lg * F::EPL;
    }
  }
#pragma unroll
  for (int j = 0; j < BNT; ++j) {
    if constexpr (FRAG != 0) {
      int cf = (n0 >> 4) + j;
      cf = cf < (N >> 4) ? cf : (N >> 4) - 1;
      wrow[j] = W + ((size_t)cf * nsteps * 64 + lane) * F::EPL;
    } else {
      int n = n0 + j * 16 + lr;
      n = n < N ? n : N - 1;
      wrow[j] = W + (size_t)n * K + lg * F::EPL;
    }
  }
  constexpr int ASTRIDE = FRAG == 1 ? 64 * F::EPL : F::KS;  // elements between consecutive k-steps of an operand
  constexpr int WSTRIDE = FRAG != 0 ? 64 * F::EPL : F::KS;
  f32x4 acc[BMT][BNT];
#pragma unroll
  for (int i = 0; i < BMT; ++i)
#pragma unroll
    for (int j = 0; j < BNT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // The fragment this wave finishes (waves >= BMT * BNT repeat the last one and store nothing).  Its residual values
  // are requested WITH the operands, not after the reduction, and everything the epilogue touches goes through one
  // raw buffer resource: an element past the edge gets the offset ~0, which the hardware drops (stores) or answers
  // with 0 (loads).  Under `if (m < M)` branches hipcc put an s_waitcnt vmcnt(0) in front of every single store - four
  // dependent store round trips at the end of every workgroup (tools/isa_waits.py).
  const bool finisher = wave < BMT * BNT;
  const int fw_ = finisher ? wave : BMT * BNT - 1;
  const int fi = fw_ / BNT, fj = fw_ % BNT;
  const int ncol = n0 + fj * 16 + lr;
  constexpr bool OUT_ACT = (EPI == EM_EPI_STORE || EPI == EM_EPI_SWISH || EPI == EM_EPI_RELU);
  constexpr unsigned ES = OUT_ACT ? sizeof(T) : 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Cv, 0, (int)(unsigned)((size_t)M * ldc * ES), 0x00020000);
  unsigned coff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + fi * 16 + lg * 4 + r;
    coff[r] = (finisher && m < M && ncol < N) ? (unsigned)(((size_t)m * ldc + ncol) * ES) : 0xffffffffu;
  }
  float cres[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == EM_EPI_RESID_F32) {
#pragma unroll
    for (int r = 0; r < 4; ++r) cres[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, coff[r], 0, 0));
  }
  const float b = bias ? bias[ncol < N ? ncol : N - 1] : 0.f;
  for (int s0 = s_lo; s0 < s_hi; s0 += U) {
    F fa[U][BMT], fw[U][BNT];
    // ALL requests of the batch, then all its MFMAs.  Nothing here is conditional: steps past the slice repeat its
    // last step and contribute through a zeroed A operand.  (With `if (step < s_hi)` around the MFMAs hipcc sank every
    // step's loads into that step's branch: load, s_waitcnt vmcnt(0), MFMA, eight times over - the dependent chain
    // this kernel exists to avoid; tools/isa_waits.py.)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = s0 + u < s_hi ? s0 + u : s_hi - 1;
#pragma unroll
      for (int j = 0; j < BNT; ++j) fw[u][j].load(wrow[j] + (size_t)s * WSTRIDE);
#pragma unroll
      for (int i = 0; i < BMT; ++i) fa[u][i].load(arow[i] + (size_t)s * ASTRIDE);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool live = s0 + u < s_hi;
#pragma unroll
      for (int i = 0; i < BMT; ++i) {
        fa[u][i].keep(live);
#pragma unroll
        for (int j = 0; j < BNT; ++j) acc[i][j] = F::mma(fa[u][i], fw[u][j], acc[i][j]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < BMT; ++i)
#pragma unroll
    for (int j = 0; j < BNT; ++j) red[wave][i * BNT + j][lane] = acc[i][j];
  __syncthreads();
  if (!finisher) return;
  f32x4 v = red[0][wave][lane];
#pragma unroll
  for (int w = 1; w < NW; ++w) v += red[w][wave][lane];  // fixed order: deterministic
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = v[r] + b;
    if (EPI == EM_EPI_SWISH) x = swishf_(x);
    if (EPI == EM_EPI_RELU) x = fmaxf(x, 0.f);
    if (EPI == EM_EPI_RESID_F32) x = cres[r] + scale * x;
    if (EPI == EM_EPI_SCALE_F32) x = scale * x;
    if constexpr (OUT_ACT && sizeof(T) == 2) {
      const bf16 h = (bf16)x;
      __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), rs, coff[r], 0, 0);
    } else {
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), rs, coff[r], 0, 0);
    }
  }
}

template <typename T, int EPI, int BMT, int BNT, int FRAG = 0>
int launch_mid_tile(const EmGemmArgs* p, hipStream_t s) {
  static_assert(BMT * BNT <= NW, "one finishing wave per fragment");
  dim3 grid(em_cdiv(p->N, 16 * BNT), em_cdiv(p->M, 16 * BMT));
  const int per = em_cdiv(p->K / Frag16<T>::KS, NW);  // steps per wave: K = 512 bf16 -> 2, K = 2048 -> 8
#define EM_MID_LAUNCH(UU)                                                                                              \
  hipLaunchKernelGGL((mid_gemm_kernel<T, EPI, BMT, BNT, UU, FRAG>), grid, dim3(64 * NW), 0, s, (const T*)p->A, (const T*)p->W, \
                     p->C, p->bias, p->M, p->N, p->K, p->lda, p->ldc, p->scale)
  if (per <= 2) EM_MID_LAUNCH(2);
  else if (per <= 4) EM_MID_LAUNCH(4);
  else EM_MID_LAUNCH(8);
#undef EM_MID_LAUNCH
  EM_CHECK_LAUNCH();
  return EM_OK;
}

template <typename T, int EPI>
int launch_mid(const EmGemmArgs* p, hipStream_t s) {
  // 16 x 32 tiles while 32 x 32 ones would not even give every CU a workgroup: at 160 x 512 the label step runs 0.383
  // against 0.399 ms with them (profiles/r03h: more workgroups beat fewer operand bytes; 16 x 16 and 32 x 16 do not
  // help further, 16 x 64 and 32 x 64 lose).  Developer A/B switch: ESPNET_AMD_MID_TILE = "11" | "12" | "21" | "22" | "14" | "24".
  const int force = em_sw().mid_tile;
  // Round 5: 32 x 64 tiles once even 32 x 32 ones give every CU more than a workgroup - configs[3]'s per-GPU step (640 rows x
  // 512 columns: 320 tiles of 32 x 32) runs 0.684 against 0.715 ms per label step with them, the 160-row step 0.40 against
  // 0.35 (profiles/r05l_label_step_dispatch_sweep.txt), hence by tile count and not for all.
  // (the residual projections of the label step, N = 512, at 240 / 320 / 480 / 640 rows - 128 / 160 / 240 / 320 tiles of 32 x 32 -
  // run fastest with 16 x 32 / 32 x 32 / 32 x 32 / 32 x 64 tiles: 0.410, 0.464 against 0.492, 0.540 against 0.559, 0.684 against
  // 0.715 ms per label step; same file)
  const long t32 = (long)em_cdiv(p->M, 32) * em_cdiv(p->N, 32);
  const int tile = force ? force : (t32 < 144 ? 12 : (t32 < 288 ? 22 : 24));
  switch (tile) {
    case 11: return launch_mid_tile<T, EPI, 1, 1>(p, s);
    case 12: return launch_mid_tile<T, EPI, 1, 2>(p, s);
    case 21: return launch_mid_tile<T, EPI, 2, 1>(p, s);
    case 14: return launch_mid_tile<T, EPI, 1, 4>(p, s);
    case 24: return launch_mid_tile<T, EPI, 2, 4>(p, s);
    default: return launch_mid_tile<T, EPI, 2, 2>(p, s);
  }
}

template <typename T>
int dispatch_mid(int epi, const EmGemmArgs* p, hipStream_t s) {
  switch (epi) {
    case EM_EPI_STORE: return launch_mid<T, EM_EPI_STORE>(p, s);
    case EM_EPI_SWISH: return launch_mid<T, EM_EPI_SWISH>(p, s);
    case EM_EPI_RELU: return launch_mid<T, EM_EPI_RELU>(p, s);
    case EM_EPI_RESID_F32: return launch_mid<T, EM_EPI_RESID_F32>(p, s);
    case EM_EPI_SCALE_F32: return launch_mid<T, EM_EPI_SCALE_F32>(p, s);
    case EM_EPI_STORE_F32: return launch_mid<T, EM_EPI_STORE_F32>(p, s);
  }
  return EM_ERR_UNSUPPORTED;
}

}  // namespace

// bf16 on fragment-major operands, C row-major f32 += (EM_EPI_RESID_F32): frag 1 = A [M][K] and W [N][K] both fragment-major
// (M, N multiples of 16, K of 32: the decoder feed-forward's second projection, csrc/dec_ffn.hip), frag 2 = W only (A row-major
// with leading dimension lda: the attention out-projections of a label step, whose `ctx` the attention kernels write as rows).
// Tiles by workgroup count as launch_mid above.
template <int FRAG>
int launch_mid_frag(const EmGemmArgs* p, hipStream_t s) {
  const int force = em_sw().mid_tile;
  const long t32 = (long)em_cdiv(p->M, 32) * em_cdiv(p->N, 32);
  const int tile = force ? force : (t32 < 144 ? 12 : (t32 < 288 ? 22 : 24));
  switch (tile) {
    case 12: return launch_mid_tile<bf16, EM_EPI_RESID_F32, 1, 2, FRAG>(p, s);
    case 24: return launch_mid_tile<bf16, EM_EPI_RESID_F32, 2, 4, FRAG>(p, s);
    case 14: return launch_mid_tile<bf16, EM_EPI_RESID_F32, 1, 4, FRAG>(p, s);
    default: return launch_mid_tile<bf16, EM_EPI_RESID_F32, 2, 2, FRAG>(p, s);
  }
}
int em_gemm_mid_frag(int epilogue, int frag, const EmGemmArgs* p, void* stream) {
  if (epilogue != EM_EPI_RESID_F32 || p->N % 16 != 0 || p->K % 32 != 0) return EM_ERR_UNSUPPORTED;
  if ((size_t)p->M * p->ldc * 4 >= 0xffffffc0ull) return EM_ERR_UNSUPPORTED;
  if (frag == 1) return p->M % 16 == 0 ? launch_mid_frag<1>(p, (hipStream_t)stream) : EM_ERR_UNSUPPORTED;
  if (frag == 2) return p->lda % 8 == 0 ? launch_mid_frag<2>(p, (hipStream_t)stream) : EM_ERR_UNSUPPORTED;
  return EM_ERR_BAD_ARG;
}

// Called by em_gemm (gemm.hip) for EM_A_PLAIN launches whose tiled grid would leave most of the chip idle; returns
// EM_ERR_UNSUPPORTED for what it does not implement (the tiled kernel then takes the launch).
int em_gemm_mid(int dtype, int epilogue, const EmGemmArgs* p, void* stream) {
  const int ks = dtype == EM_BF16 ? 32 : 16;
  if (p->K % ks != 0 || p->lda % (16 / (dtype == EM_BF16 ? 2 : 4)) != 0) return EM_ERR_UNSUPPORTED;
  if ((size_t)p->M * p->ldc * 4 >= 0xffffffc0ull) return EM_ERR_UNSUPPORTED;  // 32-bit offsets of the raw-buffer epilogue
  if (dtype == EM_F32) return dispatch_mid<float>(epilogue, p, (hipStream_t)stream);
  if (dtype == EM_BF16) return dispatch_mid<bf16>(epilogue, p, (hipStream_t)stream);
  return EM_ERR_BAD_ARG;
}
