// Skinny GEMM: C[M][N] = A[M][K] * W[N][K]^T with M <= 48 rows (streaming encoder step: M =
// block_size + 2; single-utterance beam search: M = beam), same fused epilogues as gemm.hip.
// Measured (tools/gemm_bench.py, bf16): M = 42: 4.5-5.4 us at K = 512, 11 us at K = 2048 -- on par
// or slightly ahead of the tiled kernel; for M = 160 every workgroup re-streams all of A (655 KB at
// K = 2048) at the ~30 GB/s a single CU sustains on latency-bound loads and the tiled kernel, which
// shares A across 128 columns through LDS, wins (11 vs 34 us) -- hence the M <= 48 cut-off.  These launches are pure latency: the weights are streamed exactly once and there is
// almost no reuse, so the tiled kernel's LDS pipeline (one K-step in flight, 12 workgroups for
// N = 512) only serialises L2 round trips.  Here
//   * a workgroup owns ALL M rows x 16 (or 32) columns and the whole K extent, so the grid is
//     N/16 workgroups and there is no split-K reduction across workgroups (deterministic);
//   * inside the workgroup K is split across the 4 waves; every wave loads its A / W fragments
//     straight from global memory into MFMA operand registers (16 bytes per lane, no LDS staging:
//     nothing is reused) with the k-loop unrolled so that 4 k-steps of loads are in flight;
//   * the 4 partial accumulators meet in LDS, and the waves share the epilogue tiles.
#include "em_common.h"

namespace {

constexpr int MAXMT = 3;  // 16-row tiles: M <= 48

__device__ __forceinline__ bf16x8 keep_frag(bf16x8 v, bool live) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 r = __builtin_bit_cast(u32x4, v);
  r = live ? r : (u32x4){0u, 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ float keep_frag(float v, bool live) { return live ? v : 0.f; }

// Epilogue of one 16-row tile i held in the standard C/D layout (col = lr, row = lg*4 + r).
template <typename T, int EPI, int BNT>
__device__ __forceinline__ void skinny_epilogue(const f32x4 (&v)[BNT], int i, int n0, int lr, int lg,
                                                void* __restrict__ Cv,
                                                const float (&bpre)[BNT], int M, int N,
                                                int ldc, float scale) {
  if constexpr (EPI == EM_EPI_GLU) {
    // BNT == 2: fragment 0 = 16 value columns, fragment 1 = their 16 gate columns
    const int ncol = n0 + lr;
    const float bv = bpre[0], bg = bpre[BNT - 1];
    const int ocol = n0 / 2 + lr;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(Cv, 0, (int)(unsigned)((size_t)M * ldc * sizeof(T)), 0x00020000);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = i * 16 + lg * 4 + r;
      const float y = (v[0][r] + bv) * sigmoidf_(v[BNT - 1][r] + bg);
      const unsigned o = (m < M && ncol < N) ? (unsigned)(((size_t)m * ldc + ocol) * sizeof(T)) : 0xffffffffu;
      if constexpr (sizeof(T) == 2) {
        const bf16 hv = (bf16)y;
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), rg, o, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rg, o, 0, 0);
      }
    }
    return;
  }
  // Everything goes through one raw buffer resource: an element past the edge gets the offset ~0, which the hardware
  // drops (stores) or answers with 0 (loads).  Under `if (m < M)` branches hipcc put an s_waitcnt vmcnt(0) in front of
  // every single store (and of every residual load): BNT * 4 dependent round trips at the end of a ~5 us kernel
  // (tools/isa_waits.py, round 3).
  constexpr bool OUT_ACT = (EPI == EM_EPI_STORE || EPI == EM_EPI_SWISH || EPI == EM_EPI_RELU);
  constexpr unsigned ES = OUT_ACT ? sizeof(T) : 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(Cv, 0, (int)(unsigned)((size_t)M * ldc * ES), 0x00020000);
  unsigned off[BNT][4];
  float res[BNT][4];
#pragma unroll
  for (int j = 0; j < BNT; ++j) {
    const int ncol = n0 + j * 16 + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = i * 16 + lg * 4 + r;
      off[j][r] = (m < M && ncol < N) ? (unsigned)(((size_t)m * ldc + ncol) * ES) : 0xffffffffu;
      res[j][r] = 0.f;
      if constexpr (EPI == EM_EPI_RESID_F32) res[j][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off[j][r], 0, 0));
    }
  }
#pragma unroll
  for (int j = 0; j < BNT; ++j) {
    const float b = bpre[j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = v[j][r] + b;
      if (EPI == EM_EPI_SWISH) x = swishf_(x);
      if (EPI == EM_EPI_RELU) x = fmaxf(x, 0.f);
      if (EPI == EM_EPI_RESID_F32) x = res[j][r] + scale * x;
      if (EPI == EM_EPI_SCALE_F32) x = scale * x;
      if constexpr (OUT_ACT && sizeof(T) == 2) {
        const bf16 hv = (bf16)x;
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), rs, off[j][r], 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), rs, off[j][r], 0, 0);
      }
    }
  }
}

template <typename T, int EPI, int BNT>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const T* __restrict__ A,
                                                          const T* __restrict__ W,
                                                          void* __restrict__ Cv,
                                                          const float* __restrict__ bias, int M,
                                                          int N, int K, int lda, int ldc,
                                                          float scale) {
  using MM = Mma<T>;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][mt][BNT][64][4]
  const int tid = threadIdx.x, lane = tid & 63;
  // provably wave-uniform: keeps every `wave`-dependent loop bound / branch scalar
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int mt = (M + 15) >> 4;
  const int n0 = blockIdx.x * (16 * BNT);
  f32x4 acc[MAXMT][BNT];
#pragma unroll
  for (int i = 0; i < MAXMT; ++i)
#pragma unroll
    for (int j = 0; j < BNT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // operand row pointers (rows clamped: out-of-range rows / columns are never stored)
  const T* wrow[BNT];
#pragma unroll
  for (int j = 0; j < BNT; ++j) {
    int n = n0 + j * 16 + lr;
    n = n < N ? n : N - 1;
    wrow[j] = W + (size_t)n * K + lg * MM::EPL;
  }
  const T* arow0 = A + lg * MM::EPL;
  const int nsteps = K / MM::K;
  float bpre[BNT];  // the bias of this lane's columns: requested with the operands, used after the reduction
#pragma unroll
  for (int j = 0; j < BNT; ++j) {
    const int nc = n0 + j * 16 + lr;
    bpre[j] = bias ? bias[nc < N ? nc : N - 1] : 0.f;
  }
  // wave w takes k-steps w, w+4, w+8, ...; 4 of its steps are requested back to back, for W and for all MAXMT row tiles,
  // BEFORE the first MFMA.  Nothing in the batch is conditional: steps past the end repeat the last step and contribute
  // through a zeroed A operand, row tiles past mt repeat row M-1 and are never stored.  (With `if (i < mt)` around the A
  // loads and `if (step < nsteps)` around the MFMAs hipcc issued ONE load at a time and waited for it with
  // s_waitcnt vmcnt(0): twelve dependent global round trips per batch instead of one - tools/isa_waits.py, round 3.
  // This kernel is every projection of the streaming encoder step.)
  for (int s0 = wave; s0 < nsteps; s0 += 16) {
    typename MM::frag fw[4][BNT], fa[MAXMT][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = s0 + 4 * u;
      const int so = s < nsteps ? s : nsteps - 1;
#pragma unroll
      for (int j = 0; j < BNT; ++j) fw[u][j] = MM::load(wrow[j] + (size_t)so * MM::K);
#pragma unroll
      for (int i = 0; i < MAXMT; ++i) {
        int m = i * 16 + lr;
        m = m < M ? m : M - 1;
        fa[i][u] = MM::load(arow0 + (size_t)m * lda + (size_t)so * MM::K);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool live = s0 + 4 * u < nsteps;
#pragma unroll
      for (int i = 0; i < MAXMT; ++i) {
        const typename MM::frag a = keep_frag(fa[i][u], live);
#pragma unroll
        for (int j = 0; j < BNT; ++j) acc[i][j] = MM::mma(a, fw[u][j], acc[i][j]);
      }
    }
  }
  // ---- cross-wave reduction through LDS
#pragma unroll
  for (int i = 0; i < MAXMT; ++i)
    if (i < mt) {
#pragma unroll
      for (int j = 0; j < BNT; ++j)
        *(f32x4*)(red + ((((size_t)wave * mt + i) * BNT + j) * 64 + lane) * 4) = acc[i][j];
    }
  __syncthreads();
  // ---- epilogue: wave w finishes row tiles w, w+4, ...  C/D layout: col = lr, row = lg*4 + r
  for (int i = wave; i < mt; i += 4) {
    f32x4 v[BNT];
#pragma unroll
    for (int j = 0; j < BNT; ++j) {
      v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 4; ++w)
        v[j] += *(const f32x4*)(red + ((((size_t)w * mt + i) * BNT + j) * 64 + lane) * 4);
    }
    skinny_epilogue<T, EPI, BNT>(v, i, n0, lr, lg, Cv, bpre, M, N, ldc, scale);
  }
}

template <typename T, int EPI, int BNT>
int launch_skinny(const EmGemmArgs* p, hipStream_t s) {
  const int mt = (p->M + 15) / 16;
  const size_t lds = (size_t)4 * mt * BNT * 64 * 4 * sizeof(float);
  static EmLdsCap cap = {};
  if (lds > 64 * 1024 && em_raise_lds_cap((const void*)skinny_gemm_kernel<T, EPI, BNT>, lds, &cap) != EM_OK) return EM_ERR_LAUNCH;
  hipLaunchKernelGGL((skinny_gemm_kernel<T, EPI, BNT>), dim3(em_cdiv(p->N, 16 * BNT)), dim3(256), lds, s,
                     (const T*)p->A, (const T*)p->W, p->C, p->bias, p->M, p->N, p->K, p->lda, p->ldc,
                     p->scale);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

template <typename T>
int dispatch_skinny(int epi, const EmGemmArgs* p, hipStream_t s) {
  switch (epi) {
    case EM_EPI_STORE: return launch_skinny<T, EM_EPI_STORE, 1>(p, s);
    case EM_EPI_SWISH: return launch_skinny<T, EM_EPI_SWISH, 1>(p, s);
    case EM_EPI_RELU: return launch_skinny<T, EM_EPI_RELU, 1>(p, s);
    case EM_EPI_RESID_F32: return launch_skinny<T, EM_EPI_RESID_F32, 1>(p, s);
    case EM_EPI_SCALE_F32: return launch_skinny<T, EM_EPI_SCALE_F32, 1>(p, s);
    case EM_EPI_STORE_F32: return launch_skinny<T, EM_EPI_STORE_F32, 1>(p, s);
    case EM_EPI_GLU: return launch_skinny<T, EM_EPI_GLU, 2>(p, s);
  }
  return EM_ERR_UNSUPPORTED;
}

}  // namespace

// Called by em_gemm (gemm.hip) for EM_A_PLAIN launches with M <= 48; returns EM_ERR_UNSUPPORTED
// for epilogues it does not implement (the tiled kernel then takes the launch).
int em_gemm_skinny(int dtype, int epilogue, const EmGemmArgs* p, void* stream) {
  if (p->M > 16 * MAXMT) return EM_ERR_UNSUPPORTED;
  if ((size_t)p->M * p->ldc * 4 >= 0xffffffc0ull) return EM_ERR_UNSUPPORTED;  // 32-bit offsets of the raw-buffer epilogue
  if (dtype == EM_F32) return dispatch_skinny<float>(epilogue, p, (hipStream_t)stream);
  if (dtype == EM_BF16) return dispatch_skinny<bf16>(epilogue, p, (hipStream_t)stream);
  return EM_ERR_BAD_ARG;
}
