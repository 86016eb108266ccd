// 256 x 256 x 64 tile, eight waves: the large-M bf16 contraction behind the Conv2d(d, d, 3, 2) of
// Conv2dSubsampling (subsampling.py:402-403: implicit GEMM over the channel-last map of the first conv,
// M = B * T2 * F2 = 151 392 rows at the bench shape, N = d = 256, K = 9 d = 2 304) and any other
// bf16 GEMM with N % 256 == 0 and enough rows to fill the chip.
//
// Why a second tile shape.  The 128 x 128 kernel of gemm.hip moves 32 KiB of operands per 64-deep K-step
// for 2 * 128 * 128 * 64 flops = 64 flop per byte; a CU needs 4 096 flop per clock at the bf16 MFMA rate,
// i.e. 64 B/clk of operand ingest - exactly the L2 -> CU port, of which 33-61 B/clk are reached in
// practice (profiles/r03b_block_stamps_fine.txt): that kernel is ingest-bound at <= 0.5 of the MFMA rate
// by construction, and measured 0.34 on the conv2 shape (211 us).  Here a K-step moves 64 KiB for 8.4
// MFLOP = 128 flop per byte (32 B/clk at the MFMA rate), the A operand - with N = 256 = one column of
// tiles - is read exactly once, and two waves per SIMD alternate between fragment reads and MFMAs.
//
// Structure: 512 threads = 4 x 2 waves, wave tile 64 x 128 (4 x 8 accumulator fragments = 128 registers);
// operands global -> LDS by LDS-DMA (global_load_lds_dwordx4, 8 per wave and K-step), double buffered
// (2 x 64 KiB), counted s_waitcnt vmcnt + raw s_barrier as in gemm.hip; LDS rows are 128 B of K,
// XOR-swizzled on the source address; epilogue through a per-wave LDS transpose, rows leaving through a
// raw buffer resource (no per-row branch: gemm.hip's epilogue note).
#include <type_traits>

#include "em_common.h"

namespace {

constexpr int BM = 256, BN = 256, ROWB = 128, BK = 64, NTHREADS = 512;
constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, BUF = A_BYTES + W_BYTES;  // 64 KiB per stage
constexpr int SMEM = 2 * BUF;                                                      // 128 KiB
constexpr int A_LD = BM / 64, W_LD = BN / 64, NLOADS = A_LD + W_LD;               // LDS-DMA instructions per wave and K-step

typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;
__device__ __forceinline__ void glds16(const unsigned char* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

struct ConvGeom256 {
  int T1, F1, T2, F2, d, kw, st;
};

template <int EPI, int AMODE>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm256_kernel(
    const bf16* __restrict__ A, const bf16* __restrict__ W, void* __restrict__ Cv, const float* __restrict__ bias,
    int M, int N, int K, int lda, int ldc, ConvGeom256 g) {
  using MM = Mma<bf16>;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;  // wave tile: rows 64 wr .., columns 128 wc ..
  const int lr = lane & 15, lg = lane >> 4;
  // XCD-aware tile order (gemm.hip): the tiles of one XCD are contiguous in M
  const int n_tiles = N / BN, m_tiles = (M + BM - 1) / BM;
  const int cpx = (n_tiles * m_tiles + 7) >> 3;
  const int q = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
  if (q >= n_tiles * m_tiles) return;
  const int m0 = (q / n_tiles) * BM, n0 = (q % n_tiles) * BN;

  // per-lane source addresses: LDS-DMA instruction i of this wave fills LDS rows (wave * LD + i) * 8 .. + 7; lane l
  // supplies row l >> 3, LDS chunk l & 7, i.e. global chunk (l & 7) ^ (row & 7)
  const int ld_row = lane >> 3;
  const int ld_chunk = (lane & 7) ^ ld_row;
  const unsigned char* a_src[A_LD];
  const unsigned char* w_src[W_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    int m = m0 + (wave * A_LD + i) * 8 + ld_row;
    m = m < M ? m : M - 1;
    size_t base;
    if (AMODE == EM_A_CONV2) {
      const int per_b = g.T2 * g.F2;
      const int bb = m / per_b, rem = m - bb * per_b;
      const int t2 = rem / g.F2, f2 = rem - t2 * g.F2;
      base = ((size_t)(bb * g.T1 + g.st * t2) * g.F1 + g.st * f2) * g.d;
    } else {
      base = (size_t)m * lda;
    }
    a_src[i] = (const unsigned char*)(A + base) + ld_chunk * 16;
  }
#pragma unroll
  for (int i = 0; i < W_LD; ++i) {
    const int n = n0 + (wave * W_LD + i) * 8 + ld_row;
    w_src[i] = (const unsigned char*)(W + (size_t)n * K) + ld_chunk * 16;
  }
  auto issue = [&](int kt, int buf) {
    const int k0 = kt * BK;
    size_t koff = (size_t)k0;
    if (AMODE == EM_A_CONV2) {
      const int qq = k0 / g.d, c0 = k0 - qq * g.d;
      const int t3 = qq / g.kw, f3 = qq - t3 * g.kw;
      koff = (size_t)(t3 * g.F1 + f3) * g.d + c0;
    }
    unsigned char* sa = smem + buf * BUF;
    unsigned char* sw = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) glds16(a_src[i] + koff * 2, sa + (wave * A_LD + i) * 1024);
#pragma unroll
    for (int i = 0; i < W_LD; ++i) glds16(w_src[i] + (size_t)k0 * 2, sw + (wave * W_LD + i) * 1024);
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int swz = lr & 7;
  const int a_row_off = (wr * 64 + lr) * ROWB;
  const int w_row_off = (wc * 128 + lr) * ROWB;

  const int nk = K / BK;
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      issue(kt + 1, cur ^ 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // tile kt has landed for every wave
    const unsigned char* sa = smem + cur * BUF;
    const unsigned char* sw = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = (((ks * 4 + lg) ^ swz) << 4);
      bf16x8 fa[4], fb[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *(const bf16x8*)(sa + a_row_off + i * 16 * ROWB + coff);
#pragma unroll
      for (int j = 0; j < 8; ++j) fb[j] = *(const bf16x8*)(sw + w_row_off + j * 16 * ROWB + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = MM::mma(fa[i], fb[j], acc[i][j]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave is done reading buf[cur]
  }

  // ---- epilogue.  C/D layout: col = lane & 15, row = (lane >> 4) * 4 + r.  Per wave and 64-column half: LDS transpose
  // [64 rows][64 cols] f32 (16 KiB per wave = the 128 KiB of staging LDS), 16-column groups XOR-swizzled by
  // (row >> 2) & 3; then 16-byte reads, bias / activation, 8-byte bf16 (or 16-byte f32) row stores.
  const int wm0 = m0 + wr * 64;
  float* ep = (float*)smem + wave * (64 * 64);
  constexpr bool OUT_F32 = EPI == EM_EPI_STORE_F32;
  constexpr size_t ES = OUT_F32 ? 4 : 2;
  const bool buf_ok = (size_t)M * ldc * ES < ((size_t)1 << 32) - 64;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(Cv, 0, buf_ok ? (int)(unsigned)((size_t)M * ldc * ES) : 0, 0x00020000);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) ep[(i * 16 + lg * 4 + r) * 64 + ((j ^ lg) << 4) + lr] = acc[i][h * 4 + j][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int jg = (lane & 15) >> 2, cin = (lane & 3) * 4;
    float4 vals[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jr = 0; jr < 4; ++jr) vals[i * 4 + jr] = *(const float4*)(ep + (i * 16 + jr * 4 + lg) * 64 + ((jg ^ jr) << 4) + cin);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int ncol = n0 + wc * 128 + h * 64 + (lane & 15) * 4;
    const float4 b4 = bias ? *(const float4*)(bias + ncol) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int qv = 0; qv < 16; ++qv) {
      const int m = wm0 + (qv >> 2) * 16 + (qv & 3) * 4 + lg;
      float4 v = vals[qv];
      v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
      if (EPI == EM_EPI_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      } else if (EPI == EM_EPI_SWISH) {
        v.x = swishf_(v.x); v.y = swishf_(v.y); v.z = swishf_(v.z); v.w = swishf_(v.w);
      }
      const unsigned off = m < M ? (unsigned)(((size_t)m * ldc + ncol) * ES) : 0xffffffffu;
      if constexpr (OUT_F32) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const u32x4 raw = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
        __builtin_amdgcn_raw_buffer_store_b128(raw, rs, off, 0, 0);
      } else {
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        const bf16x4 pk = {(bf16)v.x, (bf16)v.y, (bf16)v.z, (bf16)v.w};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rs, off, 0, 0);
      }
    }
  }
}

template <int EPI, int AMODE>
int launch256(const EmGemmArgs* p, hipStream_t s) {
  static EmLdsCap cap = {};
  if (em_raise_lds_cap((const void*)gemm256_kernel<EPI, AMODE>, SMEM, &cap) != EM_OK) return EM_ERR_LAUNCH;
  ConvGeom256 g{p->T1, p->F1, p->T2, p->F2, p->d, p->conv_k > 0 ? p->conv_k : 3, p->conv_s > 0 ? p->conv_s : 2};
  const int tiles = (p->N / BN) * em_cdiv(p->M, BM);
  hipLaunchKernelGGL((gemm256_kernel<EPI, AMODE>), dim3(8 * em_cdiv(tiles, 8)), dim3(NTHREADS), SMEM, s, (const bf16*)p->A,
                     (const bf16*)p->W, p->C, p->bias, p->M, p->N, p->K, p->lda, p->ldc, g);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

}  // namespace

// bf16 only.  EM_ERR_UNSUPPORTED = "not a shape for this tile": the caller (em_gemm) falls back to gemm.hip.
int em_gemm256_bf16(int epilogue, int a_mode, const EmGemmArgs* p, void* stream) {
  if (p->N % BN != 0 || p->K % BK != 0 || p->ldc % 4 != 0) return EM_ERR_UNSUPPORTED;
  if ((size_t)p->M * p->ldc * 4 >= ((size_t)1 << 32) - 64) return EM_ERR_UNSUPPORTED;
  // fewer than ~1.5 tiles per CU: the 128-row tiles of gemm.hip fill the chip better
  if ((long)(p->N / BN) * em_cdiv(p->M, BM) < 384) return EM_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (a_mode == EM_A_CONV2) {
    if (epilogue == EM_EPI_RELU) return launch256<EM_EPI_RELU, EM_A_CONV2>(p, s);
    return EM_ERR_UNSUPPORTED;
  }
  switch (epilogue) {
    case EM_EPI_STORE: return launch256<EM_EPI_STORE, EM_A_PLAIN>(p, s);
    case EM_EPI_RELU: return launch256<EM_EPI_RELU, EM_A_PLAIN>(p, s);
    case EM_EPI_SWISH: return launch256<EM_EPI_SWISH, EM_A_PLAIN>(p, s);
    case EM_EPI_STORE_F32: return launch256<EM_EPI_STORE_F32, EM_A_PLAIN>(p, s);
  }
  return EM_ERR_UNSUPPORTED;
}
