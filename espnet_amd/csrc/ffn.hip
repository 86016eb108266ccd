// Fused macaron feed-forward of the Conformer block (bf16 MFMA, d = 256):
//
//     x += scale * ( W2 . swish( W1 . LN(x) + b1 ) + b2 )
//
// Reference: PositionwiseFeedForward.forward (transformer/positionwise_feed_forward.py:30-32)
// inside EncoderLayer.forward (conformer/encoder_layer.py:108-121 and :160-168: norm_ff(_macaron),
// ff_scale 0.5) with LayerNorm (transformer/layer_norm.py:12-42).
//
// Why fused: at B = 32 the unfused chain LN -> GEMM(w1, Swish) -> GEMM(w2, residual) is bound by
// HBM stores, not by MFMA: the (M, ff) hidden activation (16 MB bf16) is written and re-read, as
// is LN(x) (tools/gemm_bench.py: the w1 GEMM costs 9.5 us before its first useful K-step).
// Here a workgroup owns BM = 64 rows: LN(x) is built once in LDS, the hidden activation lives
// only in registers / LDS one 128-column chunk at a time, and the only HBM traffic is x (read +
// write, f32) and the weights (streamed once per workgroup through L2).
//
// Structure (512 threads = 8 waves, two per SIMD so that one wave's LDS reads / LDS-DMA issue /
// epilogue VALU overlap the other's MFMAs):
//   prologue  each wave normalises 8 rows (f32 statistics, wave shuffles) and writes bf16 LN(x)
//             into LDS as four [64 rows][64 k] tiles, 16-byte chunks XOR-swizzled by (row & 7);
//   per 128-wide chunk of the hidden dimension, eight 16-KiB weight units ([128 rows][64 k]) stream
//             through a 6-unit LDS ring with global_load_lds_dwordx4; up to 5 units stay in flight
//             while one is consumed (counted vmcnt + raw s_barrier, swizzle on the SOURCE address) --
//             with one resident workgroup per CU the L2 latency has to be covered by ring depth:
//       4 x W1 tile [128 n][64 k]: H^T[n][m] += W1 . LN(x)^T   (computed transposed so that a lane
//             holds 4 consecutive hidden indices of one row -> one 8-byte LDS store per fragment)
//       epilogue-1: + b1, Swish, bf16 -> LDS Hs as two [64 m][64 k] A-operand tiles
//       2 x W2 tile [256 n][64 k]: Y[m][n] += Hs . W2^T, accumulated across all chunks in registers
//   epilogue-2  per-wave LDS transpose, x += scale * (Y + b2) with 16-byte loads / stores.
#include <stdlib.h>

#include "em_common.h"

namespace {

constexpr int D = 256;     // model width handled by this kernel
constexpr int FBM = 64;    // rows per workgroup
constexpr int FC = 128;    // hidden columns per chunk
constexpr int ROWB = 128;  // bytes of K per LDS tile row (64 bf16)
constexpr int FFN_MAX_FF = 2048;

typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

__device__ __forceinline__ void glds16(const unsigned char* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(512) void ffn_fused_kernel(float* __restrict__ x,
                                                        const float* __restrict__ ln_g,
                                                        const float* __restrict__ ln_b, float eps,
                                                        const bf16* __restrict__ W1,
                                                        const float* __restrict__ b1,
                                                        const bf16* __restrict__ W2,
                                                        const float* __restrict__ b2, int M, int ff,
                                                        float scale, int dbg) {
  using MM = Mma<bf16>;
  constexpr int XS_BYTES = 4 * FBM * ROWB;  // 32 KiB: LN(x), four k-tiles
  constexpr int HS_BYTES = 2 * FBM * ROWB;  // 16 KiB: hidden chunk, two k-tiles
  constexpr int UNIT = 128 * ROWB;          // 16 KiB ring unit: 128 weight rows x 64 k
  constexpr int NU = 6;                     // ring depth (units)
  // b1 is staged in LDS: an ordinary global load inside the glds pipeline would make hipcc drain
  // the prefetch (vmcnt(0)) at its first use
  constexpr int B1_BYTES = FFN_MAX_FF * 4;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[XS_BYTES + HS_BYTES + NU * UNIT + B1_BYTES];
  unsigned char* Xs = smem;
  unsigned char* Hs = smem + XS_BYTES;
  unsigned char* ring = Hs + HS_BYTES;
  float* b1s = (float*)(ring + NU * UNIT);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * FBM;
  const int nchunks = ff / FC;

  // ---- weight stream.  Unit g (g = 8*chunk + v) lives in ring slot g % NU:
  //   v = 0..3 : W1 rows n = chunk*128 + [0,128), k = v*64 .. +64
  //   v = 4..7 : W2 rows n = ((v-4)&1)*128 + [0,128), k = chunk*128 + ((v-4)>>1)*64 .. +64
  // A glds instruction moves 8 rows x 128 B; every wave loads 16 rows (2 instructions) of every
  // unit; lane l supplies row l>>3, LDS chunk l&7 = global chunk (l&7) ^ (row&7).
  const int ld_row = lane >> 3;
  const int ld_chunk = (lane & 7) ^ ld_row;
  const int total_units = 8 * nchunks;
  const unsigned char* w1_src = (const unsigned char*)(W1 + (size_t)(wave * 16 + ld_row) * D) + ld_chunk * 16;
  const unsigned char* w2_src = (const unsigned char*)(W2 + (size_t)(wave * 16 + ld_row) * ff) + ld_chunk * 16;
  // Workgroups walk the hidden chunks in a rotated order (chunk (c + blockIdx) % nchunks at
  // step c): otherwise all resident workgroups request the same 16 KiB of weights at the same
  // moment and serialise on a handful of L2 channels.
  const int crot = blockIdx.x % nchunks;
  auto issue_unit = [&](int gu) {
    if (dbg & 2) return;
    int c = (gu >> 3) + crot;
    c = c >= nchunks ? c - nchunks : c;
    const int v = gu & 7;
    unsigned char* dst = ring + (gu % NU) * UNIT + wave * 2 * 1024;
    if (v < 4) {
      const unsigned char* src = w1_src + ((size_t)c * FC * D + (size_t)v * 64) * sizeof(bf16);
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16(src + (size_t)i * 8 * D * sizeof(bf16), dst + i * 1024);
    } else {
      const int half = (v - 4) & 1, ks = (v - 4) >> 1;
      const unsigned char* src =
          w2_src + ((size_t)half * 128 * ff + (size_t)c * FC + (size_t)ks * 64) * sizeof(bf16);
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16(src + (size_t)i * 8 * ff * sizeof(bf16), dst + i * 1024);
    }
  };
  // wait until at most `units` of this wave's unit loads (2 glds each) are still in flight
  auto wait_units = [&](int units) {
    if (dbg & 2) return;
    switch (units) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    }
  };

  for (int i = tid; i < ff; i += 512) b1s[i] = b1[i];
  int issued = 0;  // units put in flight so far (wave-uniform)
  for (; issued < NU && issued < total_units; ++issued) issue_unit(issued);  // fill the ring

  // ---- prologue: LN(x) -> Xs (bf16).  Wave w owns rows 8w..8w+7; lane l owns columns 4l..4l+3.
  {
    const float4 g4 = *(const float4*)(ln_g + lane * 4), be4 = *(const float4*)(ln_b + lane * 4);
    const int kt = lane >> 4, chunk = (lane & 15) >> 1, half = lane & 1;
#pragma unroll 4
    for (int rr = 0; rr < 8; ++rr) {
      const int row = wave * 8 + rr;
      int m = m0 + row;
      m = m < M ? m : M - 1;
      float4 v = *(const float4*)(x + (size_t)m * D + lane * 4);
      const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / D);
      v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
      const float var = wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) * (1.0f / D);
      const float rstd = 1.0f / sqrtf(var + eps);
      bf16x4 pk = {(bf16)(v.x * rstd * g4.x + be4.x), (bf16)(v.y * rstd * g4.y + be4.y),
                   (bf16)(v.z * rstd * g4.z + be4.z), (bf16)(v.w * rstd * g4.w + be4.w)};
      *(bf16x4*)(Xs + kt * (FBM * ROWB) + row * ROWB + ((chunk ^ (row & 7)) << 4) + half * 8) = pk;
    }
  }

  const int wr = wave >> 1, wc = wave & 1;  // GEMM-1 wave grid: 4 (n) x 2 (m)
  const int swz = lr & 7;
  f32x4 acc2[4][2];  // Y tile of this wave: 64 rows x 32 cols (cols wave*32..)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 acc1[2][2];  // H^T tile of this wave: 32 n x 32 m

  // One pipeline stage consumes units [first, first + w).  Units < first have been released by
  // the previous stage's trailing barrier, so the ring can be refilled up to unit first + NU - 1;
  // then wait until everything below first + w has landed (for this wave; the barrier extends it
  // to the workgroup).
#define FFN_STAGE(FIRST, WIDTH, COMPUTE)                                          \
  do {                                                                            \
    for (; issued < (FIRST) + NU && issued < total_units; ++issued) issue_unit(issued); \
    wait_units(issued - ((FIRST) + (WIDTH)));                                     \
    __builtin_amdgcn_s_barrier();                                                 \
    COMPUTE;                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                            \
    __builtin_amdgcn_s_barrier();                                                 \
  } while (0)

  auto gemm1_step = [&](int ks_tile, int gu) {
    if (dbg & 1) return;
    const unsigned char* sw = ring + (gu % NU) * UNIT;
    const unsigned char* sx = Xs + ks_tile * (FBM * ROWB);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + lg) ^ swz) << 4;
      MM::frag fw[2], fx[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fw[i] = MM::load((const bf16*)(sw + (wr * 32 + i * 16 + lr) * ROWB + coff));
#pragma unroll
      for (int j = 0; j < 2; ++j) fx[j] = MM::load((const bf16*)(sx + (wc * 32 + j * 16 + lr) * ROWB + coff));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc1[i][j] = MM::mma(fw[i], fx[j], acc1[i][j]);
    }
  };
  auto epilogue1 = [&](int cstep) {
    if (dbg & 4) return;
    int c = cstep + crot;
    c = c >= nchunks ? c - nchunks : c;
    // acc1[i][j][r]: hidden index n = wr*32 + i*16 + lg*4 + r, row m = wc*32 + j*16 + lr
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 bb = *(const float4*)(b1s + c * FC + wr * 32 + i * 16 + lg * 4);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int m = wc * 32 + j * 16 + lr;
        bf16x4 pk = {(bf16)swishf_(acc1[i][j][0] + bb.x), (bf16)swishf_(acc1[i][j][1] + bb.y),
                     (bf16)swishf_(acc1[i][j][2] + bb.z), (bf16)swishf_(acc1[i][j][3] + bb.w)};
        // k_local = wr*32 + i*16 + lg*4: k-tile wr>>1, 16-byte chunk (wr&1)*4 + i*2 + lg/2
        const int chunk = (wr & 1) * 4 + i * 2 + (lg >> 1), half = lg & 1;
        *(bf16x4*)(Hs + (wr >> 1) * (FBM * ROWB) + m * ROWB + ((chunk ^ (m & 7)) << 4) + half * 8) = pk;
        acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto gemm2_step = [&](int ks_tile, int gu) {  // gu: unit of W2 rows 0..127, gu+1: rows 128..255
    if (dbg & 1) return;
    const unsigned char* sw = ring + ((gu + (wave >> 2)) % NU) * UNIT;
    const unsigned char* sh = Hs + ks_tile * (FBM * ROWB);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + lg) ^ swz) << 4;
      MM::frag fh[4], fw[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) fh[i] = MM::load((const bf16*)(sh + (i * 16 + lr) * ROWB + coff));
#pragma unroll
      for (int j = 0; j < 2; ++j) fw[j] = MM::load((const bf16*)(sw + ((wave & 3) * 32 + j * 16 + lr) * ROWB + coff));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc2[i][j] = MM::mma(fh[i], fw[j], acc2[i][j]);
    }
  };

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // Xs written (ordered by the first barrier)
  for (int c = 0; c < nchunks; ++c) {
    const int u0 = 8 * c;
    FFN_STAGE(u0 + 0, 1, gemm1_step(0, u0 + 0));
    FFN_STAGE(u0 + 1, 1, gemm1_step(1, u0 + 1));
    FFN_STAGE(u0 + 2, 1, gemm1_step(2, u0 + 2));
    FFN_STAGE(u0 + 3, 1, (gemm1_step(3, u0 + 3), epilogue1(c)));
    FFN_STAGE(u0 + 4, 2, gemm2_step(0, u0 + 4));
    FFN_STAGE(u0 + 6, 2, gemm2_step(1, u0 + 6));
  }
#undef FFN_STAGE

  // ---- epilogue-2: per-wave LDS transpose ([64 rows][32 cols] f32, 8 KiB per wave, in the ring;
  // 16-column groups XOR-swizzled by (row >> 2) & 1), then x[m][n] += scale * (Y + b2) with float4
  // accesses over 128-byte row segments.
  float* ep = (float*)ring + wave * (64 * 32);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ep[(i * 16 + lg * 4 + r) * 32 + ((j ^ (lg & 1)) << 4) + lr] = acc2[i][j][r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int rsub = lane >> 3, cg = lane & 7;  // 8 rows x 8 float4 per wave-wide access
  const int ncol = wave * 32 + cg * 4;
  const float4 b4 = *(const float4*)(b2 + ncol);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 8 + rsub;
    const float4 v = *(const float4*)(ep + row * 32 + (((cg >> 2) ^ ((row >> 2) & 1)) << 4) + (cg & 3) * 4);
    const int m = m0 + row;
    if (m < M) {
      float4* xp = (float4*)(x + (size_t)m * D + ncol);
      float4 xo = *xp;
      xo.x += scale * (v.x + b4.x); xo.y += scale * (v.y + b4.y);
      xo.z += scale * (v.z + b4.z); xo.w += scale * (v.w + b4.w);
      *xp = xo;
    }
  }
}

}  // namespace

extern "C" int em_ffn_fused_bf16(float* x, const float* ln_g, const float* ln_b, float eps,
                                 const void* w1, const float* b1, const void* w2, const float* b2,
                                 int32_t M, int32_t d, int32_t ff, float scale, void* stream) {
  if (!x || !ln_g || !ln_b || !w1 || !b1 || !w2 || !b2 || M <= 0) return EM_ERR_BAD_ARG;
  if (d != D || ff <= 0 || ff % FC != 0 || ff > FFN_MAX_FF) return EM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(ffn_fused_kernel, dim3(em_cdiv(M, FBM)), dim3(512), 0, (hipStream_t)stream, x,
                     ln_g, ln_b, eps, (const bf16*)w1, b1, (const bf16*)w2, b2, M, ff, scale,
                     getenv("EM_FFN_DBG") ? atoi(getenv("EM_FFN_DBG")) : 0);
  EM_CHECK_LAUNCH();
  return EM_OK;
}
