// Row-block fused Conformer block kernels (bf16 MFMA, d = 256): the "fused per Conformer block"
// path of the encoder.  One workgroup owns 32 consecutive frames of one utterance and carries them
// through every row-local operator of the block; only attention (which mixes frames) stays a kernel
// of its own.  A Conformer block becomes three launches instead of nineteen:
//
//   relpos_attn2        Q, K, V^T (per head)        -> ctx                       (attention2.hip)
//   block<C>            ctx -> linear_out + residual -> norm_conv -> pointwise_conv1 + GLU -> glu
//   block<D|A>          glu -> depthwise conv + BN + Swish -> pointwise_conv2 + residual
//                           -> norm_ff -> FFN + residual -> norm_final
//                       (next block:) norm_ff_macaron -> macaron FFN + residual -> norm_mha
//                           -> Q, K, V^T projections
//   (block<A> opens the stack after the embedding, block<D|FINAL> closes it with after_norm.)
//
// Reference being reproduced: EncoderLayer.forward (conformer/encoder_layer.py:79-179),
// ConvolutionModule.forward (conformer/convolution.py:56-79), PositionwiseFeedForward
// (transformer/positionwise_feed_forward.py:30-32), LayerNorm (transformer/layer_norm.py:12-42),
// the q/k/v projections of RelPositionMultiHeadedAttention.forward_qkv (transformer/attention.py:77-98).
//
// Why this shape.  At B = 32 (M = 7 968 rows) the unfused launch sequence is one wave of workgroups
// per kernel, so every kernel costs its own prologue -> K loop -> epilogue latency chain plus a launch
// boundary, and every activation makes an HBM round trip.  Here the 32-row state never
// leaves the CU: the residual x lives in registers (f32), LayerNorm statistics are wave/LDS
// reductions, the FFN hidden activation exists only as a [32][64] LDS tile, and the only streams are
// the weights (L2-resident, read by all workgroups in near lockstep) through a 4 x 32 KiB
// global_load_lds ring with counted vmcnt and ONE raw s_barrier per 32 KiB unit.  The kernel is
// bound by the per-CU L2 -> LDS ingest (every CU streams all weights of the block, ~3 MiB);
// MFMA time is half of that at the ingest peak of 64 B/clk/CU.
//
// Orientation.  Every GEMM is computed transposed, C^T[n][m] = sum_k W[n][k] A[m][k]: the weight
// fragment is the MFMA A operand, the activation fragment the B operand, so a lane ends up with four
// CONSECUTIVE output columns n of one row m (C/D layout: row = (lane >> 4) * 4 + r, col = lane & 15).
// That makes residual / bias / LayerNorm / bf16 packing 16- or 8-byte operations and lets the
// activation fragments (32 rows x K = 256) stay in registers for a whole GEMM.
//
// Wave roles (8 waves): wave = (nf = wave & 3, mf = wave >> 2).  A "K unit" is 64 weight rows x
// K = 256 (one 16 x 16 output fragment per wave: rows nf*16.., frames mf*16..); a "W2 unit" is all 256
// rows x a 64-deep K slice of the FFN's second matrix (four fragments per wave).  Both are 32 KiB =
// 256 LDS rows of 128 B, XOR-swizzled on the source address like csrc/gemm.hip.
#include "em_common.h"

namespace {

constexpr int D = 256;
constexpr int BM = 32;
constexpr int UNIT = 32768;
constexpr int NSLOT = 4;
constexpr int ABUF_OFF = NSLOT * UNIT;       // 16 KiB: LN(x) as four [32][64] k-tiles; the FFN's H tile aliases tile 0
constexpr int RED_OFF = ABUF_OFF + 16384;    // 1 KiB: LayerNorm partial sums [2][4][32]
constexpr int PAR_OFF = RED_OFF + 1024;      // 2 x 7 KiB: bias / LayerNorm vectors, double buffered per group
constexpr int PAR_FLOATS = EM_BLOCK_PARAM_GROUP;
constexpr int PAR_BYTES = PAR_FLOATS * 4;
constexpr int SMEM_BYTES = PAR_OFF + 2 * PAR_BYTES;  // 162 816 B of the 163 840 B LDS
constexpr int KW = 31, HALF = 15, TROWS = BM + KW - 1;  // depthwise conv: 62-row input tile (lives in ring slot 3)

// LDS-DMA issued from inline asm: global_load_lds_dwordx4 with a wave-uniform 64-bit base (SGPR pair)
// and a per-lane 32-bit byte offset; the LDS destination base goes through M0 (saved / restored: M0 is
// compiler-reserved).  Hidden from hipcc on purpose: told about an LDS-DMA (the builtin), its waitcnt pass
// puts s_waitcnt vmcnt(0) in front of every ds_read it cannot prove disjoint (here: every parameter /
// bias read), which drains the weight ring once per step.  All completion counting is by hand (vmcnt in
// step_begin).  s_nop 4: a base that came through v_readfirstlane needs 5 wait states before a VMEM
// instruction reads it; s_nop 0: M0 write -> LDS-DMA.
__device__ __forceinline__ void glds16(const unsigned char* sbase, int voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
__device__ __forceinline__ const unsigned char* uniform_ptr(const unsigned char* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const unsigned char*)(((unsigned long long)hi << 32) | lo);
}

#define EM_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int MODE>
__global__ __launch_bounds__(512) void block_kernel(const EmBlockArgs a) {
  using MM = Mma<bf16>;
  constexpr bool HAS_C = (MODE & EM_BLOCK_C) != 0, HAS_D = (MODE & EM_BLOCK_D) != 0;
  constexpr bool HAS_A = (MODE & EM_BLOCK_A) != 0, FINAL = (MODE & EM_BLOCK_FINAL) != 0;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const ring = smem;
  unsigned char* const abuf = smem + ABUF_OFF;
  float* const red = (float*)(smem + RED_OFF);
  const float* const par = (const float*)(smem + PAR_OFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int nf = wave & 3, mf = wave >> 2;
  const int b = blockIdx.y, t0 = blockIdx.x * BM, T = a.T;
  const int mloc = mf * 16 + lr;                       // this lane's frame inside the block
  const bool row_ok = t0 + mloc < T;
  const size_t mrow = (size_t)b * T + (row_ok ? t0 + mloc : T - 1);  // clamped: tail rows recompute frame T-1
  const int ncol = nf * 16 + lg * 4;                   // + 64 f: the four consecutive columns of fragment f
  const int swz = lr & 7;

  // valid frames of the utterance for the depthwise conv; read before any LDS-DMA is in flight (the wait for
  // this load would otherwise cover the ring prefetch as well)
  int Tv = T;
  if (HAS_D && a.tlens) Tv = a.tlens[b] < T ? a.tlens[b] : T;

  // ---- weight stream ------------------------------------------------------------------------
  const int nch = a.ff >> 6;  // 64-wide chunks of the FFN hidden dimension
  const int nffn = 2 * nch;
  const int total = HAS_C ? 12 : (HAS_D ? 4 + nffn : 0) + (HAS_A ? nffn + 12 : 0);
  // glds instruction i of this wave fills LDS rows R = 32 wave + 8 i + (lane >> 3) of the unit, lane
  // l supplying the 16-byte chunk (l & 7) ^ (R & 7) of that row's 128 B.
  //   K unit : LDS row R = kt * 64 + n  (kt = 64-deep K tile, n = weight row inside the unit)
  //   W2 unit: LDS row R = n            (256 weight rows, one 64-deep K slice)
  int kofs[4], w2ofs[4];
  {
    const int gc = (lane & 7) ^ (lane >> 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int R = 32 * wave + 8 * i + (lane >> 3);
      kofs[i] = (R & 63) * (D * 2) + (R >> 6) * 128 + gc * 16;
      w2ofs[i] = R * (a.ff * 2) + gc * 16;
    }
  }
  auto issue_unit = [&](int g) {
    const unsigned char* base;
    bool w2 = false;
    int u = g;
    if (HAS_C) {
      base = u < 4 ? (const unsigned char*)a.wout + u * UNIT : (const unsigned char*)a.pw1f + (u - 4) * UNIT;
    } else {
      bool found = false;
      base = nullptr;
      if (HAS_D) {
        if (u < 4) {
          base = (const unsigned char*)a.pw2 + u * UNIT;
          found = true;
        } else if (u < 4 + nffn) {
          const int v = u - 4, c = v >> 1;
          w2 = v & 1;
          base = w2 ? (const unsigned char*)a.ff_w2 + c * 128 : (const unsigned char*)a.ff_w1 + c * UNIT;
          found = true;
        } else {
          u -= 4 + nffn;
        }
      }
      if (HAS_A && !found) {
        if (u < nffn) {
          const int c = u >> 1;
          w2 = u & 1;
          base = w2 ? (const unsigned char*)a.ffm_w2 + c * 128 : (const unsigned char*)a.ffm_w1 + c * UNIT;
        } else {
          base = (const unsigned char*)a.wqkv + (u - nffn) * UNIT;
        }
      }
    }
    base = uniform_ptr(base);
    const unsigned dst = (unsigned)((g & (NSLOT - 1)) * UNIT + wave * 4096);  // smem is the only LDS object: offset 0
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(base, w2 ? w2ofs[i] : kofs[i], dst + i * 1024);
  };
  // parameter group grp (7 KiB of f32 vectors) -> LDS buffer grp & 1; waves 0..6 move 1 KiB each
  auto issue_params = [&](int grp) {
    if (wave < 7)
      glds16(uniform_ptr((const unsigned char*)(a.params + (size_t)grp * PAR_FLOATS)), wave * 1024 + lane * 16,
             (unsigned)(PAR_OFF + (grp & 1) * PAR_BYTES + wave * 1024));
  };
  int gi = 0;  // units issued so far
  int gs = 0;  // units consumed so far (= index of the next unit to consume)
  // Unit g may be issued once unit g - NSLOT has been consumed by every wave, i.e. after a barrier
  // that follows step g - NSLOT.  pump(done) is called right after such a barrier with done = gs.
  auto pump = [&](int done) {
    for (; gi < done + NSLOT && gi < total; ++gi) issue_unit(gi);
  };
  // Wait until unit gs has landed for this wave: at most the 4 loads of each later unit may still be in
  // flight.  Parameter loads and global stores issued in between only make the wait stricter (vmcnt
  // retires loads in order; nothing below relies on the order of stores).
  auto step_begin = [&]() {
    const int later = gi - 1 - gs;
    if (later >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    EM_LGKM0();
    __builtin_amdgcn_s_barrier();  // unit gs visible to every wave; unit gs - 1 released by every wave
    pump(gs);
  };

  // ---- row state ------------------------------------------------------------------------------
  float4 xr[4];        // residual x[m][64 f + ncol .. + 3], f32
  bf16x8 act[8];       // activation fragments of the running GEMM: frame mloc, k = 32 ks + 8 lg ..
  auto load_x = [&]() {
#pragma unroll
    for (int f = 0; f < 4; ++f) xr[f] = *(const float4*)(a.x + mrow * D + 64 * f + ncol);
  };
  auto store_x = [&]() {
    if (row_ok) {
#pragma unroll
      for (int f = 0; f < 4; ++f) *(float4*)(a.x + mrow * D + 64 * f + ncol) = xr[f];
    }
  };
  auto load_act = [&]() {  // from abuf, after a barrier
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      act[ks] = *(const bf16x8*)(abuf + ((ks >> 1) * 32 + mloc) * 128 + ((((ks & 1) * 4 + lg) ^ swz) << 4));
  };
  // LayerNorm statistics of the 32 rows (two-pass, as layer_norm.py / torch: mean, then the variance of
  // the centred values).  Row m is spread over 4 lane groups x 4 waves.  Two barriers.
  auto ln_stats = [&](float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) s += (xr[f].x + xr[f].y) + (xr[f].z + xr[f].w);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lg == 0) red[nf * 32 + mloc] = s;
    EM_LGKM0();
    __builtin_amdgcn_s_barrier();
    pump(gs);  // every wave has finished the previous unit: refill its slot during the epilogue
    mean = ((red[mloc] + red[32 + mloc]) + (red[64 + mloc] + red[96 + mloc])) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const float dx = xr[f].x - mean, dy = xr[f].y - mean, dz = xr[f].z - mean, dw = xr[f].w - mean;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    if (lg == 0) red[128 + nf * 32 + mloc] = q;
    EM_LGKM0();
    __builtin_amdgcn_s_barrier();
    const float var = ((red[128 + mloc] + red[160 + mloc]) + (red[192 + mloc] + red[224 + mloc])) * (1.0f / D);
    rstd = 1.0f / sqrtf(var + a.eps);
  };
  // y = LN(x; g, b) with g, b at float offsets go / bo of parameter buffer pb
  auto ln_apply = [&](const float* pb, int go, int bo, float4 y[4]) {
    float mean, rstd;
    ln_stats(mean, rstd);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const float4 g4 = *(const float4*)(pb + go + 64 * f + ncol);
      const float4 b4 = *(const float4*)(pb + bo + 64 * f + ncol);
      y[f].x = (xr[f].x - mean) * rstd * g4.x + b4.x;
      y[f].y = (xr[f].y - mean) * rstd * g4.y + b4.y;
      y[f].z = (xr[f].z - mean) * rstd * g4.z + b4.z;
      y[f].w = (xr[f].w - mean) * rstd * g4.w + b4.w;
    }
  };
  // LN(x) -> bf16 -> abuf -> activation fragments.  One more barrier.
  auto ln_to_act = [&](const float* pb, int go, int bo) {
    float4 y[4];
    ln_apply(pb, go, bo, y);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      bf16x4 pk = {(bf16)y[f].x, (bf16)y[f].y, (bf16)y[f].z, (bf16)y[f].w};
      *(bf16x4*)(abuf + (f * 32 + mloc) * 128 + (((2 * nf + (lg >> 1)) ^ (mloc & 7)) << 4) + (lg & 1) * 8) = pk;
    }
    EM_LGKM0();
    __builtin_amdgcn_s_barrier();
    load_act();
  };
  // one K unit: fragment C^T[n = nf*16 + lg*4 + r][m = mloc] (swap: C[m = mf*16 + lg*4 + r][n = nf*16 + lr])
  auto k_unit = [&](bool swap) -> f32x4 {
    const unsigned char* su = ring + (gs & (NSLOT - 1)) * UNIT + (nf * 16 + lr) * 128;
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ks += 2) {
      const bf16x8 w0 = *(const bf16x8*)(su + (ks >> 1) * 8192 + ((lg ^ swz) << 4));
      const bf16x8 w1 = *(const bf16x8*)(su + (ks >> 1) * 8192 + (((4 + lg) ^ swz) << 4));
      if (swap) {
        c0 = MM::mma(act[ks], w0, c0);
        c1 = MM::mma(act[ks + 1], w1, c1);
      } else {
        c0 = MM::mma(w0, act[ks], c0);
        c1 = MM::mma(w1, act[ks + 1], c1);
      }
    }
    return c0 + c1;
  };
  // x += scale * (W2 . swish(W1 . act + b1) + b2); b1 at pb + b1o, b2 at pb + b2o.  2 * nch units.
  auto ffn = [&](const float* pb, int b1o, int b2o, float scale) {
    f32x4 acc2[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) acc2[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nch; ++c) {
      step_begin();
      {
        const f32x4 h = k_unit(false);
        const float4 bb = *(const float4*)(pb + b1o + c * 64 + ncol);
        bf16x4 pk = {(bf16)swishf_(h[0] + bb.x), (bf16)swishf_(h[1] + bb.y), (bf16)swishf_(h[2] + bb.z),
                     (bf16)swishf_(h[3] + bb.w)};
        // H[m = mloc][k = ncol ..] into the [32][64] tile that aliases abuf's first k-tile
        *(bf16x4*)(abuf + mloc * 128 + (((2 * nf + (lg >> 1)) ^ (mloc & 7)) << 4) + (lg & 1) * 8) = pk;
      }
      ++gs;
      step_begin();
      {
        const unsigned char* su = ring + (gs & (NSLOT - 1)) * UNIT + (nf * 16 + lr) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int coff = ((ks * 4 + lg) ^ swz) << 4;
          const bf16x8 hf = *(const bf16x8*)(abuf + mloc * 128 + coff);
#pragma unroll
          for (int f = 0; f < 4; ++f)
            acc2[f] = MM::mma(*(const bf16x8*)(su + f * 8192 + coff), hf, acc2[f]);
        }
      }
      ++gs;
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const float4 b4 = *(const float4*)(pb + b2o + 64 * f + ncol);
      xr[f].x += scale * (acc2[f][0] + b4.x);
      xr[f].y += scale * (acc2[f][1] + b4.y);
      xr[f].z += scale * (acc2[f][2] + b4.z);
      xr[f].w += scale * (acc2[f][3] + b4.w);
    }
  };
  // x += (W . act + bias): four K units (N = 256)
  auto proj_resid = [&](const float* pb, int bo) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      step_begin();
      const f32x4 c = k_unit(false);
      ++gs;
      const float4 b4 = *(const float4*)(pb + bo + 64 * f + ncol);
      xr[f].x += c[0] + b4.x;
      xr[f].y += c[1] + b4.y;
      xr[f].z += c[2] + b4.z;
      xr[f].w += c[3] + b4.w;
    }
  };

  const float* const pb0 = par;
  const float* const pb1 = par + PAR_FLOATS;

  // ---- prologue: start the streams, then load the block's inputs ------------------------------
  issue_params(0);
  if (!HAS_C) issue_params(1);
  for (; gi < NSLOT - 1 && gi < total; ++gi) issue_unit(gi);  // slot 3 stays free for the conv tile

  if (HAS_C) {
    // linear_out over the attention context: activation fragments straight from global memory
    // (attention.py:151 linear_out; encoder_layer.py:142-147 residual)
    const bf16* crow = (const bf16*)a.ctx + mrow * D + lg * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) act[ks] = *(const bf16x8*)(crow + ks * 32);
    load_x();
    // hipcc counts only its own loads: left alone it waits for these at their first use, i.e. after the
    // next units have been issued, and the count then drains the ring.  Make it wait here instead.
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(act[ks]));
#pragma unroll
    for (int f = 0; f < 4; ++f) asm volatile("" : "+v"(xr[f].x), "+v"(xr[f].y), "+v"(xr[f].z), "+v"(xr[f].w));
    // G0: [bout 256][norm_conv g 256][b 256][pw1 bias, fused order 512]
    proj_resid(pb0, 0);
    store_x();
    ln_to_act(pb0, 256, 512);
    // pointwise_conv1 + GLU (convolution.py:66-69): unit 2j = value rows 64j.., unit 2j+1 = their gates
    for (int j = 0; j < 4; ++j) {
      step_begin();
      const f32x4 v = k_unit(false);
      ++gs;
      step_begin();
      const f32x4 gt = k_unit(false);
      ++gs;
      const float4 bv = *(const float4*)(pb0 + 768 + (2 * j) * 64 + ncol);
      const float4 bg = *(const float4*)(pb0 + 768 + (2 * j + 1) * 64 + ncol);
      bf16x4 pk = {(bf16)((v[0] + bv.x) * sigmoidf_(gt[0] + bg.x)), (bf16)((v[1] + bv.y) * sigmoidf_(gt[1] + bg.y)),
                   (bf16)((v[2] + bv.z) * sigmoidf_(gt[2] + bg.z)), (bf16)((v[3] + bv.w) * sigmoidf_(gt[3] + bg.w))};
      if (row_ok) *(bf16x4*)((bf16*)a.glu + mrow * D + 64 * j + ncol) = pk;
    }
    return;
  }

  int ga = 0;  // first parameter group of the A part
  if (HAS_D) {
    // ---- depthwise conv (k = 31, zero padded) + folded BatchNorm + Swish (convolution.py:72-75):
    // the 62-row input tile of this block (frames t0 - 15 .. t0 + 46 of the utterance, zero outside
    // [0, Tv)) goes through LDS; thread (channel c, half h) produces frames 16 h .. 16 h + 15.
    unsigned char* const tile = ring + (NSLOT - 1) * UNIT;  // [62][256] bf16
    load_x();  // before the tile loads, so that the wait for the tile also covers x (see block<C>)
    uint4 stage[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {  // unconditional loads from a clamped row (a load under a lane mask is
      const int q = tid + it * 512, r = q >> 5, ch = q & 31;  // waited for on the spot), zeroed afterwards
      const int t = t0 - HALF + r;
      const int tc = t < 0 ? 0 : (t < T ? t : T - 1);
      stage[it] = *(const uint4*)((const bf16*)a.glu + ((size_t)b * T + tc) * D + ch * 8);
    }
    const int c = tid & 255, hh = tid >> 8;
    float wk[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) wk[k] = a.dw_w[k * D + c];
    const float bc = a.dw_b[c];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = tid + it * 512, t = t0 - HALF + (q >> 5);
      if (q < TROWS * 32) *(uint4*)(tile + q * 16) = (t >= 0 && t < Tv) ? stage[it] : make_uint4(0u, 0u, 0u, 0u);
    }
    EM_LGKM0();
    __builtin_amdgcn_s_barrier();
    {
      float acc[16];
#pragma unroll
      for (int o = 0; o < 16; ++o) acc[o] = bc;
      const bf16* col = (const bf16*)tile + (hh * 16) * D + c;
#pragma unroll
      for (int r = 0; r < 16 + KW - 1; ++r) {
        const float v = (float)col[r * D];
#pragma unroll
        for (int o = 0; o < 16; ++o)
          if (r - o >= 0 && r - o < KW) acc[o] = fmaf(wk[r - o], v, acc[o]);
      }
      const int kt = c >> 6, kl = c & 63;
#pragma unroll
      for (int o = 0; o < 16; ++o) {
        const int m = hh * 16 + o;
        *(bf16*)(abuf + (kt * 32 + m) * 128 + (((kl >> 3) ^ (m & 7)) << 4) + (kl & 7) * 2) = (bf16)swishf_(acc[o]);
      }
    }
    EM_LGKM0();
    __builtin_amdgcn_s_barrier();  // conv output visible; the tile (slot 3) is free
    load_act();
    // G0: [pw2 bias 256][norm_ff g 256][b 256]; G1: [ff b1 1024][ff b2 256][norm_final g 256][b 256]
    proj_resid(pb0, 0);                  // pointwise_conv2 + residual (convolution.py:77, encoder_layer.py:149-158)
    ln_to_act(pb0, 256, 512);            // norm_ff
    issue_params(2);                     // G0 is dead (the barrier inside ln_to_act followed its last read)
    ffn(pb1, 0, 1024, 0.5f);             // x += 0.5 * FFN(norm_ff(x))   (encoder_layer.py:160-168)
    {
      float4 y[4];
      ln_apply(pb1, 1280, 1536, y);      // norm_final (encoder_layer.py:170-171): the block's output
#pragma unroll
      for (int f = 0; f < 4; ++f) xr[f] = y[f];
    }
    ga = 2;
    if (FINAL) {
      // after_norm (conformer_encoder.py:423-424); G2: [after_norm g 256][b 256]
      float4 y[4];
      ln_apply(pb0, 0, 256, y);
      if (row_ok) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          *(float4*)(a.enc_out + mrow * D + 64 * f + ncol) = y[f];
          bf16x4 pk = {(bf16)y[f].x, (bf16)y[f].y, (bf16)y[f].z, (bf16)y[f].w};
          *(bf16x4*)((bf16*)a.enc_act + mrow * D + 64 * f + ncol) = pk;
        }
      }
      return;
    }
  } else {
    load_x();
  }

  if (HAS_A) {
    // GA (buffer ga & 1 = 0): [norm_ff_macaron g 256][b 256][ffm b1 1024][ffm b2 256]
    // GA+1 (buffer 1):        [norm_mha g 256][b 256][bq | bk | bv 768]
    {
      // first LayerNorm of the next block.  Inside ln_stats the first barrier follows norm_final's last
      // read of G1, so the next group may replace it.
      float4 y[4];
      float mean, rstd;
      ln_stats(mean, rstd);
      if (HAS_D) issue_params(3);
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const float4 g4 = *(const float4*)(pb0 + 64 * f + ncol);
        const float4 b4 = *(const float4*)(pb0 + 256 + 64 * f + ncol);
        y[f].x = (xr[f].x - mean) * rstd * g4.x + b4.x;
        y[f].y = (xr[f].y - mean) * rstd * g4.y + b4.y;
        y[f].z = (xr[f].z - mean) * rstd * g4.z + b4.z;
        y[f].w = (xr[f].w - mean) * rstd * g4.w + b4.w;
        bf16x4 pk = {(bf16)y[f].x, (bf16)y[f].y, (bf16)y[f].z, (bf16)y[f].w};
        *(bf16x4*)(abuf + (f * 32 + mloc) * 128 + (((2 * nf + (lg >> 1)) ^ (mloc & 7)) << 4) + (lg & 1) * 8) = pk;
      }
      EM_LGKM0();
      __builtin_amdgcn_s_barrier();
      load_act();
    }
    ffn(pb0, 512, 1536, 0.5f);           // x += 0.5 * FFN_macaron(norm_ff_macaron(x))  (encoder_layer.py:108-121)
    store_x();
    ln_to_act(pb1, 0, 256);              // norm_mha (encoder_layer.py:123-127)
    // q / k / v projections (attention.py:91-97), written per head: Q, K as [B][H][Tpad][64], V transposed
    // as [B][H][64][Tpad] (computed with the MFMA operands swapped so a lane holds 4 consecutive frames).
    const int H = D / 64;
    for (int u = 0; u < 12; ++u) {
      step_begin();
      const int which = u >> 2, head = u & 3;
      const size_t bh = (size_t)b * H + head;
      if (which < 2) {
        const f32x4 cq = k_unit(false);
        ++gs;
        const float4 b4 = *(const float4*)(pb1 + 512 + u * 64 + ncol);
        bf16x4 pk = {(bf16)(cq[0] + b4.x), (bf16)(cq[1] + b4.y), (bf16)(cq[2] + b4.z), (bf16)(cq[3] + b4.w)};
        bf16* dst = (bf16*)(which ? a.kh : a.qh) + (bh * a.Tpad + t0 + mloc) * 64 + ncol;
        *(bf16x4*)dst = pk;
      } else {
        const f32x4 cv = k_unit(true);
        ++gs;
        const float bvv = pb1[512 + u * 64 + nf * 16 + lr];
        bf16x4 pk = {(bf16)(cv[0] + bvv), (bf16)(cv[1] + bvv), (bf16)(cv[2] + bvv), (bf16)(cv[3] + bvv)};
        bf16* dst = (bf16*)a.vt + (bh * 64 + nf * 16 + lr) * a.Tpad + t0 + mf * 16 + lg * 4;
        *(bf16x4*)dst = pk;
      }
    }
  }
}

template <int MODE>
int launch_block(const EmBlockArgs* a, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)block_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            SMEM_BYTES) != hipSuccess)
      return EM_ERR_LAUNCH;
    attr_set = true;
  }
  dim3 grid(em_cdiv(a->T, BM), a->B);
  hipLaunchKernelGGL((block_kernel<MODE>), grid, dim3(512), SMEM_BYTES, s, *a);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

}  // namespace

extern "C" int em_conformer_block_fused(int mode, const EmBlockArgs* a, void* stream) {
  if (!a || !a->x || !a->params || a->B <= 0 || a->T <= 0) return EM_ERR_BAD_ARG;
  if (a->d != D || a->ff <= 0 || a->ff % 64 != 0 || a->ff > 1024) return EM_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const bool need_a = (mode & EM_BLOCK_A) != 0, need_d = (mode & EM_BLOCK_D) != 0;
  if (need_a) {
    if (!a->ffm_w1 || !a->ffm_w2 || !a->wqkv || !a->qh || !a->kh || !a->vt) return EM_ERR_BAD_ARG;
    if (a->Tpad % 64 != 0 || a->Tpad < em_cdiv(a->T, BM) * BM) return EM_ERR_BAD_ARG;
  }
  if (need_d) {
    if (!a->glu || !a->pw2 || !a->ff_w1 || !a->ff_w2 || !a->dw_w || !a->dw_b) return EM_ERR_BAD_ARG;
    if (a->kernel != KW) return EM_ERR_UNSUPPORTED;
  }
  if (mode == EM_BLOCK_C && (!a->ctx || !a->glu || !a->wout || !a->pw1f)) return EM_ERR_BAD_ARG;
  if (mode == (EM_BLOCK_D | EM_BLOCK_FINAL) && (!a->enc_out || !a->enc_act)) return EM_ERR_BAD_ARG;
  // algorithmic flops of the GEMM-shaped stages (the depthwise conv and LayerNorms are VALU work)
  const double M = (double)a->B * a->T;
  double flops = 0.0;
  if (mode & EM_BLOCK_C) flops += 2.0 * M * D * (D + 2 * D);
  if (mode & EM_BLOCK_D) flops += 2.0 * M * D * (D + 2.0 * a->ff);
  if (mode & EM_BLOCK_A) flops += 2.0 * M * D * (2.0 * a->ff + 3 * D);
  const bool rec = em_prof_begin(stream);
  int rc = EM_ERR_BAD_ARG;
  switch (mode) {
    case EM_BLOCK_C: rc = launch_block<EM_BLOCK_C>(a, s); break;
    case EM_BLOCK_A: rc = launch_block<EM_BLOCK_A>(a, s); break;
    case EM_BLOCK_D | EM_BLOCK_A: rc = launch_block<EM_BLOCK_D | EM_BLOCK_A>(a, s); break;
    case EM_BLOCK_D | EM_BLOCK_FINAL: rc = launch_block<EM_BLOCK_D | EM_BLOCK_FINAL>(a, s); break;
  }
  if (rec) em_prof_end(stream, flops, EM_PROF_BLOCK);
  return rc;
}
