// Row-block launches of the 512-wide Conformer (bf16 MFMA): the block's row-local operators as three launches of 64-row
// workgroups (template flags, see the kernel):  [macaron FFN + residual + norm_mha]   [linear_out + residual + norm_conv +
// pointwise_conv1 + GLU] (PRE, MAIN = EM_ROWS_GLU)   [pointwise_conv2 + residual + norm_ff + FFN + residual + norm_final +
// the next LayerNorm] (PRE; POST: the CTC head's arg-max walked behind the stack's last launch).  The core is the
// feed-forward module: one launch computes
//     x += scale * (W2 . swish(W1 . xn + b1) + b2);   xn' = LayerNorm(x)          (ln_mode 1)
//     x  = LayerNorm_1(x + scale * (...));            xn' = LayerNorm_2(x)        (ln_mode 2)
// for 64 consecutive rows per workgroup, i.e. the three launches  GEMM(Swish) -> GEMM(+residual) -> LayerNorm  of the
// per-operator sequence (csrc/encoder.hip) with the [M][ff] hidden activation never written to memory and the f32
// residual stream read and written once.
//
// Reference: PositionwiseFeedForward.forward (transformer/positionwise_feed_forward.py:30-32: w_2(act(w_1 x))) inside
// EncoderLayer.forward (conformer/encoder_layer.py:111-121 macaron, :160-168 feed_forward with ff_scale 0.5, :170-171
// norm_final), followed by the LayerNorm of the NEXT consumer (norm_mha :124; norm_ff_macaron of the next block :112;
// ConformerEncoder's after_norm, conformer_encoder.py:410).  LayerNorm eps 1e-12 (transformer/layer_norm.py:23).
//
// Why this shape.  At B = 64 the large model has M = 15 936 rows = 249 workgroups of 64 rows: one per
// CU.  A workgroup that owns 64 rows ingests every weight byte once per 64 rows, so the L2 -> CU weight stream (64 B/clk
// per CU) and the matrix cores take the same time (per 128 hidden units: 256 KiB of weights = 4 096 cycles; 1 024 MFMAs
// over four SIMDs = 4 096 cycles): the balance the 256-wide block kernels (32 rows, csrc/block.hip) cannot reach.
//   * EIGHT waves, two per SIMD.  The first version had four (one per SIMD, 512 registers each) and measured 7.1 K cycles
//     per chunk against the 4.1 K of its MFMAs: total = MFMA time + 4 cycles x (every other instruction) to within 3 %
//     (developer builds without weight requests / LDS reads / both, profiles/r04m) - a wave alone on its SIMD does not
//     hide its own loads, LDS reads, waits and Swish arithmetic behind its own MFMAs.  With two waves per SIMD one
//     wave's non-MFMA instructions issue under the other's MFMAs.
//   * GEMM 1 is split over the HIDDEN dimension (wave w: hidden columns 16 w .. 16 w + 15 of every 128-wide chunk, all 64
//     rows), GEMM 2 over the OUTPUT columns (wave w: columns 64 w .. 64 w + 63, all 64 rows: 16 accumulator fragments =
//     64 registers that live for the whole kernel);
//   * LN(x) of the 64 rows (bf16, 64 KiB) is LDS-resident: loaded once by LDS-DMA, 16-byte chunks XOR-swizzled by the
//     row so that the MFMA B-operand reads (one row per lane) are bank-conflict free;
//   * weights go global memory -> MFMA A-operand registers, fragment-major (a wave-wide load = 1 KiB contiguous, the
//     eight waves' current fragments = 8 consecutive KiB, i.e. every L2 channel) through a ring of 16 fragments per
//     wave: a slot is refilled the moment its 4 MFMAs are issued, with the fragment the wave needs 16 fragments (~2 000
//     cycles of the SIMD's MFMA time) later.  Requests are buffer loads (scalar chunk base in the resource, the fragment
//     offset a scalar constant, lane offset in one VGPR): no address arithmetic in the loop.  No LDS staging, no
//     barrier for the stream;
//   * the hidden activation of a chunk (bias + Swish + bf16 in registers, already in B-operand layout because the host
//     orders W2's contraction index to match) is exchanged between the waves through a double-buffered 16 KiB LDS
//     tile: ONE barrier per chunk, placed half a GEMM-1 chunk after the tile was written;
//   * software pipeline: [GEMM 1 of chunk c, barrier in its middle] [GEMM 2 of chunk c - 1 with the Swish of chunk c
//     between its MFMAs];
//   * epilogue: residual add, LayerNorm statistics (Chan's pairwise update: lane groups through register swaps, the eight
//     waves through one LDS exchange), normalised rows out as bf16 for the next GEMM.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "em_common.h"
#include "switches.h"

#ifndef EM_FFN_DBG
#define EM_FFN_DBG 0  // developer builds (tools/build_block_variants.sh ffn<d>; timing only, results wrong by design unless 8):
#endif                // 1 no MFMAs, 2 no weight requests in the loop, 4 no LDS operand reads in the loop, 8 cycle stamps (EM_FFN_STAMPS=1)

namespace {

constexpr int D = 512;    // model width
constexpr int RB = 64;    // rows per workgroup
constexpr int NW = 8;     // waves
constexpr int NT = 64 * NW;
constexpr int CH = 128;   // hidden units per chunk
constexpr int CHUNK_BYTES = CH * D * 2;       // of W1 and of W2 each: 16 fragments x 8 waves x 1 KiB
constexpr int ACT_OFF = 0;                    // 64 KiB: LN(x) [64 rows][1 KiB], chunk q of row r at (q & ~15) | ((q ^ r) & 15)
constexpr int H_OFF = ACT_OFF + RB * D * 2;   // 2 x 16 KiB: hidden activation tiles [buf][k-step s][row fragment][lane][16 B]
constexpr int RED_OFF = H_OFF + 2 * 16384;    // 2 x 4 KiB: LayerNorm partials [set][wave][64 rows] (sum, M2)
constexpr int PAR_OFF = RED_OFF + 8192;       // 8 x 2 KiB: pre_b, pre_g, pre_be, b2, g1, be1, g2, be2 (512 f32 each), by LDS-DMA in the prologue
constexpr int SMEM_BYTES = PAR_OFF + 8 * 2048;
static_assert(SMEM_BYTES <= 160 * 1024, "LDS");

typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// LDS-DMA from inline asm (as csrc/attention2.hip): 16 bytes per lane from sbase + voff to LDS lds_dst + 16 lane
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

__device__ __forceinline__ float rsqrt_nr(float var) {
  // v_rsq_f32 (1 ulp) + one Newton step: f32-accurate without the IEEE sqrt + division sequences (as csrc/block.hip)
  const float y0 = __builtin_amdgcn_rsqf(var);
  return y0 * (1.5f - 0.5f * var * y0 * y0);
}

// PRE: the launch starts with a projection of its own - x += W_pre . pre_in + b_pre (pointwise_conv2 of the conv module,
// convolution.py:78-79 + the residual of encoder_layer.py:158) followed by LayerNorm(pre_g, pre_be) (norm_ff, :161) whose
// result becomes the feed-forward module's input: the GEMM + residual launch and the LayerNorm launch in front of the
// second feed-forward module of a block (19.1 + 9.8 us at B = 64) become 256 more MFMAs per wave in this one.
// MAIN = EM_ROWS_GLU: what follows the projection is not a feed-forward module but pointwise_conv1 + GLU of the conv module
// (convolution.py:62-66): the launch is  x += linear_out . ctx + b  (attention.py:149-151 + the residual of
// encoder_layer.py:147) -> norm_conv (:151) -> glu = value * sigmoid(gate), i.e. the GEMM + residual, LayerNorm and GLU GEMM
// launches between the attention and the depthwise conv (19.1 + 9.8 + 29.7 us at B = 64).  Its matrix is walked in the
// GEMM-1 form (wave w: 16 output columns of every 128-row chunk), chunks alternating value rows and their gate rows, so a
// lane holds matching value / gate pairs; no hidden tile, no barrier in the loop.
// POST 1: the CTC head's arg-max walk behind the launch (the last launch of the stack), see the end of the kernel.
// POST 2 (round 6): the attention's q | k | v projections walked behind a ln_mode 1 launch (macaron FFN + residual + norm_mha):
// their per-head operands (csrc/attention2.hip) written straight from the walk, LN(x) itself never stored.
template <int LNMODE, int PRE, int MAIN, int POST = 0>
__global__ __launch_bounds__(NT, 1) void ffn_rows_kernel(const EmFfnRowsArgs a, long long* __restrict__ stamps) {
  using MM = Mma<bf16>;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * RB, M = a.M;
  const int nch = a.ff / CH;
  // developer timing (-DEM_FFN_DBG=8, EM_FFN_STAMPS=1): cycle stamps of wave 0 of workgroup 5
  constexpr int dbg = EM_FFN_DBG;
  int nts = 0;
  auto stamp = [&]() {
    if constexpr ((dbg & 8) != 0) {
      if (stamps && blockIdx.x == 5 && tid == 0 && nts < 64) stamps[nts] = (long long)__builtin_amdgcn_s_memtime();
      ++nts;
    }
  };
  stamp();

  // ---- LN(x) of the 64 rows -> LDS (LDS-DMA, 1 KiB = one row per wave-instruction; lane l fills LDS chunk l of the row
  // with global chunk (l & ~15) | ((l ^ row) & 15)).  Requested FIRST: GEMM 1 needs all of it and only the head of the ring.
  // (from inline asm: an LDS-DMA hipcc knows about makes its wait-count pass put s_waitcnt vmcnt(0) in front of the first LDS
  // read - i.e. wait for the ring AND the residual rows requested behind it - instead of the counted wait below)
  // (PRE == 2, round 6: no projection and no tile to fetch - the module's input is LayerNorm(x; pre_g, pre_be) of the residual
  // rows themselves, computed where a projection launch computes it: the LayerNorm launch in front of E-Branchformer's second
  // feed-forward module, e_branchformer_encoder.py:172-174, is gone)
  if constexpr (PRE != 2) {
    const unsigned char* src = (const unsigned char*)(PRE ? a.pre_in : a.xn_in);
#pragma unroll
    for (int i = 0; i < RB / NW; ++i) {
      const int row = wave * (RB / NW) + i;
      int m = m0 + row;
      m = m < M ? m : M - 1;
      const int g = (lane & ~15) | ((lane ^ row) & 15);
      glds16(uniform_ptr(src + (size_t)m * (D * 2)), g * 16, ACT_OFF + row * 1024);
    }
  }
  // ---- the bias / LayerNorm vectors -> LDS, wave w vector w.  Read from global memory where they are used, each LayerNorm
  // exposed a dependent L2 round trip (and, behind a burst of stores, the stores' acknowledgement: loads and stores share
  // vmcnt) - the hand-over between the projection and the GLU walk took 22 K cycles (stamps r04v).
  {
    const float* const vecs[8] = {a.pre_b, a.pre_g, a.pre_be, a.b2, a.g1, a.be1, a.g2, a.be2};
    const float* v = vecs[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) v = wave == i ? vecs[i] : v;
    if (v != nullptr) {  // (uniform)
      const unsigned char* vb = uniform_ptr((const unsigned char*)v);
      glds16(vb, lane * 16, PAR_OFF + wave * 2048);
      glds16(vb, 1024 + lane * 16, PAR_OFF + wave * 2048 + 1024);
    }
  }
  // ---- the residual rows (lane (lr, lg): columns 64 wave + 16 cf + 4 lg + r of rows 16 rf + lr).  A launch that starts with
  // a projection needs them at its first MFMA - the accumulators start as x - so it requests them HERE, in front of the ring's
  // head, and the counted wait below covers them: requested behind the prologue barrier (as the plain launch does, whose
  // first GEMM 1 does not need them) the waves got their rows up to 8 K cycles apart, ran the projection out of step and the
  // LayerNorm barrier behind it waited for the last one (hand-over 11 K cycles, stamps r04x).
  const int col0 = 64 * wave + 4 * lg;
  f32x4 acc2[4][4];
  auto load_x = [&](f32x4 (&v)[4][4]) {
#pragma unroll
    for (int rf = 0; rf < 4; ++rf) {
      int m = m0 + rf * 16 + lr;
      m = m < M ? m : M - 1;
      const float* xr = a.x + (size_t)m * D + col0;
#pragma unroll
      for (int cf = 0; cf < 4; ++cf) v[cf][rf] = *(const f32x4*)(xr + 16 * cf);
    }
  };
  if constexpr (PRE != 0) load_x(acc2);
  __builtin_amdgcn_sched_barrier(0);
  // ---- weight stream: per chunk 16 fragments x [8 waves] x 1 KiB of W1 and as many of W2 (include/espnet_amd.h,
  // EmFfnRowsArgs): fragment i of this wave at chunk + i * 8 KiB + wave * 1 KiB + lane * 16
  bf16x8 ring[16];
  const unsigned voff = wave * 1024 + lane * 16;
  auto rsrc = [&](const void* base, int c) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)base + (size_t)c * CHUNK_BYTES), 0, CHUNK_BYTES, 0x00020000);
  };
  auto ld = [&](__amdgpu_buffer_rsrc_t rs, int i) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, i * 8192, 0));
  };
  // (PRE: the projection's 64 fragments x [8 waves] x 1 KiB, [ks][cf][wave][lane][e] = W_pre[64 wave + 16 cf + lr][32 ks + 8 lg + e])
  const __amdgpu_buffer_rsrc_t rs_pre =
      __builtin_amdgcn_make_buffer_rsrc((void*)(PRE == 1 ? a.pre_w : a.w1p), 0, 4 * CHUNK_BYTES, 0x00020000);
  {
    const __amdgpu_buffer_rsrc_t r0 = PRE == 1 ? rs_pre : rsrc(a.w1p, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) ring[i] = ld(r0, i);
  }
  // this lane's first-bias values: hidden 128 c + 16 wave + 4 lg + r
  const float* b1p = a.b1 + 16 * wave + 4 * lg;
  float4 bia = *(const float4*)(b1p);

  // B-operand read offsets of GEMM 1: row 16 rf + lr, k-step ks = 4 kq + kj: chunk 16 kq + ((4 kj + lg) ^ lr)
  int aoff[4];
  {
    const int x0 = lg ^ lr;
#pragma unroll
    for (int kj = 0; kj < 4; ++kj) aoff[kj] = ACT_OFF + lr * 1024 + (((kj * 4) ^ x0) << 4);
  }
  const int hrd = H_OFF + lane * 16;                                        // + buf * 16384 + s * 4096 + rf * 1024
  const int hwr = H_OFF + (wave >> 1) * 4096 + lane * 16 + (wave & 1) * 8;  // + buf * 16384 + rf * 1024

  f32x4 acc1[4];
  // The residual rides in GEMM 2's accumulators: acc2 starts as x / scale + b2 and the epilogue multiplies by scale (exact
  // for the ff_scale of 0.5: a power of two).  The rows are requested right behind the prologue barrier and travel under
  // the first GEMM 1 (older than every refill: the in-order return costs nothing).  Measured alternatives (gpurun calls
  // r04m - r04q, developer stamps): all rows in front of the last GEMM 2 (the first version) - every workgroup asks at once
  // and waits ~5 K cycles with idle matrix cores, 83.6 us; all rows in the prologue beside LN(x) - LN(x) lands at 16 K
  // instead of 7 K cycles, 88 us; one 16-row piece per iteration of the first four chunks - every GEMM 2 phase 4.4 K
  // instead of 3.1 K cycles (requests with 64-byte segments between the weight requests), 87 us; this one 82.9 us.
  // Lane (lr, lg) holds columns 64 wave + 16 cf + 4 lg + r of rows 16 rf + lr.
  // bias / LayerNorm vector `which` (order of PAR_OFF) at this lane's columns 16 cf ..
  auto par4 = [&](int which, int cf) { return *(const float4*)(smem + PAR_OFF + which * 2048 + (col0 + 16 * cf) * 4); };
  auto parv = [&](int which, int cf) { return *(const f32x4*)(smem + PAR_OFF + which * 2048 + (col0 + 16 * cf) * 4); };
  const float scale = a.scale, inv_scale = 1.0f / a.scale;

  // The instruction order below is pinned with sched_barrier(0) after every group of [4 MFMAs + 1 weight request]: left
  // alone hipcc gathers the requests of a phase into one cluster at the phase's end, i.e. directly in front of
  // their first use, and the ring's lookahead is gone.  The LDS operand reads are therefore pipelined by hand: a
  // k-step's fragments are requested one k-step ahead, across the phase boundaries too (af0 / h0).
  bf16x8 af0[4], h0[4];
  auto read_act = [&](int ks, bf16x8 (&af)[4]) {
#pragma unroll
    for (int rf = 0; rf < 4; ++rf) af[rf] = *(const bf16x8*)(smem + aoff[ks & 3] + rf * 16384 + (ks >> 2) * 256);
  };
  auto read_h = [&](int buf, int s, bf16x8 (&hf_)[4]) {
#pragma unroll
    for (int rf = 0; rf < 4; ++rf) hf_[rf] = *(const bf16x8*)(smem + hrd + buf * 16384 + s * 4096 + rf * 1024);
  };
  auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // GEMM 1 of a chunk: H^T[hidden 16 wave + ..][row] over K = 512; every slot refilled from chunk `refill`.
  // SYNC: the chunk barrier sits in the middle (k-step 8) and the first hidden fragments of tile `hbuf` are requested at the end.
  auto gemm1 = [&](__amdgpu_buffer_rsrc_t refill, auto sync, int hbuf, auto walk) {
    constexpr bool SYNC = decltype(sync)::value;
    constexpr bool WALK = decltype(walk)::value || MAIN == EM_ROWS_GLU;  // chunk after chunk of GEMM 1: prefetch the next one's head
    bf16x8 cur[4], nxt[4];
#pragma unroll
    for (int rf = 0; rf < 4; ++rf) cur[rf] = af0[rf];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 1 < 16 && !(dbg & 4)) read_act(ks + 1, nxt);
      if (SYNC && ks == 8) barrier();
      if (SYNC && ks == 15) read_h(hbuf, 0, h0);
      if (WALK && ks == 15) read_act(0, af0);  // (the next chunk's first fragments; af0 was copied to cur above)
#pragma unroll
      for (int rf = 0; rf < 4; ++rf) {
        if constexpr ((dbg & 1) == 0)
          acc1[rf] = MM::mma(ring[ks], cur[rf], ks == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc1[rf]);
        else
          acc1[rf][rf] = (float)ring[ks][rf] + (float)cur[rf][1];
      }
      if constexpr ((dbg & 2) == 0) ring[ks] = ld(refill, ks);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rf = 0; rf < 4; ++rf)
        if (!(dbg & 4)) cur[rf] = nxt[rf];
    }
  };
  // bias + Swish of ONE hidden value (row fragment rf, hidden 16 wave + 4 lg + r); after the fourth: bf16 -> the wave's
  // half of the B fragments of k-step (wave >> 1) in hidden tile `buf`
  float hv[4];
  auto swish_piece = [&](int rf, int r, int buf) {
    const float bias = r == 0 ? bia.x : r == 1 ? bia.y : r == 2 ? bia.z : bia.w;
    hv[r] = swishf_(acc1[rf][r] + bias);
    if (r == 3) {
      const bf16x4 hb = {(bf16)hv[0], (bf16)hv[1], (bf16)hv[2], (bf16)hv[3]};
      *(bf16x4*)(smem + hwr + buf * 16384 + rf * 1024) = hb;
    }
  };
  // GEMM 2 of a chunk from hidden tile `buf` (its first fragments already in h0): out^T[col 64 wave + 16 cf + ..][row] +=
  // W2 . H^T; `piece(s, cf)` runs beside every group of 4 MFMAs (the Swish of the NEXT chunk); ends by requesting the first
  // activation fragments of the next GEMM 1
  auto gemm2 = [&](int buf, __amdgpu_buffer_rsrc_t refill, auto do_refill, auto&& piece) {
    bf16x8 cur[4], nxt[4];
#pragma unroll
    for (int rf = 0; rf < 4; ++rf) cur[rf] = h0[rf];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if constexpr ((dbg & 4) == 0) {
        if (s + 1 < 4) read_h(buf, s + 1, nxt);
        else read_act(0, af0);
      }
#pragma unroll
      for (int cf = 0; cf < 4; ++cf) {
        piece(s, cf);
#pragma unroll
        for (int rf = 0; rf < 4; ++rf) {
          if constexpr ((dbg & 1) == 0) acc2[cf][rf] = MM::mma(ring[s * 4 + cf], cur[rf], acc2[cf][rf]);
          else acc2[cf][rf][rf] += (float)ring[s * 4 + cf][rf] + (float)cur[rf][cf];
        }
        if constexpr (decltype(do_refill)::value && (dbg & 2) == 0) ring[s * 4 + cf] = ld(refill, s * 4 + cf);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int rf = 0; rf < 4; ++rf)
        if (!(dbg & 4)) cur[rf] = nxt[rf];
    }
  };

  // LN(x) (and a projection launch's residual rows) have landed (requested before the 16 ring fragments and the bias: loads
  // return in order; the memory clobbers keep
  // hipcc from moving any other request across either asm statement, so the count is exact)
  asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  stamp();
  if constexpr (PRE == 0) load_x(acc2);
  __builtin_amdgcn_sched_barrier(0);
  read_act(0, af0);

  // LayerNorm statistics of the 64 rows over values v[cf][rf] (lane: columns 64 wave + 16 cf + 4 lg + r of rows 16 rf + lr):
  // every lane reduces its 16 values to (sum, M2 about its own mean), the lane groups combine with Chan's update for
  // equal-sized groups (M2 = M2a + M2b + (sa - sb)^2 / (2 n)) through register swaps, the eight waves through ONE LDS
  // exchange, summed in wave order whatever the arrival order: deterministic.  `set` alternates the exchange buffer.
  auto ln_stats = [&](const f32x4 (&v)[4][4], int set, float (&mean)[4], float (&rstd)[4]) {
    float2* const red = (float2*)(smem + RED_OFF + set * 4096);
#pragma unroll
    for (int rf = 0; rf < 4; ++rf) {
      // (whole-vector arithmetic: hipcc turns it into v_pk_*_f32, two values per instruction - written value by value one
      // LayerNorm was ~850 VALU instructions per wave, 8 K cycles with the matrix cores idle: stamps r04w)
      const f32x4 s4 = (v[0][rf] + v[1][rf]) + (v[2][rf] + v[3][rf]);
      float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
      const float mu = s * (1.0f / 16.0f);
      const f32x4 d0 = v[0][rf] - mu, d1 = v[1][rf] - mu, d2 = v[2][rf] - mu, d3 = v[3][rf] - mu;
      const f32x4 q4 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      float q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
      {
        const Pair2 ps = wave_xor16_pair(s), pq = wave_xor16_pair(q);
        q = pq.a + pq.b + (ps.a - ps.b) * (ps.a - ps.b) * (1.0f / 32.0f);
        s = ps.a + ps.b;
      }
      {
        const Pair2 ps = wave_xor32_pair(s), pq = wave_xor32_pair(q);
        q = pq.a + pq.b + (ps.a - ps.b) * (ps.a - ps.b) * (1.0f / 64.0f);
        s = ps.a + ps.b;
      }
      if (lg == 0) red[wave * 64 + rf * 16 + lr] = make_float2(s, q);
    }
    barrier();
#pragma unroll
    for (int rf = 0; rf < 4; ++rf) {
      const int r = rf * 16 + lr;
      float2 p[8];
#pragma unroll
      for (int w = 0; w < 8; ++w) p[w] = red[w * 64 + r];
#pragma unroll
      for (int w = 0; w < 8; w += 2) {
        p[w].y = p[w].y + p[w + 1].y + (p[w].x - p[w + 1].x) * (p[w].x - p[w + 1].x) * (1.0f / 128.0f);
        p[w].x += p[w + 1].x;
      }
#pragma unroll
      for (int w = 0; w < 8; w += 4) {
        p[w].y = p[w].y + p[w + 2].y + (p[w].x - p[w + 2].x) * (p[w].x - p[w + 2].x) * (1.0f / 256.0f);
        p[w].x += p[w + 2].x;
      }
      const float qq = p[0].y + p[4].y + (p[0].x - p[4].x) * (p[0].x - p[4].x) * (1.0f / 512.0f);
      mean[rf] = (p[0].x + p[4].x) * (1.0f / D);
      rstd[rf] = rsqrt_nr(qq * (1.0f / D) + a.eps);
    }
  };

  if constexpr (PRE != 0) {
    // ---- the projection: accp^T[col 64 wave + 16 cf + ..][row] = W_pre . tile^T over K = 512, GEMM-2 style (the tile's
    // fragments one k-step ahead, a slot refilled behind its 4 MFMAs: the projection's fragment 16 positions on, then
    // the head of W1's first chunk)
    // The accumulators START as the residual rows (requested behind the prologue barrier like the plain launch's): no
    // registers of their own.  (Two versions with the rows in registers of their own - requested in front of the projection,
    // or at its k-step 12 - pushed the kernel past 256 registers: hipcc spilled ring refills the moment they were loaded, each
    // behind an s_waitcnt vmcnt(0), and the 256 MFMAs per wave took 20 K cycles: stamps r04u.)  The price: the first MFMAs
    // wait for the rows, ~10 K cycles at the rate 249 workgroups get their 128 KiB each.
    if constexpr (PRE == 1) {
      const __amdgpu_buffer_rsrc_t rw1 = rsrc(a.w1p, 0);
      bf16x8 cur[4], nxt[4];
#pragma unroll
      for (int rf = 0; rf < 4; ++rf) cur[rf] = af0[rf];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks + 1 < 16) read_act(ks + 1, nxt);
#pragma unroll
        for (int cf = 0; cf < 4; ++cf) {
          const int pos = ks * 4 + cf;
#pragma unroll
          for (int rf = 0; rf < 4; ++rf) acc2[cf][rf] = MM::mma(ring[pos & 15], cur[rf], acc2[cf][rf]);
          ring[pos & 15] = pos + 16 < 64 ? ld(rs_pre, pos + 16) : ld(rw1, pos + 16 - 64);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int rf = 0; rf < 4; ++rf) cur[rf] = nxt[rf];
      }
    }
    if constexpr ((dbg & 8) != 0) asm volatile("s_nop 0" : "+v"(acc2[3][3]));
    stamp();
    // x' = x + (acc + b_pre): the residual stream for the rest of the launch
    if constexpr (PRE == 1) {
#pragma unroll
      for (int cf = 0; cf < 4; ++cf) {
        const f32x4 bp = parv(0, cf);
#pragma unroll
        for (int rf = 0; rf < 4; ++rf) acc2[cf][rf] += bp;
      }
    }
    if constexpr (MAIN == EM_ROWS_GLU) {
      // the residual stream leaves here: nothing else of this launch touches it, and the GLU walk needs the registers.
      // (Stores count on vmcnt like the weight requests; issued here they drain under the LayerNorm, whose parameters come
      // from LDS - behind the LayerNorm they delayed the first four chunks of the walk: stamps r04w.)
#pragma unroll
      for (int rf = 0; rf < 4; ++rf) {
        const int m = m0 + rf * 16 + lr;
        if (m < M) {
          float* o = a.x + (size_t)m * D + col0;
#pragma unroll
          for (int cf = 0; cf < 4; ++cf) *(f32x4*)(o + 16 * cf) = acc2[cf][rf];
        }
      }
    }
    // LayerNorm(x') -> bf16 -> the activation tile (its barrier also says that every wave is done reading the tile)
    float mean[4], rstd[4];
    ln_stats(acc2, 0, mean, rstd);
#pragma unroll
    for (int cf = 0; cf < 4; ++cf) {
      const f32x4 g4 = parv(1, cf), b4 = parv(2, cf);
      const int q = 8 * wave + 2 * cf + (lg >> 1);  // 16-byte chunk of the row that holds columns col0 + 16 cf ..
      const int pos = (q & ~15) | ((q ^ lr) & 15);
#pragma unroll
      for (int rf = 0; rf < 4; ++rf) {
        const f32x4 y = (acc2[cf][rf] - mean[rf]) * (g4 * rstd[rf]) + b4;
        *(bf16x4*)(smem + ACT_OFF + (rf * 16 + lr) * 1024 + pos * 16 + (lg & 1) * 8) = __builtin_convertvector(y, bf16x4);
      }
    }
    barrier();
    read_act(0, af0);
    stamp();
  }

  if constexpr (MAIN == EM_ROWS_GLU) {
    static_assert(PRE == 1, "the GLU launch starts with the projection");
    // chunk 2 j: value rows of output columns 128 j .., chunk 2 j + 1: their gate rows (host: pack_rows_glu); this lane:
    // columns 128 j + 16 wave + 4 lg + r of rows 16 rf + lr.  Nothing is stored INSIDE the walk: stores count on vmcnt like
    // the weight requests, and with every pair's outputs stored behind its gate chunk every chunk took 4.3 K cycles instead
    // of 2.6 K (stamps r04u).  The four pairs' packed outputs wait in 32 registers.
    bf16x4 outp[4][4];
    f32x4 val[4];
    float4 bv = bia;
    constexpr int NPAIR = 4;  // 2 * 512 rows = 8 chunks
#pragma unroll
    for (int c = 0; c < 2 * NPAIR; ++c) {
      const float4 bnext = *(const float4*)(b1p + (c + 1 < 2 * NPAIR ? c + 1 : c) * CH);
      gemm1(rsrc(a.w1p, c + 1 < 2 * NPAIR ? c + 1 : c), std::false_type{}, 0, std::true_type{});
      if ((c & 1) == 0) {
#pragma unroll
        for (int rf = 0; rf < 4; ++rf) val[rf] = acc1[rf] + (f32x4){bv.x, bv.y, bv.z, bv.w};
      } else {
#pragma unroll
        for (int rf = 0; rf < 4; ++rf)
          outp[c >> 1][rf] = (bf16x4){(bf16)(val[rf][0] * sigmoidf_(acc1[rf][0] + bv.x)), (bf16)(val[rf][1] * sigmoidf_(acc1[rf][1] + bv.y)),
                                      (bf16)(val[rf][2] * sigmoidf_(acc1[rf][2] + bv.z)), (bf16)(val[rf][3] * sigmoidf_(acc1[rf][3] + bv.w))};
      }
      bv = bnext;
      if constexpr ((dbg & 8) != 0) asm volatile("s_nop 0" : "+v"(acc1[3]));
      stamp();
    }
    bf16* const out = (bf16*)a.xn_out;
#pragma unroll
    for (int rf = 0; rf < 4; ++rf) {
      const int m = m0 + rf * 16 + lr;
      if (m < M) {
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) *(bf16x4*)(out + (size_t)m * D + j * CH + 16 * wave + 4 * lg) = outp[j][rf];
      }
    }
    stamp();
    return;
  }

  gemm1(nch > 1 ? rsrc(a.w1p, 1) : rsrc(a.w2p, 0), std::false_type{}, 0, std::false_type{});
#pragma unroll
  for (int rf = 0; rf < 4; ++rf)
#pragma unroll
    for (int r = 0; r < 4; ++r) swish_piece(rf, r, 0);
#pragma unroll
  for (int cf = 0; cf < 4; ++cf) {
    const f32x4 b2 = parv(3, cf);
#pragma unroll
    for (int rf = 0; rf < 4; ++rf) acc2[cf][rf] = acc2[cf][rf] * inv_scale + b2;
  }
#pragma unroll 1
  for (int c = 1; c < nch; ++c) {
    // (requests are unconditional and their addresses always valid: hipcc counts its waits only over straight-line loads)
    bia = *(const float4*)(b1p + c * CH);
    // barrier inside: tile (c - 1) & 1 is complete; everybody is done reading tile c & 1 (chunk c - 2)
    gemm1(rsrc(a.w2p, c - 1), std::true_type{}, (c - 1) & 1, std::false_type{});
    if constexpr ((dbg & 8) != 0) asm volatile("s_nop 0" : "+v"(acc1[3]));
    stamp();
    gemm2((c - 1) & 1, c + 1 < nch ? rsrc(a.w1p, c + 1) : rsrc(a.w2p, nch - 1), std::true_type{},
          [&](int s, int cf) { swish_piece(s, cf, c & 1); });
    stamp();
  }
  barrier();
  read_h((nch - 1) & 1, 0, h0);
  if constexpr (POST != 0)
    gemm2((nch - 1) & 1, rsrc(a.post_w, 0), std::true_type{}, [&](int, int) {});  // (the ring leaves with the walked matrix's head)
  else
    gemm2((nch - 1) & 1, rsrc(a.w2p, 0), std::false_type{}, [&](int, int) {});
  if constexpr ((dbg & 8) != 0) asm volatile("s_nop 0" : "+v"(acc2[3][3]));
  stamp();

  // ---- epilogue: x' = scale * acc2, then the LayerNorm(s) behind the module
  f32x4 xin[4][4];
#pragma unroll
  for (int cf = 0; cf < 4; ++cf)
#pragma unroll
    for (int rf = 0; rf < 4; ++rf) xin[cf][rf] = acc2[cf][rf] * scale;
  auto layer_norm = [&](int gi, int set) {  // in place; weight / bias = vectors gi, gi + 1 of PAR_OFF
    float mean[4], rstd[4];
    ln_stats(xin, set, mean, rstd);
#pragma unroll
    for (int cf = 0; cf < 4; ++cf) {
      const f32x4 g4 = parv(gi, cf), b4 = parv(gi + 1, cf);
#pragma unroll
      for (int rf = 0; rf < 4; ++rf) xin[cf][rf] = (xin[cf][rf] - mean[rf]) * (g4 * rstd[rf]) + b4;
    }
  };
  auto store_f32 = [&](float* __restrict__ dst) {
#pragma unroll
    for (int rf = 0; rf < 4; ++rf) {
      const int m = m0 + rf * 16 + lr;
      if (m < M) {
        float* o = dst + (size_t)m * D + col0;
#pragma unroll
        for (int cf = 0; cf < 4; ++cf) *(f32x4*)(o + 16 * cf) = xin[cf][rf];
      }
    }
  };
  if constexpr (LNMODE == 1) {
    store_f32(a.x);
    if constexpr (POST == 2) {
      if (a.xn_out != nullptr) {  // (uniform) a SIBLING LayerNorm of the same rows leaves as xn_out (vectors g2 / be2): E-Branchformer's
        float mean[4], rstd[4];   // norm_mlp beside norm_mha (e_branchformer_encoder.py:138-139) - one set of statistics, two affine maps
        ln_stats(xin, 1, mean, rstd);
        bf16* const out = (bf16*)a.xn_out;
#pragma unroll
        for (int cf = 0; cf < 4; ++cf) {
          const f32x4 g4 = parv(6, cf), b4 = parv(7, cf);
#pragma unroll
          for (int rf = 0; rf < 4; ++rf) {
            const int m = m0 + rf * 16 + lr;
            const f32x4 y = (xin[cf][rf] - mean[rf]) * (g4 * rstd[rf]) + b4;
            if (m < M) *(bf16x4*)(out + (size_t)m * D + col0 + 16 * cf) = __builtin_convertvector(y, bf16x4);
          }
        }
#pragma unroll
        for (int cf = 0; cf < 4; ++cf) {
          const f32x4 g4 = parv(4, cf), b4 = parv(5, cf);
#pragma unroll
          for (int rf = 0; rf < 4; ++rf) xin[cf][rf] = (xin[cf][rf] - mean[rf]) * (g4 * rstd[rf]) + b4;
        }
      } else {
        layer_norm(4, 1);
      }
    } else {
      layer_norm(4, 1);
    }
  } else {
    layer_norm(4, 1);
    store_f32(a.x);
    layer_norm(6, 0);
    if (a.out_f32) store_f32(a.out_f32);
  }
  if constexpr (POST != 2) {
    bf16* const out = (bf16*)a.xn_out;
#pragma unroll
    for (int rf = 0; rf < 4; ++rf) {
      const int m = m0 + rf * 16 + lr;
      if (m < M) {
        bf16* o = out + (size_t)m * D + col0;
#pragma unroll
        for (int cf = 0; cf < 4; ++cf) {
          *(bf16x4*)(o + 16 * cf) = __builtin_convertvector(xin[cf][rf], bf16x4);
        }
      }
    }
  }
  stamp();
  if constexpr (POST == 2) {
    // ---- q | k | v of the attention that follows (RelPositionMultiHeadedAttention.forward_qkv, transformer/attention.py:77-98)
    // as a walk behind norm_mha: the 64 normalised rows go back into the LDS tile (bf16: what the projection GEMMs read from
    // `xn`), [linear_q | linear_k | linear_v] (1 536 x 512, twelve chunks of 128 rows in the w1p layout) is walked in the GEMM-1
    // form and every chunk's results leave in the layouts csrc/attention2.hip reads: q, k [B][heads][Tpad][64] (a lane's four
    // columns are four consecutive dims of one head: 8-byte stores), V^T [B][512][Tpad] (2-byte stores: T is odd in general).
    // Replaces two tiled-GEMM launches per block (26.0 + 21.4 us at B = 64) and the store + reload of LN(x).
#pragma unroll
    for (int cf = 0; cf < 4; ++cf) {
      const int q = 8 * wave + 2 * cf + (lg >> 1);
      const int pos = (q & ~15) | ((q ^ lr) & 15);
#pragma unroll
      for (int rf = 0; rf < 4; ++rf)
        *(bf16x4*)(smem + ACT_OFF + (rf * 16 + lr) * 1024 + pos * 16 + (lg & 1) * 8) = __builtin_convertvector(xin[cf][rf], bf16x4);
    }
    barrier();
    read_act(0, af0);
    const int Tn = a.post_T, Tp = a.post_Tpad;
    size_t qk_off[4], vt_off[4];  // element offsets of this lane's four rows: (b * 8 heads * Tpad + t) * 64, b * 512 * Tpad + t
    bool rok[4];
#pragma unroll
    for (int rf = 0; rf < 4; ++rf) {
      const int m = m0 + rf * 16 + lr;
      rok[rf] = m < M;
      const int mm = rok[rf] ? m : M - 1;
      const int bb = mm / Tn, tt = mm - bb * Tn;
      qk_off[rf] = ((size_t)bb * (D / 64) * Tp + tt) * 64;
      vt_off[rf] = (size_t)bb * D * Tp + tt;
    }
    const float* cbp = a.post_b + 16 * wave + 4 * lg;
    const int nc = a.post_chunks;  // 12
    float4 cb = *(const float4*)cbp;
#pragma unroll 1
    for (int c = 0; c < nc; ++c) {
      const float4 cbn = *(const float4*)(cbp + (c + 1 < nc ? c + 1 : c) * CH);
      gemm1(rsrc(a.post_w, c + 1 < nc ? c + 1 : c), std::false_type{}, 0, std::true_type{});
      const int j0 = (c & 3) * CH + 16 * wave + 4 * lg;  // this lane's first column inside q, k or v (a chunk never straddles two)
      if (c < 8) {
        bf16* const base = (bf16*)(c < 4 ? a.post_q : a.post_k) + (size_t)(j0 >> 6) * Tp * 64 + (j0 & 63);
#pragma unroll
        for (int rf = 0; rf < 4; ++rf) {
          const f32x4 v = acc1[rf] + (f32x4){cb.x, cb.y, cb.z, cb.w};
          if (rok[rf]) *(bf16x4*)(base + qk_off[rf]) = __builtin_convertvector(v, bf16x4);
        }
      } else {
        bf16* const base = (bf16*)a.post_vt + (size_t)j0 * Tp;
#pragma unroll
        for (int rf = 0; rf < 4; ++rf) {
          const f32x4 v = acc1[rf] + (f32x4){cb.x, cb.y, cb.z, cb.w};
          if (rok[rf]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) base[vt_off[rf] + (size_t)r * Tp] = (bf16)v[r];
          }
        }
      }
      cb = cbn;
    }
  }
  if constexpr (POST == 1) {
    {
      // ---- the CTC head's arg-max (asr/ctc.py:207-215: argmax over ctc_lo of the encoder output) as a walk behind the last
      // launch of the stack: the 64 finished rows go back into the LDS tile (bf16: what the stand-alone GEMM reads from
      // enc_act) and the [V rounded up to 128][512] matrix is walked in the GEMM-1 form, every lane keeping the running
      // (maximum, label) of its rows over the labels 128 c + 16 wave + 4 lg + r it sees - ascending, strict >: the lowest
      // label wins a tie, like torch.argmax.  The logits never exist; the stand-alone arg-max GEMM (201 us at B = 64: 5 000
      // tiles of eight K-steps with a reduction epilogue each) and its partials pass are not launched.
#pragma unroll
      for (int cf = 0; cf < 4; ++cf) {
        const int q = 8 * wave + 2 * cf + (lg >> 1);
        const int pos = (q & ~15) | ((q ^ lr) & 15);
#pragma unroll
        for (int rf = 0; rf < 4; ++rf)
          *(bf16x4*)(smem + ACT_OFF + (rf * 16 + lr) * 1024 + pos * 16 + (lg & 1) * 8) = __builtin_convertvector(xin[cf][rf], bf16x4);
      }
      barrier();
      read_act(0, af0);
      float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int bidx[4] = {0, 0, 0, 0};
      const float* cbp = a.post_b + 16 * wave + 4 * lg;
      const int nc = a.post_chunks;
      float4 cb = *(const float4*)cbp;
#pragma unroll 1
      for (int c = 0; c < nc; ++c) {
        const float4 cbn = *(const float4*)(cbp + (c + 1 < nc ? c + 1 : c) * CH);
        gemm1(rsrc(a.post_w, c + 1 < nc ? c + 1 : c), std::false_type{}, 0, std::true_type{});
        const int id0 = c * CH + 16 * wave + 4 * lg;
#pragma unroll
        for (int rf = 0; rf < 4; ++rf) {
          const f32x4 v = acc1[rf] + (f32x4){cb.x, cb.y, cb.z, cb.w};
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (v[r] > best[rf]) {
              best[rf] = v[r];
              bidx[rf] = id0 + r;
            }
        }
        cb = cbn;
      }
      // the four lane groups of a row (labels differ), then the eight waves through LDS; larger value, lower label on ties
      float2* const red = (float2*)(smem + RED_OFF);
#pragma unroll
      for (int rf = 0; rf < 4; ++rf) {
#pragma unroll
        for (int step = 0; step < 2; ++step) {
          const auto pv = step == 0 ? __builtin_amdgcn_permlane16_swap(__float_as_uint(best[rf]), __float_as_uint(best[rf]), false, false)
                                    : __builtin_amdgcn_permlane32_swap(__float_as_uint(best[rf]), __float_as_uint(best[rf]), false, false);
          const auto pi = step == 0 ? __builtin_amdgcn_permlane16_swap((unsigned)bidx[rf], (unsigned)bidx[rf], false, false)
                                    : __builtin_amdgcn_permlane32_swap((unsigned)bidx[rf], (unsigned)bidx[rf], false, false);
          const float va = __uint_as_float(pv[0]), vb = __uint_as_float(pv[1]);
          const int ia = (int)pi[0], ib = (int)pi[1];
          const bool take_b = vb > va || (vb == va && ib < ia);
          best[rf] = take_b ? vb : va;
          bidx[rf] = take_b ? ib : ia;
        }
        if (lg == 0) red[wave * 64 + rf * 16 + lr] = make_float2(best[rf], __int_as_float(bidx[rf]));
      }
      barrier();
      if (tid < RB) {
        float bv = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          const float2 p = red[w * 64 + tid];
          const int pi = __float_as_int(p.y);
          if (p.x > bv || (p.x == bv && pi < bi)) {
            bv = p.x;
            bi = pi;
          }
        }
        if (m0 + tid < M) a.post_ids[m0 + tid] = bi;
      }
    }
  }
}

}  // namespace

extern "C" int em_ffn_rows_fused(const EmFfnRowsArgs* a, void* stream) {
  const bool post_qkv = a && a->post_q != nullptr;
  if (!a || !a->x || !a->w1p || !a->b1 || (!a->xn_out && !post_qkv)) return EM_ERR_BAD_ARG;
  if (a->main == EM_ROWS_FFN && (!a->w2p || !a->b2 || !a->g1 || !a->be1)) return EM_ERR_BAD_ARG;
  const bool pre = a->pre_in != nullptr;
  // (round 6: no projection, no xn_in, but pre_g / pre_be - the module's input is LayerNorm(x; pre_g, pre_be), computed in the launch)
  const bool lnin = !pre && !a->xn_in && a->pre_g && a->pre_be;
  if (pre ? (!a->pre_w || !a->pre_b || !a->pre_g || !a->pre_be) : (!a->xn_in && !lnin)) return EM_ERR_BAD_ARG;
  if (lnin && (a->main != EM_ROWS_FFN || a->ln_mode != 2 || a->post_w)) return EM_ERR_UNSUPPORTED;
  if (a->M <= 0 || a->ff <= 0) return EM_ERR_BAD_ARG;
  if (a->d != D || a->ff % CH != 0 || a->ff < 2 * CH) return EM_ERR_UNSUPPORTED;
  if (post_qkv) {  // q | k | v walked behind a ln_mode 1 launch without a projection in front
    if (a->xn_out && (!a->g2 || !a->be2)) return EM_ERR_BAD_ARG;  // (xn_out then = LayerNorm(x; g2, be2), a sibling of norm_mha)
    if (a->main != EM_ROWS_FFN || a->ln_mode != 1 || pre || !a->post_w || !a->post_b || !a->post_k || !a->post_vt ||
        a->post_chunks != 3 * D / CH || a->post_T <= 0 || a->post_Tpad < a->post_T || a->M % a->post_T != 0)
      return EM_ERR_BAD_ARG;
  } else if (a->post_w && (a->main != EM_ROWS_FFN || a->ln_mode != 2 || !a->post_b || !a->post_ids || a->post_chunks <= 0)) {
    return EM_ERR_BAD_ARG;
  }
  const bool glu = a->main == EM_ROWS_GLU;
  if (a->main != EM_ROWS_FFN && !glu) return EM_ERR_BAD_ARG;
  if (glu && !pre) return EM_ERR_BAD_ARG;
  if (glu && a->ff != 2 * D) return EM_ERR_UNSUPPORTED;
  if (!glu && a->ln_mode != 1 && a->ln_mode != 2) return EM_ERR_BAD_ARG;
  if (!glu && a->ln_mode == 2 && (!a->g2 || !a->be2)) return EM_ERR_BAD_ARG;
  const dim3 grid(em_cdiv(a->M, RB));
  static long long* stamps = nullptr;
  const bool want_stamps = (EM_FFN_DBG & 8) && em_sw().ffn_stamps;
  if (want_stamps && !stamps && hipMalloc((void**)&stamps, 64 * sizeof(long long)) != hipSuccess) return EM_ERR_LAUNCH;
  if (want_stamps && hipMemsetAsync(stamps, 0, 64 * sizeof(long long), (hipStream_t)stream) != hipSuccess) return EM_ERR_LAUNCH;
  long long* const st = want_stamps ? stamps : nullptr;
  typedef void (*kern_t)(const EmFfnRowsArgs, long long*);
  static EmLdsCap caps[9] = {};
  const int which = lnin ? 8 : post_qkv ? 7 : glu ? 4 : a->post_w ? 5 + (pre ? 1 : 0) : (a->ln_mode - 1) * 2 + (pre ? 1 : 0);
  const kern_t kern = which == 0   ? ffn_rows_kernel<1, 0, EM_ROWS_FFN>
                      : which == 1 ? ffn_rows_kernel<1, 1, EM_ROWS_FFN>
                      : which == 2 ? ffn_rows_kernel<2, 0, EM_ROWS_FFN>
                      : which == 3 ? ffn_rows_kernel<2, 1, EM_ROWS_FFN>
                      : which == 4 ? ffn_rows_kernel<1, 1, EM_ROWS_GLU>
                      : which == 5 ? ffn_rows_kernel<2, 0, EM_ROWS_FFN, 1>
                      : which == 6 ? ffn_rows_kernel<2, 1, EM_ROWS_FFN, 1>
                      : which == 7 ? ffn_rows_kernel<1, 0, EM_ROWS_FFN, 2>
                                   : ffn_rows_kernel<2, 2, EM_ROWS_FFN>;
  if (em_raise_lds_cap((const void*)kern, SMEM_BYTES, &caps[which]) != EM_OK) return EM_ERR_LAUNCH;
  const bool rec = em_prof_begin(stream);
  hipLaunchKernelGGL(kern, grid, dim3(NT), SMEM_BYTES, (hipStream_t)stream, *a, st);
  if (want_stamps) {
    long long hs[64];
    if (hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost) == hipSuccess) {
      printf("[ffn_rows<%d,%d> stamps, cycles since entry]", a->ln_mode, (int)pre);
      for (int i = 1; i < 64 && hs[i]; ++i) printf(" %lld", hs[i] - hs[0]);
      printf("\n");
      fflush(stdout);
    }
  }
  if (rec)
    em_prof_end(stream, (glu ? 2.0 : 4.0) * a->M * (double)D * a->ff + (pre ? 2.0 * a->M * (double)D * D : 0.0) +
                            (a->post_w ? 2.0 * a->M * (double)D * (post_qkv ? 3 * D : a->post_vocab) : 0.0), EM_PROF_ROWS);
  EM_CHECK_LAUNCH();
  return EM_OK;
}
