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
// Round 4: block<C|D|...> FOLDS the C part into the launch that follows it (two launches per Conformer block): the
// depthwise conv needs the GLU output of 15 frames either side of the workgroup's 32, so the workgroup computes the C
// part for 64 frames (four 16-frame fragments: its own two and a halo fragment either side - the 12 weight units are
// ingested once, at 32 MFMAs per unit the matrix cores now take as long as the ingest), keeps its own rows' residual
// in registers and writes the GLU output straight into the conv's LDS tile.  The `glu` round trip through HBM, twelve
// launches per step and their cold prologues go; the residual stream is read from `x` and written to `x_out`
// (another workgroup of the same launch reads this one's rows as ITS halo).
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
// reductions, the FFN hidden activation exists only as two [32][64] LDS tiles, and the only streams are
// the weights, read by all workgroups of the launch in near lockstep: straight from global memory into MFMA
// operand registers, from a fragment-major packing, through a ring of four 32 KiB units per wave, out of an
// L2 that the workgroups of the XCD warmed together during the prologue.
//
// Orientation.  Every GEMM is computed transposed, C^T[n][m] = sum_k W[n][k] A[m][k]: the weight
// fragment is the MFMA A operand, the activation fragment the B operand, so a lane ends up with four
// CONSECUTIVE output columns n of one row m (C/D layout: row = (lane >> 4) * 4 + r, col = lane & 15).
// That makes residual / bias / LayerNorm / bf16 packing 16- or 8-byte operations and lets the
// activation fragments (32 rows x K = 256) stay in registers for a whole GEMM.
//
// Four waves, wave w owning the 16-column slice nf = w of every 64-column group and BOTH 16-frame halves.  A
// "K unit" is 64 weight rows x K = 256 (per wave one weight fragment column x two frame fragments: 16 MFMAs); a
// "W2 unit" is all 256 rows x a 64-deep K slice of the FFN's second matrix (four weight fragments x two frame
// fragments x two k-steps: 16 MFMAs).  Both are 32 contiguous KiB, packed by the host so that the eight MFMA
// operands a wave takes from a unit are eight contiguous 1 KiB lines (include/espnet_amd.h, EmBlockArgs).  One CU
// ingests 64 B/clk, i.e. 512 cycles per unit against 256 cycles of MFMA: the kernels are built around keeping
// that stream busy (requests three units ahead, unconditional so that hipcc can count its waits, barriers only
// where waves exchange data).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "em_common.h"
#include "switches.h"

namespace {

constexpr int D = 256;
constexpr int BM = 32;
constexpr int NT = 256;                      // threads per workgroup: 4 waves, wave w = 16-column slice nf = w
constexpr int NC = 256;
constexpr int UNIT = 32768;                  // bytes of one weight unit (64 rows x K = 256, or 256 rows x 64-deep K slice)
constexpr int PAR_FLOATS = EM_BLOCK_PARAM_GROUP;
constexpr int PAR_BYTES = PAR_FLOATS * 4;
constexpr int MAX_GROUPS = 4;
constexpr int PAR_OFF = 0;                                   // up to 4 x 7 KiB: every bias / LayerNorm group of the launch
constexpr int RED_OFF = PAR_OFF + MAX_GROUPS * PAR_BYTES;    // 2 x 1 KiB: LayerNorm partials [2][4][32], alternating between consecutive LayerNorms
constexpr int TOUCH_OFF = RED_OFF + 2048;                    // 1 KiB: where the L2 warm-up's LDS-DMA lands (never read)
constexpr int TILE_OFF = TOUCH_OFF + 1024;                   // 32 KiB: the depthwise conv's 62-row input tile; during an FFN the parked residual
constexpr int ABUF_OFF = TILE_OFF + 32768;                   // 16 KiB: LN(x) as four [32][64] k-tiles
constexpr int WK_OFF = ABUF_OFF + 16384;                     // 32 KiB: depthwise conv weights [31][256] f32 (tap-major) + bias [256]
constexpr int XCH_OFF = WK_OFF + 32768;                      // 48 KiB more: with ABUF and WK the 96 KiB of the FFN's cross-wave reduction
constexpr int XSLOT = 8192;                                  // one (destination wave, source wave) slot: [4 fragments][2 halves][64 lanes] f32x4
constexpr int SMEM_BYTES = XCH_OFF + 49152;                  // 159 KiB
static_assert(ABUF_OFF % 1024 == 0 && TILE_OFF % 1024 == 0 && SMEM_BYTES <= 160 * 1024, "LDS layout");
constexpr int KW_MAX = 31, TROWS_MAX = BM + KW_MAX - 1;  // depthwise conv: kernel width is a template argument (31: the
                                                          // Conformer recipes, 15: the streaming recipe), 62-row input tile at most
// Row pitch of the conv tile: 512 + 16 bytes.  The conv reads a row's 256 channels contiguously (any pitch will do);
// the folded C part WRITES it from the GEMM layout - lane (lr, lg) 8 bytes of row lr - and with the dense pitch the 16
// rows of a fragment fall on the same banks (16-way conflict); 132 words per row step them by 4 banks: conflict-free.
constexpr int TPITCH = 528;
static_assert(TROWS_MAX * TPITCH <= 32768, "conv tile");
// The folded C part borrows the FFN's exchange area (idle until the first FFN): its parameter group and LN(x) of its 64
// rows as four [64][64] bf16 k-tiles.
constexpr int CPAR_OFF = XCH_OFF;
constexpr int ABUF64_OFF = XCH_OFF + 8192;
static_assert(PAR_BYTES <= 8192 && ABUF64_OFF + 32768 <= SMEM_BYTES, "fold layout");

// weight units are read through GLOBAL-address-space pointers: through a generic pointer the loads become flat_load,
// which counts on lgkmcnt too and forces vmcnt(0) (flat loads may return out of order)
typedef const __attribute__((address_space(1))) unsigned char* GU8;
typedef const __attribute__((address_space(1))) bf16x8* GFRAG;
constexpr int BAR_UNIT = 1, BAR_PARAMS = 2, BAR_TILE = 4, BAR_LAST = 8;  // what a barrier is for (documentation)

// A unit of the weight stream is 32 contiguous KiB, FRAGMENT-MAJOR: [wave nf][fragment q][lane][16 B]
// (host: pack_k_units / pack_w2 in asr/encoder/conformer_encoder.py), so that one wave-wide load instruction reads 1 KiB
// contiguous.  tools/experiments/direct_frag_bench.hip: such loads stream 128 GB/s per CU (0.257 us per unit) WITH the
// unit's 16 MFMAs and a Swish epilogue hidden behind them; the same fragments fetched from row-major weights reach 37
// GB/s, and the LDS-DMA ring this kernel used before (DMA write + fragment read = 64 KB of LDS traffic per unit, a
// barrier per unit, four loader waves) ~0.43 us per unit.
#ifndef EM_BLOCK_DBG
#define EM_BLOCK_DBG 0  // developer builds: 1 = no MFMA / epilogue work (what does the streaming cost alone?)
#endif

#ifndef EM_BLOCK_VAR
#define EM_BLOCK_VAR 0  // developer A/B builds (tools/build_block_variants.sh)
#endif
#ifndef EM_BLOCK_FINE
#define EM_BLOCK_FINE 0  // 1: EM_BLOCK_STAMPS records every wave of workgroup (3, 5) at sub-stage granularity
#endif

#define EM_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// wave-uniform values handed to inline asm through "s" constraints: made scalar HERE (folds away when the value already
// lives in SGPRs; a developer build with extra stamp code once kept `wave`-derived addresses in VGPRs and the asm did not assemble)
__device__ __forceinline__ const unsigned char* em_uniform_ptr(const unsigned char* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const unsigned char*)(((unsigned long long)hi << 32) | lo);
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): loop indices that must be constant EXPRESSIONS (asm "n" operands)
template <int N, typename F>
__device__ __forceinline__ void em_static_for(F&& f) {
  if constexpr (N > 0) {
    em_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// KWT: depthwise-conv kernel width (D part).  RELU: the feed-forward activation is ReLU (the contextual-block streaming
// encoder, contextual_block_conformer_encoder.py:148-154) instead of Swish; the conv module's Swish is unaffected.
template <int MODE, int KWT = 31, bool RELU = false>
__global__ __launch_bounds__(NT, 1) void block_kernel(const EmBlockArgs a_in, long long* __restrict__ stamps) {
  // Round 6, streaming (RELU) instantiations with a.ffn_split = S > 1: the FFN's hidden dimension is dealt to S workgroups
  // per 32-row block (grid z): each streams 1 / S of w_1 / w_2 and leaves its partial sum in memory; the LAST of them to
  // arrive (a ticket: nobody ever waits) adds the S partial sums in split order and carries the rows on.  A tick of a few
  // streams runs every launch on a few of the chip's 256 CUs, each streaming the whole 2 MiB of an FFN through ONE CU.
  const EmBlockArgs& a = a_in;
  const void *ffm_w1 = a_in.ffm_w1, *ffm_w2 = a_in.ffm_w2, *ff_w1 = a_in.ff_w1, *ff_w2 = a_in.ff_w2;
  const float *ffm_b1g = a_in.ffm_b1g, *ff_b1g = a_in.ff_b1g;
  int nsplit = 1;
  if constexpr (RELU && (MODE & (EM_BLOCK_A | EM_BLOCK_D)) != 0) {  // (the launches with an FFN)
    nsplit = a_in.ffn_split > 1 ? a_in.ffn_split : 1;
    if (nsplit > 1) {
      const int nch_all = ((a_in.ff >> 6) + 1) & ~1, nch_s = nch_all / nsplit, sidx = blockIdx.z;
      const size_t wo = (size_t)sidx * nch_s * UNIT;  // (w_1: nch_s units; w_2: nch_s / 2 pairs of 2 units: the same bytes)
      if (ffm_w1) { ffm_w1 = (const unsigned char*)ffm_w1 + wo; ffm_w2 = (const unsigned char*)ffm_w2 + wo; }
      if (ff_w1) { ff_w1 = (const unsigned char*)ff_w1 + wo; ff_w2 = (const unsigned char*)ff_w2 + wo; }
      if (ffm_b1g) ffm_b1g += sidx * nch_s * 64;
      if (ff_b1g) ff_b1g += sidx * nch_s * 64;
    }
  }
  bool ffn_exit = false;  // (RELU, split: this workgroup was not the last to arrive at an FFN's meeting point)
  constexpr int KW = KWT, HALF = KWT / 2, TROWS = BM + KWT - 1;
  static_assert(KWT == 31 || KWT == 15, "depthwise conv width");
  constexpr int dbg = EM_BLOCK_DBG;
  using MM = Mma<bf16>;
  constexpr bool HAS_C = (MODE & EM_BLOCK_C) != 0, HAS_D = (MODE & EM_BLOCK_D) != 0;
  constexpr bool HAS_A = (MODE & EM_BLOCK_A) != 0, FINAL = (MODE & EM_BLOCK_FINAL) != 0;
  constexpr bool CTC = (MODE & EM_BLOCK_CTC) != 0;
  constexpr bool FOLD = HAS_C && HAS_D;  // the C part computed in this launch, for 64 frames (round 4)
  constexpr bool ATT = (MODE & EM_BLOCK_ATT) != 0;  // the attention of the workgroup's 32 queries in front of the C part (round 6)
  static_assert(!ATT || (HAS_C && !HAS_D && !HAS_A && !FINAL), "block<ATT|C> is the only instantiation with the attention phase");
  static_assert(!FOLD || KWT == 31, "the folded C part is written for the 15-frame halo of k = 31");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const abuf = smem + ABUF_OFF;
  float* const red0 = (float*)(smem + RED_OFF);
  float* const par = (float*)(smem + PAR_OFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nf = wave & 3;  // this wave's 16-column slice of every 64-column group
  const int lr = lane & 15, lg = lane >> 4;
  int b = blockIdx.y, t0 = blockIdx.x * BM;
  const int T = a.T;
  if constexpr (ATT) {
    // The eight 32-query workgroups of an utterance read the same K / V^T (4 x 64 KiB) and the same position rows: they should
    // share an L2.  Consecutive linear workgroup ids go round-robin over the 8 XCDs (see touch() below: driver behaviour, used
    // for speed only), so inside a group of 8 utterances the id is dealt as (utterance = id % 8, query block = id / 8): every
    // XCD then serves whole utterances.  A last group of fewer than 8 utterances keeps the plain order.
    const int gx = gridDim.x, lid = blockIdx.y * gx + blockIdx.x, grp = lid / (8 * gx);
    if ((int)blockIdx.y < a.B && grp * 8 + 8 <= a.B) {
      b = grp * 8 + (lid & 7);
      t0 = ((lid - grp * 8 * gx) >> 3) * BM;
    }
  }

  // this lane's two frames: mi * 16 + lr
  bool row_ok[2];
  size_t mrow[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int t = t0 + mi * 16 + lr;
    row_ok[mi] = t < T;
    mrow[mi] = (size_t)b * T + (t < T ? t : T - 1);  // clamped: tail rows recompute frame T-1
  }
  const int ncol = nf * 16 + lg * 4;                   // + 64 f: the four consecutive columns of fragment f
  const int swz = lr & 7;
  // byte offset of (frame lr, columns ncol .. ncol + 3) inside a [32][64] bf16 k-tile of abuf (+ 2048 for frame 16 + lr)
  const int tile_wr = lr * 128 + (((2 * nf + (lg >> 1)) ^ swz) << 4) + (lg & 1) * 8;
  // 64-wide chunks of the FFN hidden dimension, rounded up to whole PAIRS (the host pads with a zero chunk); with a split
  // FFN (streaming instantiations) this workgroup's share of them
  const int nch = (((a.ff >> 6) + 1) & ~1) / nsplit;
  // developer profiling (EM_BLOCK_STAMPS / EM_BLOCK_DBG, tools/block_bench.py): stage-level cycle stamps of thread 0
  // of workgroup (0, 0); dbg 1 = no MFMA / epilogue work, dbg 2 = no DMA
  int nts = 0;
  // stamp(code): (code << 56) | cycle counter.  Codes: 1 kernel entry, 2 D prologue inputs requested, 3 inputs landed (tile
  // stored), 4 tile barrier passed, 5 conv computed, 6 conv output stored, 7 activations loaded, 10 pointwise_conv2 done,
  // 11-14 LayerNorm (partials written, first barrier, applied + tiles written, second barrier), 15 LayerNorm done /
  // activations loaded, 20 FFN h0 done, 21 FFN loop done, 22 FFN residual done, 30 norm_final done, 31 x stored,
  // 40 q k v done, 50 linear_out done, 51 pointwise_conv1 + GLU done
  auto stamp = [&](int code) __attribute__((always_inline)) {
    if constexpr (EM_BLOCK_FINE) {
      if (stamps && blockIdx.x == (gridDim.x > 3 ? 3 : 0) && blockIdx.y == (a.B > 5 ? 5 : 0) && blockIdx.z == 0 && lane == 0 && nts < 64)
        stamps[wave * 64 + nts] = ((long long)code << 56) | ((long long)__builtin_amdgcn_s_memtime() & 0xffffffffffffffll);
    } else {
      if (stamps && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z < 4 && tid == 0 && nts < 64)  // (split FFN: a row per share)
        stamps[blockIdx.z * 64 + nts] = ((long long)code << 56) | ((long long)__builtin_amdgcn_s_memtime() & 0xffffffffffffffll);
    }
    ++nts;
  };
  auto fstamp = [&](int code) __attribute__((always_inline)) {  // sub-stage stamps: fine builds only
    if constexpr (EM_BLOCK_FINE) stamp(code);
  };
  auto rstamp = [&](int code) __attribute__((always_inline)) {  // ... and the streaming instantiations (FFN h0 / loop / meeting point: 20, 21, 23, 24)
    if constexpr (EM_BLOCK_FINE || RELU) stamp(code);
  };
  stamp(1);
  int nbar = 0;  // barriers passed
  // every barrier goes through here: retire own LDS operations, synchronise
  auto bar = [&](int code) __attribute__((always_inline)) {  // `code` documents what the barrier is for
    (void)code;
    ++nbar;
    EM_LGKM0();
    __builtin_amdgcn_s_barrier();
  };

  // ---- row state ------------------------------------------------------------------------------
  float4 xr[2][4];     // residual x[frame mi][64 f + ncol .. + 3], f32
  bf16x8 act[2][8];    // activation fragments of the running GEMM: frame mi*16 + lr, k = 32 ks + 8 lg ..
  auto load_x = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int f = 0; f < 4; ++f) xr[mi][f] = *(const float4*)(a.x + mrow[mi] * D + 64 * f + ncol);
  };
  float* const xdst = FOLD ? a.x_out : a.x;  // (folded: neighbours read this workgroup's rows of x as their halo)
  auto store_x = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
      if (row_ok[mi]) {
#pragma unroll
        for (int f = 0; f < 4; ++f) *(float4*)(xdst + mrow[mi] * D + 64 * f + ncol) = xr[mi][f];
      }
  };
  auto load_act = [&]() __attribute__((always_inline)) {  // from abuf, after a barrier
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        act[mi][ks] = *(const bf16x8*)(abuf + ((ks >> 1) * 32 + mi * 16 + lr) * 128 + ((((ks & 1) * 4 + lg) ^ swz) << 4));
  };
  // LayerNorm statistics of the 32 rows.  A row is spread over 4 lane groups x 4 waves: every holder reduces its
  // 16 values to (sum, sum of squares about its own mean), the lane groups combine with Chan's update over two
  // shuffles, the four waves through ONE LDS exchange (instead of the mean pass + centred pass with a barrier
  // each; same result to f32 round-off: nothing is formed as a difference of large numbers).  One barrier.
  int nln = 0;  // LayerNorms so far: consecutive ones alternate the exchange buffer (there is no barrier between one's reads and the next one's writes)
  // (two halves so that the folded C part can run the statistics of its 64 rows - own rows and halo rows - through ONE barrier)
  auto ln_partials = [&](const float4 (&v)[2][4], float* red) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      float s = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f) s += (v[mi][f].x + v[mi][f].y) + (v[mi][f].z + v[mi][f].w);
      const float m0 = s * (1.0f / 16.0f);
      float q = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const float dx = v[mi][f].x - m0, dy = v[mi][f].y - m0, dz = v[mi][f].z - m0, dw = v[mi][f].w - m0;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
      // combine equal-sized groups: n doubles, M2 = M2a + M2b + (sa - sb)^2 / (2 n)
      // (the partner's values through gfx950's register swaps: own + partner's come back as a pair, and the update is
      // symmetric in them - one VALU instruction where __shfl_xor was a ds_bpermute round trip, eight per LayerNorm)
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
      // (all four lane groups of a row hold the same totals - the pair updates are symmetric - and all of them store:
      // no exec-masked branch, so the LayerNorm's first half is ONE scheduling region and the weight requests issued in
      // front of it can be dealt between its VALU instructions, see ln_to_act)
      red[nf * 32 + mi * 16 + lr] = s;        // sum over this wave's 64 columns
      red[128 + nf * 32 + mi * 16 + lr] = q;  // sum of squares about their mean
    }
  };
  auto ln_finish = [&](const float* red, float mean[2], float rstd[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = mi * 16 + lr;
      const float s0 = red[m], s1 = red[32 + m], s2 = red[64 + m], s3 = red[96 + m];
      const float q0 = red[128 + m], q1 = red[160 + m], q2 = red[192 + m], q3 = red[224 + m];
      const float sa = s0 + s1, sb = s2 + s3;
      const float qa = q0 + q1 + (s0 - s1) * (s0 - s1) * (1.0f / 128.0f);
      const float qb = q2 + q3 + (s2 - s3) * (s2 - s3) * (1.0f / 128.0f);
      const float qq = qa + qb + (sa - sb) * (sa - sb) * (1.0f / 256.0f);
      mean[mi] = (sa + sb) * (1.0f / D);
      // v_rsq_f32 (1 ulp) + one Newton step: f32-accurate without the IEEE sqrt + division sequences
      const float var = qq * (1.0f / D) + a.eps;
      const float y0 = __builtin_amdgcn_rsqf(var);
      rstd[mi] = y0 * (1.5f - 0.5f * var * y0 * y0);
    }
  };
  // `pre1()` / `pre2()` issue weight requests of the stage that FOLLOWS the LayerNorm (ring slots that are free by now):
  // pre1 rides in the LayerNorm's first half (N1 wave-wide loads between the ~120 VALU instructions of the partial sums),
  // pre2 in its second half (statistics, normalisation, bf16 tile).  Round 5: until then the requests were one cluster in
  // front of the LayerNorm - 24 or 32 loads of 1 KiB per wave, four waves: 1 500 - 2 000 cycles in the ISSUE alone (a CU
  // ingests 64 B/clk and the queue is short), during which the wave's LayerNorm arithmetic waited behind them although it
  // needs none of them: every hand-over between two GEMMs cost stream time + LayerNorm time (4 - 6 K cycles, profiles/
  // r03b_block_stamps_fine.txt) instead of the larger of the two.  sched_group_barrier pins the deal: left alone hipcc
  // clusters the loads first.
  auto no_pre = [] {};
  auto ln_stats = [&](float mean[2], float rstd[2], int code, auto&& pre1, auto n1c) __attribute__((always_inline)) {
    constexpr int N1 = decltype(n1c)::value;
    float* const red = red0 + (nln & 1) * 256;
    ++nln;
    pre1();
    ln_partials(xr, red);
    if constexpr (N1 > 0 && !(EM_BLOCK_VAR & 32)) {
#pragma unroll
      for (int i = 0; i < N1; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);               // one weight request
        __builtin_amdgcn_sched_group_barrier(0x002, N1 > 16 ? 4 : 7, 0);  // ... and its share of the partial sums
      }
    }
    fstamp(11);
    bar(code);
    fstamp(12);
    ln_finish(red, mean, rstd);
  };
  // y = LN(x; g, b) with g, b at float offsets go / bo of parameter buffer pb
  auto ln_apply_pre = [&](const float* pb, int go, int bo, float4 y[2][4], int code, auto&& pre1, auto n1c) __attribute__((always_inline)) {
    float mean[2], rstd[2];
    ln_stats(mean, rstd, code, pre1, n1c);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const float4 g4 = *(const float4*)(pb + go + 64 * f + ncol);
      const float4 b4 = *(const float4*)(pb + bo + 64 * f + ncol);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        y[mi][f].x = (xr[mi][f].x - mean[mi]) * rstd[mi] * g4.x + b4.x;
        y[mi][f].y = (xr[mi][f].y - mean[mi]) * rstd[mi] * g4.y + b4.y;
        y[mi][f].z = (xr[mi][f].z - mean[mi]) * rstd[mi] * g4.z + b4.z;
        y[mi][f].w = (xr[mi][f].w - mean[mi]) * rstd[mi] * g4.w + b4.w;
      }
    }
  };
  auto ln_apply = [&](const float* pb, int go, int bo, float4 y[2][4], int code) __attribute__((always_inline)) {
    ln_apply_pre(pb, go, bo, y, code, no_pre, std::integral_constant<int, 0>{});
  };
  // LN(x) -> bf16 -> abuf -> activation fragments.  One more barrier, carrying `code` (the parameter reads of
  // this LayerNorm precede it).
  auto ln_to_act_pre = [&](const float* pb, int go, int bo, int code, auto&& pre1, auto n1c, auto&& pre2, auto n2c) __attribute__((always_inline)) {
    constexpr int N2 = decltype(n2c)::value;
    float4 y[2][4];
    __builtin_amdgcn_sched_barrier(0);
    ln_apply_pre(pb, go, bo, y, 0, pre1, n1c);
    pre2();
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        bf16x4 pk = {(bf16)y[mi][f].x, (bf16)y[mi][f].y, (bf16)y[mi][f].z, (bf16)y[mi][f].w};
        *(bf16x4*)(abuf + f * 4096 + mi * 2048 + tile_wr) = pk;
      }
    if constexpr (N2 > 0 && !(EM_BLOCK_VAR & 32)) {
#pragma unroll
      for (int i = 0; i < N2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, N2 > 8 ? 6 : 10, 0);
      }
    }
    fstamp(13);
    bar(code);
    fstamp(14);
    load_act();
  };
  auto ln_to_act = [&](const float* pb, int go, int bo, int code) __attribute__((always_inline)) {
    ln_to_act_pre(pb, go, bo, code, no_pre, std::integral_constant<int, 0>{}, no_pre, std::integral_constant<int, 0>{});
  };
  // ---- units: weight fragments go global memory -> registers, THREE UNITS AHEAD of the MFMAs ------------
  // A unit is 32 contiguous KiB, fragment-major; this wave's eight fragments are eight 1 KiB lines at
  // unit + nf * 8 KiB + q * 1 KiB, one wave-wide 16-byte load each.  A single wave per SIMD has no other wave to
  // hide a load's latency with, so a stage keeps a ring of four fragment sets: unit u computes from ring[u & 3]
  // while units u + 1 .. u + 3 are in flight (their L2 / MALL latency is several unit times).  No table, no
  // barrier: a stage knows its matrix (a kernel argument in SGPRs) and counts units.
  struct WF {
    bf16x8 v[8];
  };
  WF ring[4];
  const unsigned voff = nf * 8192 + lane * 16;
  auto read_unit = [&](const void* base, int u, WF& w) __attribute__((always_inline)) {
    GU8 su = (GU8)base + (size_t)u * UNIT + voff;
#pragma unroll
    for (int q = 0; q < 8; ++q) w.v[q] = *(GFRAG)(su + q * 1024);
  };
  // request the first units of a stage's K stream into ring[0 ..]: called as early as the ring is free (before
  // the LayerNorm / convolution that precedes the stage)
  auto k_pre = [&](const void* base, int n) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 3; ++j) read_unit(base, j < n ? j : n - 1, ring[j]);
  };
  // N K units of `base` (compile-time count: fully unrolled, the body sees constants), the first min(3, N) already
  // requested by k_pre: body(fragments, u) per unit
  auto stream_k = [&](const void* base, auto count, auto&& body) __attribute__((always_inline)) {
    constexpr int N = decltype(count)::value;
#pragma unroll
    for (int u = 0; u < N; ++u) {
      if (u + 3 < N) read_unit(base, u + 3, ring[(u + 3) & 3]);
      body(ring[u & 3], u);
    }
  };
  // ---- L2 warm-up ------------------------------------------------------------------------------------------
  // Every workgroup of an XCD reads every weight unit of the launch at about the same moment, so each unit is an
  // L2 MISS for all of them (hit-on-miss at best): its latency is the MALL / HBM latency, and three units in
  // flight per wave then bound the stream at ~48 GB/s per CU where L2 hits run at the 128 GB/s ingest limit.  So
  // the workgroups of an XCD share out the launch's weights in 8 KiB chunks (64 lines: one wave-wide dword load,
  // a lane per line) and touch them all while the prologue's own global loads are in flight; the weights are in
  // the XCD's L2 by the time the streams ask for them.  A workgroup's share is its index among the launch's workgroups
  // on its XCD, taken as (linear workgroup id / 8): consecutive ids go round-robin over the 8 XCDs on every box
  // measured (tools/experiments/xcc_map.hip, profiles/r02t_xcc_map.txt).  That map is driver behaviour, not a
  // contract - under another map some chunks are touched twice and some not at all, which costs speed, never
  // correctness.  (Round 2 asked HW_REG_XCC_ID and a per-XCC arrival counter instead: equally fast,
  // profiles/r02u_static_vs_xcc_share.txt, but the returning atomic sat in front of every other request of the
  // prologue.)  Called as soon as the prologue's own requests are out.
  constexpr int MAXT = 16;  // chunks per wave at most: small grids warm what they can
  auto touch = [&]() __attribute__((always_inline)) {
#ifdef EM_BLOCK_NO_TOUCH
    return;
#endif
    const int nwg = gridDim.x * gridDim.y;
    const int per_xcc = (nwg + 7) >> 3;                      // workgroups of this launch on an XCC (balanced dispatch)
    const int nworker = per_xcc * 4;                         // their waves
    const int worker = (int)((blockIdx.y * gridDim.x + blockIdx.x) >> 3) * 4 + wave;
    const unsigned dst = TOUCH_OFF + wave * 256;
    const int loff = lane * 128;
    int total = (HAS_C ? 48 : 0) + (HAS_D ? 16 + 8 * nch : 0) + (HAS_A ? 48 + 8 * nch : 0);  // 8 KiB chunks of the launch
    if (CTC) total += a.ctc_units * 4;
#pragma unroll 1
    for (int i = 0; i < MAXT && worker + i * nworker < total; ++i) {
      int j = worker + i * nworker;                         // chunk of the launch's weight list (wave-uniform)
      const unsigned char* p = nullptr;
      auto take = [&](const void* base, int n) __attribute__((always_inline)) {
        if (p == nullptr) {
          if (j < n) p = (const unsigned char*)base + (size_t)j * 8192;
          else j -= n;
        }
      };
      if (HAS_C) {
        take(a.wout, 16);
        take(a.pw1f, 32);
      }
      if (HAS_D) {
        take(a.pw2, 16);
        take(ff_w1, nch * 4);
        take(ff_w2, nch * 4);
      }
      if (HAS_A) {
        take(ffm_w1, nch * 4);
        take(ffm_w2, nch * 4);
        take(a.wqkv, 48);
      }
      if (CTC) take(a.ctc_w, a.ctc_units * 4);
      // LDS-DMA (global_load_lds_dword: the word goes to LDS at M0 + 4 * lane, no destination register that a
      // late return could clobber), from inline asm: a C++ load with no use is dropped, one with a use only at
      // the kernel's exit is sunk down to it.  hipcc's wait counting does not see them: a wait for a load
      // issued BEFORE them is still exact (in-order return), a wait for a later one can only over-wait.
      unsigned keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %3\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dword %1, %2\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(loff), "s"(p), "s"(dst)
          : "memory");
    }
  };
  auto touch_done = [&]() __attribute__((always_inline)) {};
  // Round 5, launches that do not fill the chip (a tick of a batch of streams: 64 workgroups; one stream: 2): the launcher
  // adds HELPER workgroups (rows of the grid past a.B) on the idle CUs.  A helper takes its share of the warm-up and leaves.
  // Without them the few workgroups of an XCD touch the launch's whole list themselves - 10 to 16 wave-wide requests per wave,
  // 640 to 1 024 lines, queued IN FRONT of the wave's own first units in the same in-order return stream - and everything not
  // yet touched when a stream reaches it is a MALL-latency miss.  (The DMA lands in this workgroup's LDS: wait for it before
  // the allocation is released to whoever comes next on this CU.)
  if ((int)blockIdx.y >= a.B) {
    touch();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ---- prologue staging: parameter groups (and the depthwise conv's weights) go global memory -> LDS by LDS-DMA,
  // 1 KiB per wave-instruction, lines dealt round-robin to the four waves.  No registers, no wait at the point of
  // issue: the lines are requested FIRST, the stage's own inputs after them, and since memory operations complete
  // in issue order a wave's first wait for one of its inputs also covers its DMA lines; the first barrier then
  // publishes them to the other waves.  (Until round 3 this was a register copy that hipcc serialised into one
  // global round trip per 16 bytes - load, s_waitcnt vmcnt(0), ds_write, repeat - in front of every other request:
  // 6 dependent round trips in block<D|A>, next to the returning atomic of the warm-up and a vmcnt(0) for `tlens`.)
  // (n1k is a compile-time count and every wave issues ceil(n1k / 4) requests, the surplus ones repeating the last line:
  // no loop, no branch.  Behind a LOOP around the asm hipcc put an s_waitcnt vmcnt(0) at the loop exit, i.e. a full
  // round trip for the lines just requested.)
  auto dma_lines = [&](const void* g, unsigned lds_off, auto n1k_c) __attribute__((always_inline)) {
    constexpr int n1k = decltype(n1k_c)::value;
    const unsigned voff16 = lane * 16;
#pragma unroll
    for (int i = 0; i < (n1k + 3) / 4; ++i) {
      int j = wave + 4 * i;
      j = j < n1k ? j : n1k - 1;
      const unsigned char* p = em_uniform_ptr((const unsigned char*)g + (size_t)j * 1024);
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_off + j * 1024);
      unsigned keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %3\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %1, %2\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(voff16), "s"(p), "s"(dst)
          : "memory");
    }
  };

  // 16 MFMAs of a K unit: out[mi] = C^T[n = nf*16 + lg*4 + r][m = mi*16 + lr]
  // (swap: C[m = mi*16 + lg*4 + r][n = nf*16 + lr])
  auto mma_of = [&](const bf16x8 (&av)[2][8], const WF& w, bool swap, f32x4 out[2]) {
    if constexpr (dbg & 1) {
      out[0] = out[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      return;
    }
    f32x4 c[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) c[mi][0] = c[mi][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        c[mi][ks & 1] = swap ? MM::mma(av[mi][ks], w.v[ks], c[mi][ks & 1]) : MM::mma(w.v[ks], av[mi][ks], c[mi][ks & 1]);
    out[0] = c[0][0] + c[0][1];
    out[1] = c[1][0] + c[1][1];
  };
  auto mma_k = [&](const WF& w, bool swap, f32x4 out[2]) __attribute__((always_inline)) { mma_of(act, w, swap, out); };
  // ---- FFN: x += scale * (W2 . swish(W1 . act + b1) + b2); b1 at pb + b1o, b2 at pb + b2o -------------------------
  // Round 3: the hidden activation never leaves the wave that computed it.  Wave w owns the 16 hidden columns
  // 16 w .. 16 w + 15 of every 64-wide chunk c (the K unit c of W1: 16 MFMAs over K = 256, as before).  In the
  // transposed orientation its lane (lr, lg) ends up with h[frame lr][hidden 4 lg .. 4 lg + 3] - and that is exactly
  // a B-operand fragment of the second GEMM if the contraction index of that GEMM is ORDERED to match: for a PAIR
  // of chunks (2p, 2p + 1) the wave's 32-deep K slice is k = 8 lg + j  <->  hidden 64 (2p + (j >> 2)) + 16 w + 4 lg
  // + (j & 3), i.e. the lane's own eight values (bias + Swish + bf16 in registers, no LDS, no barrier).  The host
  // packs W2 in that order (pack_w2_pairs): per pair and wave 16 output fragments x 1 KiB = two 8 KiB sub-units.
  // Every wave therefore accumulates a partial sum of ALL 256 output columns over its own quarter of the hidden
  // dimension (16 x 2 accumulator fragments: AGPRs) and the four partials meet ONCE per FFN in LDS.  Until round 3
  // the four waves shared every chunk's H through LDS tiles: a barrier, an LDS round trip and a Swish epilogue per
  // 64-wide chunk, 16 times per FFN - stage stamps put that at ~400 of the 1 600 cycles per iteration, and the
  // barrier made every wave wait for the slowest wave's weight lines every 32 MFMAs (profiles/r03a_*).
  // The weight stream is unchanged in bytes: per pair and wave 2 x 8 KiB of W1 (ring[0], ring[1]) and 2 x 8 KiB of
  // W2 (ring[2], ring[3]); a ring slot is refilled as soon as its 16 MFMAs are issued, i.e. requested one whole
  // iteration (64 MFMAs) ahead of its use.  The host pads odd chunk counts with a zero chunk (nch is even here).
  auto read_w2 = [&](const void* w2, int p, int half, WF& w) __attribute__((always_inline)) {
    GU8 su = (GU8)w2 + (size_t)p * (2 * UNIT) + nf * 16384 + half * 8192 + lane * 16;
#pragma unroll
    for (int q = 0; q < 8; ++q) w.v[q] = *(GFRAG)(su + q * 1024);
  };
  auto ffn_pre = [&](const void* w1, const void* w2) __attribute__((always_inline)) {
    read_unit(w1, 0, ring[0]);
    read_unit(w1, 1, ring[1]);
    read_w2(w2, 0, 0, ring[2]);
    read_w2(w2, 0, 1, ring[3]);
  };
  // b1g != NULL: the first bias comes from GLOBAL memory (ff > 1024 does not fit the 7 KiB parameter group: the streaming
  // recipe's 2048), requested one pair ahead so that its latency is never in the chain
  // `nbase` (may be null): W1-style units 0, 1 of the stage behind this module - what the two requests that run past the
  // last chunk ask for (they used to repeat the last unit: issued unconditionally so that hipcc can count its waits, waited
  // for by nobody); `after()`: the requests for ring[2], ring[3], issued behind the module's last MFMAs.
  auto ffn = [&](const float* pb, int b1o, int b2o, float scale, const void* w1, const void* w2, bool repark,
                 const float* b1g, const void* nbase, auto&& after) __attribute__((always_inline)) {
    const int np = nch >> 1, lastc = nch - 1;
    // chunk c of W1, or - past the end - unit c - nch of the next stage
    auto read_w1 = [&](int c, WF& w) __attribute__((always_inline)) {
      const bool past = c >= nch;
      const void* b = (past && nbase) ? nbase : w1;
      const int u = past ? (nbase ? c - nch : lastc) : c;
      read_unit(b, u, w);
    };
    // (compile-time: only the streaming instantiations carry the global-bias path; in the Conformer kernels a run-time
    // choice cost 30 registers)
    constexpr bool B1G = RELU;
    float4 nb0 = make_float4(0.f, 0.f, 0.f, 0.f), nb1 = nb0;
    if constexpr (B1G) {
      nb0 = *(const float4*)(b1g + ncol);
      nb1 = *(const float4*)(b1g + 64 + ncol);
    }
    auto act_ = [](float v) { return RELU ? fmaxf(v, 0.f) : swishf_(v); };
    f32x4 acc2[16][2];
#pragma unroll
    for (int f = 0; f < 16; ++f) acc2[f][0] = acc2[f][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the residual is not needed until the end: its 32 registers are parked in LDS (the conv tile's space, lane-major:
    // conflict-free, private to the lane, no barrier)
    float4* const xpark = (float4*)(smem + TILE_OFF) + tid;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int f = 0; f < 4; ++f) xpark[(mi * 4 + f) * NT] = xr[mi][f];
    // bias + Swish + bf16 of the two chunks' 16 hidden columns of this wave -> the B fragments of the pair
    auto swish_pack = [&](const f32x4 h[2][2], int p, bf16x8 hb[2]) __attribute__((always_inline)) {
      float4 b0, b1;
      if constexpr (B1G) {
        b0 = nb0;
        b1 = nb1;
        const int pn = p + 1 < np ? p + 1 : p;  // (unconditional: past the end the last pair's again)
        nb0 = *(const float4*)(b1g + (2 * pn) * 64 + ncol);
        nb1 = *(const float4*)(b1g + (2 * pn + 1) * 64 + ncol);
      } else {
        b0 = *(const float4*)(pb + b1o + (2 * p) * 64 + ncol);
        b1 = *(const float4*)(pb + b1o + (2 * p + 1) * 64 + ncol);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        hb[mi] = (bf16x8){(bf16)act_(h[0][mi][0] + b0.x), (bf16)act_(h[0][mi][1] + b0.y),
                          (bf16)act_(h[0][mi][2] + b0.z), (bf16)act_(h[0][mi][3] + b0.w),
                          (bf16)act_(h[1][mi][0] + b1.x), (bf16)act_(h[1][mi][1] + b1.y),
                          (bf16)act_(h[1][mi][2] + b1.z), (bf16)act_(h[1][mi][3] + b1.w)};
    };
    // 16 MFMAs of a W2 sub-unit: output fragments 8 half .. 8 half + 7, both frame halves
    auto mma_w2 = [&](const WF& w, int half, const bf16x8 hb[2]) __attribute__((always_inline)) {
      if constexpr (dbg & 1) return;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        acc2[half * 8 + q][0] = MM::mma(w.v[q], hb[0], acc2[half * 8 + q][0]);
        acc2[half * 8 + q][1] = MM::mma(w.v[q], hb[1], acc2[half * 8 + q][1]);
      }
    };
    f32x4 h[2][2];
    bf16x8 hb[2];
    mma_k(ring[0], false, h[0]);
    read_w1(2, ring[0]);
    mma_k(ring[1], false, h[1]);
    read_w1(3, ring[1]);
    swish_pack(h, 0, hb);
    // every wave holds its activation fragments and has read the conv weights: ABUF / WK may receive partial sums
    // from a wave that finishes early (the only barrier of the FFN besides the reduction's)
    bar(0);
    rstamp(20);
    // The requests are UNCONDITIONAL (past the end they repeat the last unit, which nobody waits for): a load under
    // a branch makes hipcc's wait-count pass assume it may not have been issued, and every wait for an older load
    // degrades to "wait for everything" - the whole point of the ring.
#pragma unroll 1
    for (int p = 0; p + 1 < np; ++p) {
      mma_k(ring[0], false, h[0]);                             // chunk 2p + 2
      read_w1(2 * p + 4, ring[0]);
      mma_k(ring[1], false, h[1]);                             // chunk 2p + 3
      read_w1(2 * p + 5, ring[1]);
      mma_w2(ring[2], 0, hb);                                  // pair p
      read_w2(w2, p + 1, 0, ring[2]);
      mma_w2(ring[3], 1, hb);
      read_w2(w2, p + 1, 1, ring[3]);
      swish_pack(h, p + 1, hb);
    }
    mma_w2(ring[2], 0, hb);                                    // the last pair
    mma_w2(ring[3], 1, hb);
    after();
    rstamp(21);
    // ---- the four partial sums meet: wave w keeps the output fragments f = 4 f4 + w (its columns of the residual
    // layout) and hands the other twelve to their owners through 8 KiB slots (destination, source); summed in wave
    // order 0, 1, 2, 3 whatever the arrival order: deterministic
    float4 psum[2][4];  // (split FFN, streaming instantiations: this share's partial sum)
    auto reduce_as = [&](auto nfc) __attribute__((always_inline)) {
      constexpr int NF = decltype(nfc)::value;
#pragma unroll
      for (int dst = 0; dst < 4; ++dst) {
        if (dst == NF) continue;
        unsigned char* slot = smem + ABUF_OFF + (dst * 3 + (NF > dst ? NF - 1 : NF)) * XSLOT + lane * 16;
#pragma unroll
        for (int f4 = 0; f4 < 4; ++f4)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) *(f32x4*)(slot + (f4 * 2 + mi) * 1024) = acc2[4 * f4 + dst][mi];
      }
      bar(0);
#pragma unroll
      for (int f4 = 0; f4 < 4; ++f4) {
        const float4 b4 = *(const float4*)(pb + b2o + 64 * f4 + ncol);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          f32x4 sum;
          bool first = true;
#pragma unroll
          for (int src = 0; src < 4; ++src) {
            f32x4 v;
            if (src == NF) v = acc2[4 * f4 + NF][mi];
            else v = *(const f32x4*)(smem + ABUF_OFF + (NF * 3 + (src > NF ? src - 1 : src)) * XSLOT + lane * 16 + (f4 * 2 + mi) * 1024);
            sum = first ? v : sum + v;
            first = false;
          }
          if constexpr (RELU) {
            if (nsplit > 1) {  // (split FFN: the partial sum of THIS workgroup's share; finished below, behind the meeting point)
              psum[mi][f4] = make_float4(sum[0], sum[1], sum[2], sum[3]);
              continue;
            }
          }
          xr[mi][f4] = xpark[(mi * 4 + f4) * NT];
          xr[mi][f4].x += scale * (sum[0] + b4.x);
          xr[mi][f4].y += scale * (sum[1] + b4.y);
          xr[mi][f4].z += scale * (sum[2] + b4.z);
          xr[mi][f4].w += scale * (sum[3] + b4.w);
          if (repark) xpark[(mi * 4 + f4) * NT] = xr[mi][f4];  // the caller stores x from here at the very end of the kernel
        }
      }
    };
    switch (nf) {
      case 0: reduce_as(std::integral_constant<int, 0>{}); break;
      case 1: reduce_as(std::integral_constant<int, 1>{}); break;
      case 2: reduce_as(std::integral_constant<int, 2>{}); break;
      default: reduce_as(std::integral_constant<int, 3>{}); break;
    }
    if constexpr (RELU) {
      if (nsplit > 1) {
        // ---- the S shares of this row block meet.  Publish: plain stores of the partial sum (lane-major, as the residual is
        // parked), barrier, ONE lane releases at agent scope and takes a ticket; the workgroup that draws the last ticket
        // acquires, resets the ticket for the next launch and sums the S partial sums in split order (whatever the arrival
        // order: deterministic); the others are done.  Nobody waits for anybody: no deadlock whatever the residency.
        // (MI355X_MICROARCH.md, "valid forms": plain stores -> __syncthreads -> lane-0 release fence -> drained -> relaxed
        // agent atomic; consumer: one agent acquire -> __syncthreads -> plain loads.)
        const int wg = b * (int)gridDim.x + (int)blockIdx.x;
        float4* const part = (float4*)a.ffn_part + ((size_t)wg * nsplit) * 8 * NT;
        float4* const mine = part + (size_t)blockIdx.z * 8 * NT + tid;
        // Developer builds (profiles/r06u_stream_seam_ab*.txt): EM_BLOCK_VAR & 8192 = agent-scope (sc1) stores and no release
        // fence, & 4096 = agent-scope loads and no acquire fence.  One stream: 661 us per call as built, 679 - 697 with either;
        // 32 streams: 823 against 802 - 807 with the sc1 stores (the release fence writes back what the XCD's OTHER
        // workgroups have dirtied).  The fenced form is the documented one and stays.
        constexpr bool SEAM_ST_PLAIN = (EM_BLOCK_VAR & 8192) == 0, SEAM_LD_PLAIN = (EM_BLOCK_VAR & 4096) == 0;
        constexpr bool SEAM_REL = SEAM_ST_PLAIN, SEAM_ACQ = SEAM_LD_PLAIN;
        auto put = [&](float4* p, const float4& v) __attribute__((always_inline)) {
          if constexpr (SEAM_ST_PLAIN) {
            *p = v;
          } else {  // agent-scope relaxed stores (sc1: written through this XCD's L2)
            __hip_atomic_store((unsigned long long*)p, ((const unsigned long long*)&v)[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((unsigned long long*)p + 1, ((const unsigned long long*)&v)[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        };
        auto get = [&](const float4* p) __attribute__((always_inline)) -> float4 {
          if constexpr (SEAM_LD_PLAIN) {
            return *p;
          } else {  // agent-scope relaxed loads (sc1: past whatever this XCD's L2 holds of an earlier launch's partial sums)
            unsigned long long w[2];
            w[0] = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            w[1] = __hip_atomic_load((const unsigned long long*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float4 v;
            __builtin_memcpy(&v, w, 16);
            return v;
          }
        };
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int f4 = 0; f4 < 4; ++f4) put(mine + (mi * 4 + f4) * NT, psum[mi][f4]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (acknowledged: at the memory side for the sc1 stores)
        bar(0);
        int* const flag = (int*)(red0 + 480);  // (a word of the LayerNorm exchange area nobody uses at this point)
        if (tid == 0) {
          if constexpr (SEAM_REL) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          const int t = __hip_atomic_fetch_add(a.ffn_ticket + wg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (t == nsplit - 1) {
            __hip_atomic_store(a.ffn_ticket + wg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every share has arrived
            if constexpr (SEAM_ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          }
          *flag = t;
        }
        bar(0);
        if (*flag != nsplit - 1) {
          ffn_exit = true;
          return;
        }
        rstamp(23);
        // (every partial sum of a row half requested before the first is added: one trip to memory per half instead of S - the
        // shares were written by other XCDs, nothing of them is in this one's L2; profiles/r06u_stream_stamps.txt)
        auto sum_shares = [&](auto SC) __attribute__((always_inline)) {
          constexpr int SU = decltype(SC)::value;  // shares read per batch (0: any S, one at a time)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            float4 tot[4];
#pragma unroll
            for (int f4 = 0; f4 < 4; ++f4) tot[f4] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (SU > 0) {
              float4 v[SU][4];
#pragma unroll
              for (int sp = 0; sp < SU; ++sp)
#pragma unroll
                for (int f4 = 0; f4 < 4; ++f4) v[sp][f4] = get(part + ((size_t)sp * 8 + (mi * 4 + f4)) * NT + tid);
#pragma unroll
              for (int sp = 0; sp < SU; ++sp)
#pragma unroll
                for (int f4 = 0; f4 < 4; ++f4) {
                  tot[f4].x += v[sp][f4].x; tot[f4].y += v[sp][f4].y; tot[f4].z += v[sp][f4].z; tot[f4].w += v[sp][f4].w;
                }
            } else {
              for (int sp = 0; sp < nsplit; ++sp)
#pragma unroll
                for (int f4 = 0; f4 < 4; ++f4) {
                  const float4 v = get(part + ((size_t)sp * 8 + (mi * 4 + f4)) * NT + tid);
                  tot[f4].x += v.x; tot[f4].y += v.y; tot[f4].z += v.z; tot[f4].w += v.w;
                }
            }
#pragma unroll
            for (int f4 = 0; f4 < 4; ++f4) {
              const float4 b4 = *(const float4*)(pb + b2o + 64 * f4 + ncol);
              xr[mi][f4] = xpark[(mi * 4 + f4) * NT];
              xr[mi][f4].x += scale * (tot[f4].x + b4.x);
              xr[mi][f4].y += scale * (tot[f4].y + b4.y);
              xr[mi][f4].z += scale * (tot[f4].z + b4.z);
              xr[mi][f4].w += scale * (tot[f4].w + b4.w);
              if (repark) xpark[(mi * 4 + f4) * NT] = xr[mi][f4];
            }
          }
        };
        if (nsplit == 2) sum_shares(std::integral_constant<int, 2>{});
        else if (nsplit == 4) sum_shares(std::integral_constant<int, 4>{});
        else sum_shares(std::integral_constant<int, 0>{});
        rstamp(24);
      }
    }
  };
  // x += (W . act + bias): four K units (N = 256), requested by k_pre(w, 4)
  // `next(j)`, j = 0 .. 3: the request that fills ring[j] for the stage BEHIND this one.  Round 5: the stream does not stop
  // at a stage boundary - a ring slot is handed to the next stage the moment its unit has been consumed (slot j once unit
  // j is through, i.e. in front of unit j + 1; slot 3 behind the last unit), so the next stage's first units travel under
  // this stage's last MFMAs and the LayerNorm between the two instead of being requested - 24 to 32 wave-wide loads, 1.5 -
  // 2 K cycles of issue at a CU's 64 B/clk - at the head of that LayerNorm (profiles/r05b_block_stamps_fine.txt).
  auto no_next = [](int) {};
  auto proj_resid = [&](const float* pb, int bo, const void* w, auto&& next) __attribute__((always_inline)) {
    stream_k(w, std::integral_constant<int, 4>{}, [&](const WF& cur, int f) {
      if (f > 0) next(f - 1);
      f32x4 c[2];
      mma_k(cur, false, c);
      const float4 b4 = *(const float4*)(pb + bo + 64 * f + ncol);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        xr[mi][f].x += c[mi][0] + b4.x;
        xr[mi][f].y += c[mi][1] + b4.y;
        xr[mi][f].z += c[mi][2] + b4.z;
        xr[mi][f].w += c[mi][3] + b4.w;
      }
    });
    next(3);
  };

  // every parameter group of the launch goes to LDS once (C: 1 group, A: 2, D|FINAL: 3, D|A: 4; 7 KiB each), and with a
  // D part the depthwise conv's weights [31][256] f32 + bias [256].  With a D part the conv's own inputs go FIRST (its
  // weights here, the tile below) and everything the conv does not need is requested behind the tile barrier, to
  // travel while the conv computes: per-wave stamps (profiles/r03a_block_stamps_fine.txt) showed the workgroup
  // sitting 7 500 cycles in the ISSUE of one 240 KiB burst (every CU of the chip asks at once: ~33 B/clk per CU)
  // before the conv could start on the 64 KiB it actually needs.
  constexpr int NG = HAS_D ? (HAS_A ? 4 : 3) : (HAS_C ? 1 : 2);
  if (FOLD) {
    dma_lines(a.params_c, CPAR_OFF, std::integral_constant<int, PAR_BYTES / 1024>{});  // the C part's group goes first
  } else if (HAS_D) {
    dma_lines(a.dw_w, WK_OFF, std::integral_constant<int, KW>{});
    dma_lines(a.dw_b, WK_OFF + KW * 1024, std::integral_constant<int, 1>{});
  } else {
    dma_lines(a.params, PAR_OFF, std::integral_constant<int, NG * (PAR_BYTES / 1024)>{});
  }
  const float* const pb0 = par;
  const float* const pb1 = par + PAR_FLOATS;
  const float* const pb2 = par + 2 * PAR_FLOATS;
  const float* const pb3 = par + 3 * PAR_FLOATS;

  if (HAS_C && !HAS_D) {
    if constexpr (ATT && RELU) {
      // ---- Round 6, streaming layers: the plain multi-head attention over a block's slots (contextual_block_encoder_layer.py:
      // 236-262; attention.py:100-151) IN this launch - what csrc/streaming.hip `cb_mha_heads_mfma_kernel` did in a launch of
      // its own between block<A> and block<C>: 5 us of which 4 were a launch's cold start, with the context making a round
      // trip through memory.  A wave per head, both 16-query fragments of the workgroup in that wave; every operand of the
      // (<= 64 keys) tile requested up front together with linear_out's first units and the residual rows: one round trip.
      // Same arithmetic, same rounding points as that kernel (scores transposed S^T[key][query], scaled q in bf16, softmax
      // statistics per lane, bf16 probabilities as the B operand of O^T = V^T . P^T, the row sum from a ones fragment, the
      // context rounded to bf16): bit for bit the two-launch form (tests/test_gpu_streaming.py).
      // q / k row-major [B][H][Tpad][64], V^T [B][H][64][Tpad] (kv_frag = 0), T <= 64; att_mask: the contextual mask - the
      // last slot is no key, slot 0 (the context vector) attends to nothing and its context is zero.
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      const int hh = wave;
      const int Tp = a.Tpad;
      const size_t bh = (size_t)b * 4 + hh;
      const bf16* const qb = (const bf16*)a.qh + bh * Tp * 64;
      const bf16* const kb = (const bf16*)a.kh + bh * Tp * 64;
      const bf16* const vb = (const bf16*)a.vt + bh * 64 * Tp;
      const int nkeys = a.att_mask ? T - 1 : T;
      bf16x8 qraw[2][2], kf[4][2];
      uint2 vraw[4][2][2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qraw[mi][ks] = *(const bf16x8*)(qb + (size_t)(t0 + 16 * mi + lr) * 64 + ks * 32 + lg * 8);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kf[n][ks] = *(const bf16x8*)(kb + (size_t)(16 * n + lr) * 64 + ks * 32 + lg * 8);
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
            vraw[f][jp][hf] = *(const uint2*)(vb + (size_t)(16 * f + lr) * Tp + 32 * jp + 16 * hf + 4 * lg);
      read_unit(a.wout, 0, ring[0]);
      load_x();
      read_unit(a.wout, 1, ring[1]);
      read_unit(a.wout, 2, ring[2]);
      constexpr float LOG2E = 1.4426950408889634f;
      const bf16x8 ones = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        bf16x8 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) qf[ks][e] = (bf16)((float)qraw[mi][ks][e] * 0.125f);
        f32x4 sc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) sc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int n = 0; n < 4; ++n) sc[n] = MM::mma(kf[n][ks], qf[ks], sc[n]);
        float tm = -INFINITY;  // this lane: keys 16 n + 4 lg + r of query lr
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sc[n][r] = (16 * n + 4 * lg + r) < nkeys ? sc[n][r] : -INFINITY;
            tm = fmaxf(tm, sc[n][r]);
          }
        tm = wave_xor16_max(tm);
        tm = wave_xor32_max(tm);
        const float mnl = tm * LOG2E;
        unsigned pbu[2][4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[n][2 * h2], LOG2E, -mnl));
            const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[n][2 * h2 + 1], LOG2E, -mnl));
            pbu[n >> 1][(n & 1) * 2 + h2] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){e0, e1}, bf16x2));
          }
        f32x4 acc_o[4], acc_l = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < 4; ++f) acc_o[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const bf16x8 pbf = __builtin_bit_cast(bf16x8, (u32x4){pbu[jp][0], pbu[jp][1], pbu[jp][2], pbu[jp][3]});
          acc_l = MM::mma(ones, pbf, acc_l);
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            unsigned w[4];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
              const int nv = nkeys - (32 * jp + 16 * hf + 4 * lg);  // how many of the piece's four keys exist
              w[2 * hf] = vraw[f][jp][hf].x & (nv >= 2 ? 0xffffffffu : (nv == 1 ? 0xffffu : 0u));
              w[2 * hf + 1] = vraw[f][jp][hf].y & (nv >= 4 ? 0xffffffffu : (nv == 3 ? 0xffffu : 0u));
            }
            acc_o[f] = MM::mma(__builtin_bit_cast(bf16x8, (u32x4){w[0], w[1], w[2], w[3]}), pbf, acc_o[f]);
          }
        }
        // ctx[frame 16 mi + lr][64 hh + 16 f + 4 lg + r] -> bf16 -> k-tile hh of the LDS tile linear_out's activation fragments
        // are read from
        const bool none = a.att_mask && t0 + 16 * mi + lr == 0;
        const float inv = 1.0f / acc_l[0];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const bf16x4 pk = {(bf16)(none ? 0.f : acc_o[f][0] * inv), (bf16)(none ? 0.f : acc_o[f][1] * inv),
                             (bf16)(none ? 0.f : acc_o[f][2] * inv), (bf16)(none ? 0.f : acc_o[f][3] * inv)};
          *(bf16x4*)(abuf + hh * 4096 + mi * 2048 + lr * 128 + (((2 * f + (lg >> 1)) ^ swz) << 4) + (lg & 1) * 8) = pk;
        }
      }
      // this wave's parameter lines were requested before everything above
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      touch();
      bar(0);  // the context tile is complete; the parameter groups are in LDS
      load_act();
      stamp(4);
    } else if constexpr (ATT) {
      // ---- Round 6: the relative-position self-attention of the workgroup's 32 queries, IN this launch
      // (RelPositionMultiHeadedAttention.forward, attention.py:416-459; rel_shift :391-408; forward_attention :121-151):
      //     AC[i][j] = (q_i + u) . k_j     BD[i][j] = (q_i + v) . p[T-1-i+j]     ctx_i = softmax_j((AC + BD) / 8) . v_j
      // A wave per head, both 16-query fragments of the workgroup in that wave.  NO staging through LDS and no barrier: the
      // K rows, the position rows and the V^T rows of a 64-key tile are MFMA A operands exactly as they lie in memory (a lane
      // reads 16 bytes of "its" row), requested one tile ahead of the tile being computed, and every operand fragment feeds the
      // MFMAs of BOTH query fragments.  As in csrc/attention2.hip the scores are computed transposed, S^T[key][query], so the
      // softmax statistics are per-lane scalars and the bf16 probabilities are already the B operand of O^T = V^T . P^T; the
      // key rows of a K fragment are chosen (row lr of fragment n = key 32 (n >> 1) + 8 (lr >> 2) + 4 (n & 1) + (lr & 3)) so that
      // the eight contraction slots of a lane in P . V are eight CONSECUTIVE keys: V^T fragments are one 16-byte load per lane.
      // rel_shift: the dense window D[c][i] = p[c] . (q_i + v) over the 96 position rows the (32 query, 64 key) rectangle
      // can reach is computed per query fragment (80 rows each), written to a per-wave scratch and read back skewed.
      // Why this unit of work: relpos_attn2_kernel (utterance, head, 128 queries) and block<C> were two launches
      // of 11.3 + 12.3 us at B = 32, each a third cold prologue, with the context making a round trip through HBM between them
      // and five barriers per key tile in the first; here the context goes from the accumulators to the LDS tile linear_out
      // reads, the C part's weights are warmed into L2 while the attention runs, and the launch + prologue are paid once.
      typedef __attribute__((ext_vector_type(4))) float f4;
      typedef const __attribute__((address_space(1))) f4* GF4;
      constexpr int LDB = 84;
      const int hh = wave;  // head
      int klen_raw;
      asm volatile("s_load_dword %0, %1, 0x0" : "=s"(klen_raw) : "s"(a.klens + b) : "memory");  // (waited for below)
      // The residual rows go to their LDS parking place (lane-major, where an FFN parks them) by LDS-DMA - requested from
      // inside the LAST key tile, behind its score MFMAs: cold HBM lines (8 MB over the chip) that are needed only behind the
      // attention.  Requested at kernel entry (second version) they were part of the burst every workgroup opens with - 15 MB
      // that have to cross the fabric before the first MFMA: the prologue took 7.5 K cycles (profiles/r06c_block_stamps.txt).
      auto park_x = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            const float* xp = a.x + mrow[mi] * D + 64 * f + ncol;
            const unsigned dst = __builtin_amdgcn_readfirstlane(TILE_OFF + ((mi * 4 + f) * NT + wave * 64) * 16);
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %2\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, off\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(xp), "s"(dst)
                : "memory");
          }
      };
      const int Tp = a.Tpad;
      const size_t bh = (size_t)b * 4 + hh;
      GU8 qb = (GU8)a.qh + (bh * Tp + t0) * 128;
      // K, V^T and the position rows arrive FRAGMENT-MAJOR (EmBlockArgs.kv_frag: the A part of the launch in front wrote K
      // and V^T that way; em_relpos_pack_pos_bf16 the position windows): every operand fragment is 1 KiB contiguous, lane l
      // reads bytes [16 l, 16 l + 16).  A first version read them "as they lie" row-major - lane (lr, lg) 16 bytes of row lr:
      // 16 rows per wave-wide load, no two neighbouring lanes in the same line - and ran at the ~13 B/clk per CU of
      // uncoalesced requests: 27.6 us per launch against 23.6 for the two launches it replaces (profiles/r06a_*).
      // (wave-uniform bases in scalar registers + ONE 32-bit lane offset)
      const unsigned lo16 = lane * 16;
      GU8 kb = (GU8)em_uniform_ptr((const unsigned char*)a.kh + bh * Tp * 128);
      GU8 vb = (GU8)em_uniform_ptr((const unsigned char*)a.vt + bh * 64 * (size_t)Tp * 2);
      const int npg = a.ldp;  // position fragments per head
      const int g_q = 2 * (int)gridDim.x - (t0 >> 4);  // first position fragment of this query block at key tile 0
      GU8 ppb = (GU8)em_uniform_ptr((const unsigned char*)a.pos + ((size_t)hh * npg + g_q) * 2048);
      struct KP {
        bf16x8 k[8], p[12];  // k[2 n + ks], p[2 pn + ks]
      };
      struct VV {
        bf16x8 v[8];         // v[2 f + jp]
      };
      // ONE set of operand registers (K + position rows 80, V^T 32), refilled the moment its MFMAs are issued: K and the
      // position rows of tile t + 1 are requested behind the score MFMAs of tile t and travel under its rel-shift, softmax
      // and P . V; V^T of tile t + 1 is requested behind P . V of tile t and travels under the scores and softmax of t + 1.
      // (Second version.  The first kept two whole tiles in flight - 224 operand registers next to the ~190 the two
      // query fragments' softmax needs: at the 256-VGPR line hipcc turned every skewed scratch read of the rel-shift
      // into "read, wait, add" through the same two registers, 16 dependent LDS round trips per fragment and tile:
      // 4.1 K cycles per tile for 0.9 K of MFMA, profiles/r06b_block_stamps.txt.  Requests written as inline asm with AGPR
      // destinations and hand-counted waits tied to their registers cost ~200 accumulator-register copies per tile.)
      // Every request is unconditional (a tile past the end repeats the last one), so that hipcc counts its waits exactly.
      auto load_kp = [&](int jt, KP& kp) __attribute__((always_inline)) {
        GU8 rk = kb + (size_t)jt * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i) kp.k[i] = *(GFRAG)(rk + i * 1024 + lo16);
        GU8 rp = ppb + (size_t)jt * 8192;
#pragma unroll
        for (int i = 0; i < 12; ++i) kp.p[i] = *(GFRAG)(rp + i * 1024 + lo16);
      };
      auto load_v = [&](int jt, VV& vv) __attribute__((always_inline)) {
        GU8 rv = vb + (size_t)jt * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i) vv.v[i] = *(GFRAG)(rv + i * 1024 + lo16);
      };
      // the queries (B operand: column = query 16 mi + lr, k-slice lg) and the two position biases
      bf16x8 qraw[2][2];
      f4 ur[2][2], vr[2][2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qraw[mi][ks] = *(GFRAG)(qb + (16 * mi + lr) * 128 + (ks * 4 + lg) * 16);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          ur[ks][h2] = *(GF4)(a.pos_u + hh * 64 + ks * 32 + lg * 8 + h2 * 4);
          vr[ks][h2] = *(GF4)(a.pos_v + hh * 64 + ks * 32 + lg * 8 + h2 * 4);
        }
      KP kp;
      VV vv;
      load_kp(0, kp);
      load_v(0, vv);
      touch();  // the C part's weights into this XCD's L2 while the attention runs
      int klen;
      {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(klen_raw) : : "memory");
        klen = klen_raw < T ? klen_raw : T;
        klen = klen > 0 ? klen : 1;
      }
      const int ntile = (klen + 63) >> 6, nfull = klen >> 6;
      bf16x8 qu[2][2], qv[2][2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float qf = (float)qraw[mi][ks][e];
            // 1 / sqrt(64) goes into the operands (a power of two: (AC + BD) / 8 comes out bit for bit the same)
            qu[mi][ks][e] = (bf16)((qf + ur[ks][e >> 2][e & 3]) * 0.125f);
            qv[mi][ks][e] = (bf16)((qf + vr[ks][e >> 2][e & 3]) * 0.125f);
          }
      f32x4 acc_o[2][4], acc_l[2];
      float row_m[2] = {-INFINITY, -INFINITY};
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        acc_l[mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < 4; ++f) acc_o[mi][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      // (the softmax denominator rides on the matrix core: a fifth "V^T fragment" of ones, as in attention2.hip)
      const bf16x8 ones = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
      constexpr float LOG2E = 1.4426950408889634f;
      float* const bdw = (float*)(smem + XCH_OFF) + wave * (2 * 16 * LDB);
      // one 64-key tile; MASK: the tile holds the utterance's end (or lies past it: odd tile counts) - two instantiations,
      // chosen by a uniform branch AROUND the tile, so that each is one scheduling region
      const int last = ((klen + 63) >> 6) - 1;
      auto tile = [&](int jt, auto maskc) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(maskc)::value;
        const int j0 = jt * 64;
        const int jn = jt < last ? jt + 1 : last;  // the tile whose operands are requested from inside this one
        f32x4 sc[2][4], dd[2][5];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
          for (int n = 0; n < 4; ++n) sc[mi][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int n = 0; n < 5; ++n) dd[mi][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          // window row cw = 31 - 16 mi - li + jl: queries 0 .. 15 reach position fragments 1 .. 5, queries 16 .. 31 fragments 0 .. 4
#pragma unroll
          for (int pn = 0; pn < 6; ++pn) {
            if (pn >= 1) dd[0][pn - 1] = MM::mma(kp.p[2 * pn + ks], qv[0][ks], dd[0][pn - 1]);
            if (pn <= 4) dd[1][pn] = MM::mma(kp.p[2 * pn + ks], qv[1][ks], dd[1][pn]);
          }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) sc[mi][n] = MM::mma(kp.k[2 * n + ks], qu[mi][ks], sc[mi][n]);
        if constexpr (!(EM_BLOCK_VAR & 512)) load_kp(jn, kp);  // (the registers are free: their MFMAs have been issued)
        if (jt == last) park_x();
        // rel_shift: D[c][i] -> scratch[i][c], read back at c = 15 - i + (key of the tile)
        if constexpr (!(EM_BLOCK_VAR & 256)) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int n = 0; n < 5; ++n)
            *(float4*)(bdw + (mi * 16 + lr) * LDB + 16 * n + 4 * lg) = make_float4(dd[mi][n][0], dd[mi][n][1], dd[mi][n][2], dd[mi][n][3]);
        }
        bf16x8 pb[2][2];
        float alpha[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const float* bdr = bdw + (mi * 16 + lr) * LDB + 15 - lr + 8 * lg;
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[mi][n][r] += (EM_BLOCK_VAR & 256) ? dd[mi][n][r] : bdr[32 * (n >> 1) + 4 * (n & 1) + r];
          if constexpr (MASK) {
            const int kl = klen - j0 - 8 * lg;  // this lane's keys of the tile are 32 (n >> 1) + 4 (n & 1) + r < kl
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
              for (int r = 0; r < 4; ++r) sc[mi][n][r] = (32 * (n >> 1) + 4 * (n & 1) + r < kl) ? sc[mi][n][r] : -INFINITY;
          }
          float tm = -INFINITY;
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) tm = fmaxf(tm, sc[mi][n][r]);
          tm = wave_xor16_max(tm);  // the four lane groups of a query
          tm = wave_xor32_max(tm);
          const float mn = fmaxf(row_m[mi], tm);
          alpha[mi] = __expf(row_m[mi] - mn);
          row_m[mi] = mn;
          const float mnl = mn * LOG2E;
          unsigned pbu[2][4];
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              // (developer timing builds, wrong results: EM_BLOCK_VAR & 128 no exponentials, & 256 no rel-shift scratch round
              // trip, & 512 the operand requests of tile 0 only)
              const float a0 = __builtin_fmaf(sc[mi][n][2 * h], LOG2E, -mnl), a1 = __builtin_fmaf(sc[mi][n][2 * h + 1], LOG2E, -mnl);
              const float e0 = (EM_BLOCK_VAR & 128) ? a0 : __builtin_amdgcn_exp2f(a0);
              const float e1 = (EM_BLOCK_VAR & 128) ? a1 : __builtin_amdgcn_exp2f(a1);
              pbu[n >> 1][(n & 1) * 2 + h] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){e0, e1}, bf16x2));
            }
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            pb[mi][jp] = __builtin_bit_cast(bf16x8, (u32x4){pbu[jp][0], pbu[jp][1], pbu[jp][2], pbu[jp][3]});
          }
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            acc_o[mi][f][0] *= alpha[mi]; acc_o[mi][f][1] *= alpha[mi]; acc_o[mi][f][2] *= alpha[mi]; acc_o[mi][f][3] *= alpha[mi];
          }
          acc_l[mi][0] *= alpha[mi];  // (the other three rows of the ones fragment carry the same sum; only this one is read)
        }
        // O^T += V^T . P^T
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            acc_l[mi] = MM::mma(ones, pb[mi][jp], acc_l[mi]);
#pragma unroll
            for (int f = 0; f < 4; ++f) acc_o[mi][f] = MM::mma(vv.v[2 * f + jp], pb[mi][jp], acc_o[mi][f]);
          }
        if constexpr (!(EM_BLOCK_VAR & 512)) load_v(jn, vv);
      };
      stamp(60);
#pragma unroll 1
      for (int jt = 0; jt < ntile; ++jt) {
        if (jt < nfull) tile(jt, std::false_type{});
        else tile(jt, std::true_type{});
      }
      stamp(61);
      // linear_out's first units travel while the context is normalised and handed over; the residual rows come out of
      // their parking place
      read_unit(a.wout, 0, ring[0]);
      read_unit(a.wout, 1, ring[1]);
      read_unit(a.wout, 2, ring[2]);
      // the residual rows' DMA (requested inside the last tile, not known to hipcc's counting) has landed once at most the 24
      // requests just made are outstanding
      asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      {
        const float4* const xp = (const float4*)(smem + TILE_OFF) + tid;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int f = 0; f < 4; ++f) xr[mi][f] = xp[(mi * 4 + f) * NT];
      }
      // ctx[frame 16 mi + lr][64 hh + 16 f + 4 lg + r] -> bf16 -> k-tile hh of the LDS tile linear_out's activation fragments
      // are read from (the same rounding point as the ctx buffer of the two-launch form)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const float row_l = acc_l[mi][0];
        const float inv = row_l > 0.f ? 1.0f / row_l : 0.f;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const bf16x4 pk = {(bf16)(acc_o[mi][f][0] * inv), (bf16)(acc_o[mi][f][1] * inv), (bf16)(acc_o[mi][f][2] * inv),
                             (bf16)(acc_o[mi][f][3] * inv)};
          *(bf16x4*)(abuf + hh * 4096 + mi * 2048 + lr * 128 + (((2 * f + (lg >> 1)) ^ swz) << 4) + (lg & 1) * 8) = pk;
        }
      }
      bar(0);  // the context tile is complete; the parameter lines (requested before everything else) are in LDS
      load_act();
      stamp(4);
    } else {
    // linear_out over the attention context: activation fragments straight from global memory
    // (attention.py:151 linear_out; encoder_layer.py:142-147 residual)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const bf16* crow = (const bf16*)a.ctx + mrow[mi] * D + lg * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) act[mi][ks] = *(const bf16x8*)(crow + ks * 32);
    }
    read_unit(a.wout, 0, ring[0]);
    __builtin_amdgcn_sched_barrier(0);  // what the first MFMAs wait for goes first (left alone hipcc requests unit 0 last)
    load_x();
    read_unit(a.wout, 1, ring[1]);
    read_unit(a.wout, 2, ring[2]);
    // this wave's parameter lines were requested before the 48 loads above: in LDS once at most 48 are outstanding
    fstamp(2);
    asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    fstamp(3);
    touch();
    bar(0);  // the parameter groups are in LDS
    fstamp(4);
    }
    // G0: [bout 256][norm_conv g 256][b 256][pw1 bias, fused order 512]
    proj_resid(pb0, 0, a.wout, [&](int j) {
      if (j < 3) read_unit(a.pw1f, j, ring[j]);
    });
    fstamp(50);
    // (x is stored at the end of the kernel, behind the last weight request: see the A part)
    ln_to_act(pb0, 256, 512, 0);
    fstamp(15);
    // pointwise_conv1 + GLU (convolution.py:66-69): unit 2j = value rows 64j.., unit 2j+1 = their gates
    f32x4 v[2], gt[2];
    stream_k(a.pw1f, std::integral_constant<int, 8>{}, [&](const WF& cur, int u) {
      if (!(u & 1)) {
        mma_k(cur, false, v);
      } else {
        mma_k(cur, false, gt);
        const int j = u >> 1;
        const float4 bv = *(const float4*)(pb0 + 768 + (2 * j) * 64 + ncol);
        const float4 bg = *(const float4*)(pb0 + 768 + (2 * j + 1) * 64 + ncol);
        // (8-byte pieces, as they come: traded into 16-byte stores through v_permlane16_swap like q / k / v in the A part
        // this kernel got SLOWER, 12.46 -> 12.83 us - its tail is the 32 KiB of x, not the 16 KiB of glu; profiles/r03ac)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          bf16x4 pk = {(bf16)((v[mi][0] + bv.x) * sigmoidf_(gt[mi][0] + bg.x)),
                       (bf16)((v[mi][1] + bv.y) * sigmoidf_(gt[mi][1] + bg.y)),
                       (bf16)((v[mi][2] + bv.z) * sigmoidf_(gt[mi][2] + bg.z)),
                       (bf16)((v[mi][3] + bv.w) * sigmoidf_(gt[mi][3] + bg.w))};
          if (row_ok[mi]) *(bf16x4*)((bf16*)a.glu + mrow[mi] * D + 64 * j + ncol) = pk;
        }
      }
    });
    store_x();
    if constexpr (ATT) stamp(51);
    else fstamp(51);
    touch_done();
    return;
  }

  if (HAS_D) {
    // ---- depthwise conv (k = 31, zero padded) + folded BatchNorm + Swish (convolution.py:72-75):
    // the 62-row input tile of this block (frames t0 - 15 .. t0 + 46 of the utterance, zero outside
    // [0, Tv)) goes through LDS; thread c (= channel) produces the 32 frames of the block.
    unsigned char* const tile = smem + TILE_OFF;  // [62][256] bf16, TPITCH bytes per row
    int Tv = T;
    if constexpr (!FOLD) {
      uint4 stage[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {  // unconditional loads from a clamped row (a load under a lane mask is
        const int q = tid + it * NC, r = q >> 5, ch = q & 31;  // waited for on the spot), zeroed afterwards
        const int t = t0 - HALF + r;
        const int tc = t < 0 ? 0 : (t < T ? t : T - 1);
        stage[it] = *(const uint4*)((const bf16*)a.glu + ((size_t)b * T + tc) * D + ch * 8);
      }
      if (a.tlens) {  // a scalar load (uniform address), behind the tile requests
        int tl;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl) : "s"(a.tlens + b) : "memory");
        Tv = tl < T ? tl : T;
      }
      fstamp(2);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int q = tid + it * NC, t = t0 - HALF + (q >> 5);
        if (q < TROWS * 32)
          *(uint4*)(tile + (q >> 5) * TPITCH + (q & 31) * 16) = (t >= 0 && t < Tv) ? stage[it] : make_uint4(0u, 0u, 0u, 0u);
      }
      fstamp(3);
      bar(BAR_TILE);  // (the conv-weight lines were requested before the tile: a wave that has its tile rows has them too)
    } else {
      // ---- the C part, folded (round 4): linear_out + residual -> norm_conv -> pointwise_conv1 + GLU for the 64 frames
      // t0 - 16 .. t0 + 47 as four 16-frame fragments: 1 and 2 are this workgroup's own rows (mi = 0, 1 of everything
      // that follows), 0 and 3 the halo the depthwise conv needs (hi = 0, 1).  GLU rows go straight into the conv tile.
      const float* const cpar = (const float*)(smem + CPAR_OFF);  // [bout 256][norm_conv g 256][b 256][pw1 bias, fused order 512]
      unsigned char* const abuf64 = smem + ABUF64_OFF;
      int th[2];
      size_t hrow[2];
#pragma unroll
      for (int hi = 0; hi < 2; ++hi) {
        th[hi] = t0 - 16 + hi * 48 + lr;
        hrow[hi] = (size_t)b * T + (th[hi] < 0 ? 0 : (th[hi] < T ? th[hi] : T - 1));
      }
      bf16x8 acth[2][8];
      float4 xh[2][4];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const bf16* crow = (const bf16*)a.ctx + mrow[mi] * D + lg * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) act[mi][ks] = *(const bf16x8*)(crow + ks * 32);
      }
      read_unit(a.wout, 0, ring[0]);
      __builtin_amdgcn_sched_barrier(0);  // what the first MFMAs wait for goes first
      load_x();
#pragma unroll
      for (int hi = 0; hi < 2; ++hi) {
        const bf16* crow = (const bf16*)a.ctx + hrow[hi] * D + lg * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) acth[hi][ks] = *(const bf16x8*)(crow + ks * 32);
#pragma unroll
        for (int f = 0; f < 4; ++f) xh[hi][f] = *(const float4*)(a.x + hrow[hi] * D + 64 * f + ncol);
      }
      read_unit(a.wout, 1, ring[1]);
      read_unit(a.wout, 2, ring[2]);
      if (a.tlens) {
        int tl;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl) : "s"(a.tlens + b) : "memory");
        Tv = tl < T ? tl : T;
      }
      // this wave's two parameter lines were requested before the 80 loads above: landed once at most 63 are outstanding
      // (the counter's range; it asks for a few of the loads behind them as well)
      fstamp(2);
      asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
      fstamp(3);
      touch();
      bar(0);  // the C part's parameter group is in LDS
      fstamp(4);
      // linear_out + residual (attention.py:151; encoder_layer.py:142-147), own and halo rows from the same unit
      stream_k(a.wout, std::integral_constant<int, 4>{}, [&](const WF& cur, int f) {
        f32x4 c[2], ch[2];
        mma_of(act, cur, false, c);
        mma_of(acth, cur, false, ch);
        const float4 b4 = *(const float4*)(cpar + 64 * f + ncol);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          xr[mi][f].x += c[mi][0] + b4.x;
          xr[mi][f].y += c[mi][1] + b4.y;
          xr[mi][f].z += c[mi][2] + b4.z;
          xr[mi][f].w += c[mi][3] + b4.w;
          xh[mi][f].x += ch[mi][0] + b4.x;
          xh[mi][f].y += ch[mi][1] + b4.y;
          xh[mi][f].z += ch[mi][2] + b4.z;
          xh[mi][f].w += ch[mi][3] + b4.w;
        }
      });
      fstamp(50);
      // The conv's weights travel while the C part computes (nothing reads WK before the tile barrier).  Requested HERE:
      // no register load is outstanding at this point, so hipcc's wait counts for the units that follow stay exact (it
      // does not see these requests), and the 32 KiB land during the LayerNorm's two barriers.
      dma_lines(a.dw_w, WK_OFF, std::integral_constant<int, KW>{});
      dma_lines(a.dw_b, WK_OFF + KW * 1024, std::integral_constant<int, 1>{});
      k_pre(a.pw1f, 8);
      // norm_conv of the 64 rows (encoder_layer.py:149-152): both statistics through one barrier
      {
        float mean[2], rstd[2], meanh[2], rstdh[2];
        ln_partials(xr, red0);
        ln_partials(xh, red0 + 256);
        fstamp(11);
        bar(0);
        fstamp(12);
        ln_finish(red0, mean, rstd);
        ln_finish(red0 + 256, meanh, rstdh);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const float4 g4 = *(const float4*)(cpar + 256 + 64 * f + ncol);
          const float4 b4 = *(const float4*)(cpar + 512 + 64 * f + ncol);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const bf16x4 pc = {(bf16)((xr[mi][f].x - mean[mi]) * rstd[mi] * g4.x + b4.x),
                               (bf16)((xr[mi][f].y - mean[mi]) * rstd[mi] * g4.y + b4.y),
                               (bf16)((xr[mi][f].z - mean[mi]) * rstd[mi] * g4.z + b4.z),
                               (bf16)((xr[mi][f].w - mean[mi]) * rstd[mi] * g4.w + b4.w)};
            const bf16x4 ph = {(bf16)((xh[mi][f].x - meanh[mi]) * rstdh[mi] * g4.x + b4.x),
                               (bf16)((xh[mi][f].y - meanh[mi]) * rstdh[mi] * g4.y + b4.y),
                               (bf16)((xh[mi][f].z - meanh[mi]) * rstdh[mi] * g4.z + b4.z),
                               (bf16)((xh[mi][f].w - meanh[mi]) * rstdh[mi] * g4.w + b4.w)};
            *(bf16x4*)(abuf64 + f * 8192 + (1 + mi) * 2048 + tile_wr) = pc;   // fragments 1, 2
            *(bf16x4*)(abuf64 + f * 8192 + (3 * mi) * 2048 + tile_wr) = ph;   // fragments 0, 3
          }
        }
        fstamp(13);
        bar(0);
        fstamp(14);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) {
            const int o = (ks >> 1) * 8192 + lr * 128 + ((((ks & 1) * 4 + lg) ^ swz) << 4);
            act[mi][ks] = *(const bf16x8*)(abuf64 + o + (1 + mi) * 2048);
            acth[mi][ks] = *(const bf16x8*)(abuf64 + o + (3 * mi) * 2048);
          }
      }
      fstamp(15);
      // pointwise_conv1 + GLU (convolution.py:66-69): unit 2j = value rows 64j.., unit 2j+1 = their gates; the GLU row of
      // frame t goes to tile row t - (t0 - 15), zero outside [0, Tv) like the conv's zero padding
      int trow[4];
      bool tin[4], tval[4];
#pragma unroll
      for (int fi = 0; fi < 4; ++fi) {
        const int t = t0 - 16 + fi * 16 + lr;
        trow[fi] = fi * 16 + lr - 1;
        tin[fi] = trow[fi] >= 0 && trow[fi] < TROWS;
        tval[fi] = t >= 0 && t < Tv;
      }
      {
        f32x4 v[2], gt[2], vh[2], gth[2];
        stream_k(a.pw1f, std::integral_constant<int, 8>{}, [&](const WF& cur, int u) {
          if (!(u & 1)) {
            mma_of(act, cur, false, v);
            mma_of(acth, cur, false, vh);
          } else {
            mma_of(act, cur, false, gt);
            mma_of(acth, cur, false, gth);
            const int j = u >> 1;
            const float4 bv = *(const float4*)(cpar + 768 + (2 * j) * 64 + ncol);
            const float4 bg = *(const float4*)(cpar + 768 + (2 * j + 1) * 64 + ncol);
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
              const bool own = fi == 1 || fi == 2;
              const int i2 = own ? fi - 1 : (fi == 0 ? 0 : 1);
              const f32x4 vv = own ? v[i2] : vh[i2], gg = own ? gt[i2] : gth[i2];
              bf16x4 pk = {(bf16)((vv[0] + bv.x) * sigmoidf_(gg[0] + bg.x)), (bf16)((vv[1] + bv.y) * sigmoidf_(gg[1] + bg.y)),
                           (bf16)((vv[2] + bv.z) * sigmoidf_(gg[2] + bg.z)), (bf16)((vv[3] + bv.w) * sigmoidf_(gg[3] + bg.w))};
              if (!tval[fi]) pk = (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
              if (tin[fi]) *(bf16x4*)(tile + trow[fi] * TPITCH + (64 * j + ncol) * 2) = pk;
            }
          }
        });
      }
      fstamp(51);
      bar(BAR_TILE);  // the tile is complete (and the conv weights, requested before it by every wave, are in LDS)
      fstamp(52);
    }
    // Behind the tile barrier, IN the convolution: the parameter groups, the residual rows, pointwise_conv2's first units
    // and the L2 warm-up, in three pieces of <= 64 KiB dealt into the conv's row loop.  Requested in one go they
    // block the wave in their issue (the memory pipeline of a CU takes ~64 KiB before it makes the issuing wave
    // wait: ~5 000 cycles for the 190 KiB, profiles/r03b_block_stamps_fine.txt) and the conv starts that much later.
    // The conv itself: two adjacent channels x 16 frames per thread on packed f32 FMAs (v_pk_fma_f32): one 4-byte LDS
    // word carries both channels' bf16, weights [k][256] f32 tap-major in LDS (conflict-free 8-byte reads).
    {
      const int cp = tid & 127, hh = tid >> 7;
      f32x2 wk2[KW];
#pragma unroll
      for (int k = 0; k < KW; ++k) wk2[k] = *(const f32x2*)(smem + WK_OFF + (k * D + 2 * cp) * 4);
      const f32x2 bc2 = *(const f32x2*)(smem + WK_OFF + (KW * D + 2 * cp) * 4);
      f32x2 acc[16];
#pragma unroll
      for (int o = 0; o < 16; ++o) acc[o] = bc2;
      if constexpr (!(EM_BLOCK_VAR & 64)) {  // the requests in three clusters (-DEM_BLOCK_VAR=64: one per row, below - measured no faster)
        dma_lines(a.params, PAR_OFF, std::integral_constant<int, NG * (PAR_BYTES / 1024)>{});
        if constexpr (!FOLD) load_x();  // (folded: the own rows' residual is already in registers, updated by linear_out)
        __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
        for (int r = 0; r < 16 + KW - 1; ++r) {
          // (pinned: the opaque asm ties a zero offset of the request addresses to ALL accumulators of the running
          // convolution, so the requests can neither be hoisted above this row nor the rows before it sunk below them;
          // a sched_barrier alone does not do it: instruction selection places arithmetic freely around it)
          constexpr int PIN1 = KWT == 31 ? 15 : 10, PIN2 = KWT == 31 ? 30 : 20;  // (thirds of the row loop)
          if (r == PIN1 || r == PIN2) {
            unsigned o = 0;
            asm volatile("" : "+s"(o), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]),
                         "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]),
                         "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]));
            const unsigned char* wp = (const unsigned char*)a.pw2 + o;
            if (r == PIN1) {
              read_unit(wp, 0, ring[0]);
              read_unit(wp, 1, ring[1]);
            } else {
              read_unit(wp, 2, ring[2]);
            }
            __builtin_amdgcn_sched_barrier(0);  // ... nor sink below the rows that follow
          }
          const unsigned u = *(const unsigned*)(tile + (hh * 16 + r) * TPITCH + 4 * cp);
          const f32x2 v = {__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
  #pragma unroll
          for (int o = 0; o < 16; ++o)
            if (r - o >= 0 && r - o < KW) acc[o] = __builtin_elementwise_fma(wk2[r - o], v, acc[o]);
        }
      } else {
        // Round 5 experiment (built, verified, NOT faster: profiles/r05c_block_ab.txt - the conv phase drops from 9.3 K to
        // 8.4 K cycles, but pointwise_conv2's units, now requested as late as row 38, arrive late and its four units take
        // 5.3 K cycles instead of 1.9 K; the launch as a whole 1.026 - 1.029 against 1.021 - 1.025 ms per step):
        // ONE request per row of the convolution instead of three clusters.  A cluster of 15 - 16 wave-wide
        // requests (four waves: 60 KiB) stops the issuing wave for as long as the CU's memory pipeline takes to accept it
        // - ~1 K cycles at 64 B/clk - and with one wave per SIMD nothing else issues meanwhile: the conv measured 9.3 K
        // cycles against ~4.8 K of arithmetic + LDS waits (profiles/r05b_block_stamps_fine.txt: stamps 3 -> 5).  A row is
        // ~72 cycles of packed FMAs; four waves asking for 1 KiB each per row stay under the 64 B/clk the pipeline drains.
        // The order is: this wave's parameter lines (LDS-DMA), the residual rows, pointwise_conv2's first three units -
        // the waits below count on the parameter lines being the oldest.  Every request sits between two pins: an opaque
        // asm that redefines a zero offset in a VGPR (`vz`, part of every request's address) together with ALL
        // accumulators - a request can neither be hoisted above the pin in front of it (its address depends on it) nor
        // sunk below the one behind it (which overwrites the register it reads), and the pins are ordered with the rows
        // through the accumulators.  No sched_barrier: the tile reads and the arithmetic of neighbouring rows interleave
        // freely.
        constexpr int NLINES = NG * (PAR_BYTES / 1024), NPW = (NLINES + 3) / 4;  // parameter lines of this wave
        constexpr int NXL = FOLD ? 0 : 8, NREQ = NPW + NXL + 24, NROW = 16 + KW - 1, RPR = (NREQ + NROW - 1) / NROW;
        unsigned vz = 0;
        auto pin = [&]() __attribute__((always_inline)) {
          asm volatile("" : "+v"(vz), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]),
                       "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]),
                       "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]));
        };
        auto request = [&](auto sc) __attribute__((always_inline)) {
          constexpr int S = decltype(sc)::value;
          if constexpr (S < NPW) {  // parameter line wave + 4 S (past the end: the last line again)
            int j = wave + 4 * S;
            j = j < NLINES ? j : NLINES - 1;
            const unsigned char* pl = em_uniform_ptr((const unsigned char*)a.params + (size_t)j * 1024);
            const unsigned dst = __builtin_amdgcn_readfirstlane(PAR_OFF + j * 1024);
            const unsigned vo = lane * 16 + vz;
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %3\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, %2\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(vo), "s"(pl), "s"(dst)
                : "memory");
          } else if constexpr (S < NPW + NXL) {
            constexpr int i = S - NPW, mi = i >> 2, f = i & 3;
            xr[mi][f] = *(const float4*)((const unsigned char*)(a.x + mrow[mi] * D + 64 * f + ncol) + vz);
          } else {
            constexpr int i = S - NPW - NXL, u = i >> 3, q = i & 7;
            GU8 su = (GU8)a.pw2 + (size_t)u * UNIT + (voff + vz);
            ring[u].v[q] = *(GFRAG)(su + q * 1024);
          }
        };
        auto requests_of_row = [&](auto rc) __attribute__((always_inline)) {
          constexpr int R = decltype(rc)::value;
          if constexpr (R * RPR < NREQ) {
            pin();
            request(std::integral_constant<int, R * RPR>{});
            if constexpr (RPR > 1 && R * RPR + 1 < NREQ) request(std::integral_constant<int, R * RPR + 1>{});
          }
        };
        // (the tile words are read PF rows ahead, by hand: memory operations do not cross the pins - an asm with side
        // effects orders them - so left to hipcc every row's LDS read sat right in front of its first use)
        constexpr int PF = 3;
        unsigned uw[NROW + PF];
        auto tile_word = [&](int r) __attribute__((always_inline)) { return *(const unsigned*)(tile + (hh * 16 + r) * TPITCH + 4 * cp); };
#pragma unroll
        for (int r = 0; r < PF; ++r) uw[r] = tile_word(r);
        auto rows = [&](auto self, auto rc) __attribute__((always_inline)) -> void {
          constexpr int r = decltype(rc)::value;
          if constexpr (r < NROW) {
            requests_of_row(rc);
            if constexpr (r + PF < NROW) uw[r + PF] = tile_word(r + PF);
            const unsigned u = uw[r];
            const f32x2 v = {__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
#pragma unroll
            for (int o = 0; o < 16; ++o)
              if (r - o >= 0 && r - o < KW) acc[o] = __builtin_elementwise_fma(wk2[r - o], v, acc[o]);
            self(self, std::integral_constant<int, r + 1>{});
          }
        };
        rows(rows, std::integral_constant<int, 0>{});
        pin();  // (closes the last interval)
      }
      if constexpr (!FOLD) touch();  // (its chunk loop must stay OUT of the row loop: with a loop inside, hipcc does not unroll the rows and the accumulators go to scratch; folded: done in the C part)
      const int ch = 2 * cp, kt = ch >> 6, kl = ch & 63;
#pragma unroll
      for (int o = 0; o < 16; ++o) {
        const int m = hh * 16 + o;
        bf16x2 pk = {(bf16)swishf_(acc[o][0]), (bf16)swishf_(acc[o][1])};
        *(bf16x2*)(abuf + (kt * 32 + m) * 128 + (((kl >> 3) ^ (m & 7)) << 4) + (kl & 7) * 2) = pk;
      }
    }
    fstamp(5);
    fstamp(6);
    // this wave's parameter lines were requested before the 32 loads of x and pointwise_conv2: landed once at most 32 are
    // outstanding (the barrier then publishes them with the conv output)
    if constexpr (FOLD) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // (no x loads behind the parameter lines)
    else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    bar(0);  // conv output + parameter groups visible
    load_act();
    stamp(7);
    // G0: [pw2 bias 256][norm_ff g 256][b 256]; G1: [ff b1 1024][ff b2 256][norm_final g 256][b 256]
    // pointwise_conv2 + residual (convolution.py:77, encoder_layer.py:149-158); the ring goes over to the feed-forward module
    proj_resid(pb0, 0, a.pw2, [&](int j) {
      if (j == 0) read_unit(ff_w1, 0, ring[0]);
      else if (j == 1) read_unit(ff_w1, 1, ring[1]);
      else if (j == 2) read_w2(ff_w2, 0, 0, ring[2]);
      else read_w2(ff_w2, 0, 1, ring[3]);
    });
    stamp(10);
    ln_to_act(pb0, 256, 512, 0);                   // norm_ff
    stamp(15);
    // x += 0.5 * FFN(norm_ff(x))   (encoder_layer.py:160-168); with an A part behind it the ring goes over to the macaron module
    if constexpr (HAS_A)
      ffn(pb1, 0, 1024, 0.5f, ff_w1, ff_w2, false, ff_b1g, ffm_w1, [&] {
        read_w2(ffm_w2, 0, 0, ring[2]);
        read_w2(ffm_w2, 0, 1, ring[3]);
      });
    else
      ffn(pb1, 0, 1024, 0.5f, ff_w1, ff_w2, false, ff_b1g, nullptr, [] {});
    if constexpr (RELU) {
      if (ffn_exit) return;  // (split FFN: another share of this row block carries the rows on)
    }
    stamp(22);
    {
      float4 y[2][4];
      ln_apply(pb1, 1280, 1536, y, 0);             // norm_final (encoder_layer.py:170-171): the block's output
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int f = 0; f < 4; ++f) xr[mi][f] = y[mi][f];
    }
    if (FINAL) {
      // after_norm (conformer_encoder.py:423-424); G2: [after_norm g 256][b 256]
      float4 y[2][4];
      ln_apply(pb2, 0, 256, y, CTC ? 0 : BAR_LAST);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        if (row_ok[mi]) {
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            *(float4*)(a.enc_out + mrow[mi] * D + 64 * f + ncol) = y[mi][f];
            bf16x4 pk = {(bf16)y[mi][f].x, (bf16)y[mi][f].y, (bf16)y[mi][f].z, (bf16)y[mi][f].w};
            *(bf16x4*)((bf16*)a.enc_act + mrow[mi] * D + 64 * f + ncol) = pk;
          }
        }
      if (!CTC) {
        touch_done();
        return;
      }
      // ---- CTC head (asr/ctc.py:207-215 argmax over ctc_lo): the 32 rows stay in registers as activation
      // fragments, the [V][256] weight streams through the ring 64 labels at a time, every lane keeps the running
      // (max, label) of its frames.  Labels ascend with the unit and inside a lane, so a strict > keeps the lowest
      // label on ties, as torch.argmax does; the cross-lane / cross-wave merges break ties the same way.
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          bf16x4 pk = {(bf16)y[mi][f].x, (bf16)y[mi][f].y, (bf16)y[mi][f].z, (bf16)y[mi][f].w};
          *(bf16x4*)(abuf + f * 4096 + mi * 2048 + tile_wr) = pk;
        }
      const int NU = a.ctc_units;
      k_pre(a.ctc_w, NU);
      bar(0);
      load_act();
      float best[2] = {-INFINITY, -INFINITY};
      int bidx[2] = {0, 0};
      // the bias of unit u + 1 is requested while unit u computes: a global load issued next to its use would
      // put its L2 latency into every one of the 79 intervals
      float4 bnext = *(const float4*)(a.ctc_b + ncol);
      // Software-pipelined by one unit (round 6): the 16 MFMAs of unit u are issued FIRST, the arg-max update of unit u - 1
      // (bias, compare, select: ~40 VALU instructions that need nothing of unit u) runs while they execute.  As one dependent
      // body per unit - MFMAs, wait for them, update - the walk took ~760 cycles per unit for 256 of MFMA.
      auto ctc_update = [&](const f32x4 (&c)[2], const float4 bb, int u) __attribute__((always_inline)) {
        const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = c[mi][r] + bv[r];
            if (v > best[mi]) {
              best[mi] = v;
              bidx[mi] = u * 64 + ncol + r;
            }
          }
      };
      f32x4 cprev[2] = {(f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY}, (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY}};
      float4 bprev = make_float4(0.f, 0.f, 0.f, 0.f);
      int uprev = 0;  // (the first update sees -inf everywhere: nothing is taken)
#pragma unroll 1
      for (int u0 = 0; u0 < NU; u0 += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int u = u0 + j;
          read_unit(a.ctc_w, u + 3 < NU ? u + 3 : NU - 1, ring[(j + 3) & 3]);  // unconditional: see ffn()
          if (u < NU) {
            const float4 bb = bnext;
            bnext = *(const float4*)(a.ctc_b + (u + 1 < NU ? u + 1 : u) * 64 + ncol);
            f32x4 c[2];
            mma_k(ring[j], false, c);
            ctc_update(cprev, bprev, uprev);
            cprev[0] = c[0];
            cprev[1] = c[1];
            bprev = bb;
            uprev = u;
          }
        }
      }
      ctc_update(cprev, bprev, uprev);
      float* const cred = red0 + (nln & 1) * 256;  // [4 waves][32 frames] value, then label
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
          const float ov = __shfl_xor(best[mi], o, 64);
          const int oi = __shfl_xor(bidx[mi], o, 64);
          if (ov > best[mi] || (ov == best[mi] && oi < bidx[mi])) {
            best[mi] = ov;
            bidx[mi] = oi;
          }
        }
        if (lg == 0) {
          cred[nf * 32 + mi * 16 + lr] = best[mi];
          ((int*)cred)[128 + nf * 32 + mi * 16 + lr] = bidx[mi];
        }
      }
      bar(BAR_LAST);
      if (nf == 0 && lg == 0) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int m = mi * 16 + lr;
          float bv = cred[m];
          int bi = ((const int*)cred)[128 + m];
#pragma unroll
          for (int w = 1; w < 4; ++w) {
            const float ov = cred[w * 32 + m];
            const int oi = ((const int*)cred)[128 + w * 32 + m];
            if (ov > bv || (ov == bv && oi < bi)) {
              bv = ov;
              bi = oi;
            }
          }
          if (row_ok[mi]) a.ctc_ids[mrow[mi]] = bi;
        }
      }
      touch_done();
      return;
    }
    if constexpr (!HAS_A && !FINAL) {  // the D part alone (streaming layers: the context hand-over sits between D and A)
      store_x();
      // (one block per stream and call: the block's last slot is this call's context vector of the layer - written
      // here instead of by a hand-over launch, EmBlockArgs.last_dst)
      if (a.last_dst) {
        float* const dst = a.last_dst + (size_t)b * a.row_stride;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          if (t0 + mi * 16 + lr == T - 1) {
#pragma unroll
            for (int f = 0; f < 4; ++f) *(float4*)(dst + 64 * f + ncol) = xr[mi][f];
          }
      }
      touch_done();
      return;
    }
  } else {
    // (A alone.  One block per stream and call of the streaming layers: slot 0 is the previous call's context vector of the
    // layer in front, read from where that call left it - EmBlockArgs.row0_src - instead of a hand-over launch copying it
    // into x first; a select on the address, no branch around the loads)
    const bool ctx_slot = a.row0_src != nullptr && t0 == 0 && lr == 0;
    const float* const s0 = ctx_slot ? a.row0_src + (size_t)b * a.row_stride : a.x + mrow[0] * D;
#pragma unroll
    for (int f = 0; f < 4; ++f) xr[0][f] = *(const float4*)(s0 + 64 * f + ncol);
#pragma unroll
    for (int f = 0; f < 4; ++f) xr[1][f] = *(const float4*)(a.x + mrow[1] * D + 64 * f + ncol);
  }

  if (HAS_A) {
    // GA (buffer 0): [norm_ff_macaron g 256][b 256][ffm b1 1024][ffm b2 256]
    // GA+1 (buffer 1): [norm_mha g 256][b 256][bq | bk | bv 768]
    // norm_ff_macaron.  After a D part, buffer 1 still holds G1 (norm_final read it just before this
    // LayerNorm's first barrier): its successor may come in at this LayerNorm's last barrier.
    stamp(30);
    const float* const pa0 = HAS_D ? pb2 : pb0;  // GA
    const float* const pa1 = HAS_D ? pb3 : pb1;  // GA + 1
    if constexpr (HAS_D) {  // (the macaron module's first units were requested by the feed-forward module in front of it)
      ln_to_act(pa0, 0, 256, 0);
    } else {
      touch();
      ln_to_act_pre(pa0, 0, 256, 0,
                    [&] { read_unit(ffm_w1, 0, ring[0]); read_unit(ffm_w1, 1, ring[1]); read_w2(ffm_w2, 0, 0, ring[2]); }, std::integral_constant<int, 24>{},
                    [&] { read_w2(ffm_w2, 0, 1, ring[3]); }, std::integral_constant<int, 8>{});
    }
    stamp(15);
    // x += 0.5 * FFN_macaron(norm_ff_macaron(x))  (encoder_layer.py:108-121); the ring goes over to the q / k / v projections
    ffn(pa0, 512, 1536, 0.5f, ffm_w1, ffm_w2, true, ffm_b1g, a.wqkv, [&] { read_unit(a.wqkv, 2, ring[2]); });
    if constexpr (RELU) {
      if (ffn_exit) return;
    }
    stamp(22);
    // (x is stored at the very END of the kernel, from its LDS parking place: its eight 16-byte stores per lane, issued
    // in front of the q / k / v weight requests, sat in the same in-order count the waits for those requests use and
    // their slow acknowledgement from L2 stalled the stream - norm_mha took 4 300 cycles to its first barrier against
    // 3 100 for the other LayerNorms.  The q / k / v stores themselves stay inside the stream: batched behind it they
    // cost MORE, a store-issue tail of 3 300 cycles that the stream otherwise hides; profiles/r03c_*)
    ln_to_act(pa1, 0, 256, 0);                     // norm_mha (encoder_layer.py:123-127)
    stamp(15);
    // q / k / v projections (attention.py:91-97), written per head: Q, K as [B][H][Tpad][64], V transposed
    // as [B][H][64][Tpad] (computed with the MFMA operands swapped so a lane holds 4 consecutive frames).
    const int H = D / 64;
    stream_k(a.wqkv, std::integral_constant<int, 12>{}, [&](const WF& cur, int u) {
      const int which = u >> 2, head = u & 3;
      const size_t bh = (size_t)b * H + head;
      f32x4 c[2];
      mma_k(cur, which == 2, c);
      // The two 8-byte pieces of a lane (frame halves mi = 0, 1) become ONE 16-byte store: lanes l and l ^ 16 - lane
      // groups 2 j and 2 j + 1, adjacent 4-column (q, k) resp. 4-frame (v) pieces of the same rows - trade halves through
      // v_permlane16_swap, the even group keeps half mi = 0 of both, the odd group half mi = 1.  Half as many store
      // instructions in the kernel's store-bound tail (every CU writes its 80 KiB of q, k, v and x within the same
      // few microseconds; 16-byte stores halve the issue side of it, MI355X_MICROARCH.md "store tail").
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      u32x2 h[2];
      if (which < 2) {
        const float4 b4 = *(const float4*)(pa1 + 512 + u * 64 + ncol);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const bf16x4 pk = {(bf16)(c[mi][0] + b4.x), (bf16)(c[mi][1] + b4.y), (bf16)(c[mi][2] + b4.z),
                             (bf16)(c[mi][3] + b4.w)};
          h[mi] = __builtin_bit_cast(u32x2, pk);
        }
      } else {
        const float bvv = pa1[512 + u * 64 + nf * 16 + lr];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const bf16x4 pk = {(bf16)(c[mi][0] + bvv), (bf16)(c[mi][1] + bvv), (bf16)(c[mi][2] + bvv),
                             (bf16)(c[mi][3] + bvv)};
          h[mi] = __builtin_bit_cast(u32x2, pk);
        }
      }
      // swap(A = half 0, B = half 1): A' = [A.row0, B.row0, A.row2, B.row2], B' = [A.row1, B.row1, A.row3, B.row3]
      // -> even lane groups: (own half 0, partner's half 0) = (A', B'); odd groups: (partner's half 1, own half 1) = (A', B')
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const auto r = __builtin_amdgcn_permlane16_swap(h[0][e], h[1][e], false, false);
        o[e] = r[0];
        o[2 + e] = r[1];
      }
      const int odd = lg & 1;  // this lane now holds half mi = odd of lane groups (lg & ~1) and (lg | 1)
      // a.kv_frag (round 6, for block<ATT|C>): K and V^T go out FRAGMENT-MAJOR per 64-key tile, the order in which the
      // attention's MFMAs take them (EmBlockArgs.kv_frag in the header) - the same 16-byte pieces at other addresses
      if (which == 0 || (which == 1 && !a.kv_frag)) {
        bf16* const dst = (bf16*)(which ? a.kh : a.qh) + (bh * a.Tpad + t0 + odd * 16 + lr) * 64 + nf * 16 + (lg & ~1) * 4;
        *(u32x4*)dst = o;
      } else if (which == 1) {
        const int j = t0 + odd * 16 + lr, jl = j & 63;                                  // key, columns 16 nf + 4 (lg & ~1) .. + 7
        const int n = 2 * (jl >> 5) + ((jl >> 2) & 1), lrk = 4 * ((jl >> 3) & 3) + (jl & 3);  // fragment and row of key jl of a tile
        const int lgk = (nf & 1) * 2 + (lg >> 1);                                        // k-slice of the 32-deep step nf >> 1
        unsigned char* const dst = (unsigned char*)a.kh + bh * a.Tpad * 128 + (size_t)(j >> 6) * 8192 + n * 2048 + (nf >> 1) * 1024 +
                                   (16 * lgk + lrk) * 16;
        *(u32x4*)dst = o;
      } else if (!a.kv_frag) {
        bf16* const dst = (bf16*)a.vt + (bh * 64 + nf * 16 + lr) * a.Tpad + t0 + odd * 16 + (lg & ~1) * 4;
        *(u32x4*)dst = o;
      } else {
        const int j = t0 + odd * 16 + (lg & ~1) * 4, jl = j & 63;                        // eight consecutive keys of row 16 nf + lr
        unsigned char* const dst = (unsigned char*)a.vt + bh * 64 * (size_t)a.Tpad * 2 + (size_t)(j >> 6) * 8192 + nf * 2048 +
                                   (jl >> 5) * 1024 + (16 * ((jl & 31) >> 3) + lr) * 16;
        *(u32x4*)dst = o;
      }
    });
    {
      const float4* const xp = (const float4*)(smem + TILE_OFF) + tid;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int f = 0; f < 4; ++f) xr[mi][f] = xp[(mi * 4 + f) * NT];
      store_x();
    }
    stamp(40);
  }
  touch_done();
}

template <int MODE, int KWT = 31, bool RELU = false>
int launch_block(const EmBlockArgs* a, hipStream_t s) {
  static EmLdsCap cap = {};
  if (em_raise_lds_cap((const void*)block_kernel<MODE, KWT, RELU>, SMEM_BYTES, &cap) != EM_OK) return EM_ERR_LAUNCH;
  // helper workgroups for launches smaller than the chip (see the kernel: L2 warm-up shares); ESPNET_AMD_BLOCK_NO_HELPERS:
  // developer A/B switch
  const bool no_helpers = em_sw().block_no_helpers;
  static int ncu_of[64];  // compute units per device (asked once: no runtime call on later launches, legal under stream capture)
  const int gx = em_cdiv(a->T, BM);
  // grid z: the shares of a split FFN (streaming launches with an FFN; EmBlockArgs.ffn_split)
  const int zs = (RELU && (MODE & (EM_BLOCK_A | EM_BLOCK_D)) != 0 && a->ffn_split > 1) ? a->ffn_split : 1;
  int helper_rows = 0;
  if (!no_helpers) {
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
      if (ncu_of[dev] == 0) {
        hipDeviceProp_t prop;
        ncu_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
      }
      ncu = ncu_of[dev];
    }
    const long nwg = (long)gx * a->B * zs;
    if (nwg < ncu) helper_rows = (int)((ncu - nwg) / ((long)gx * zs));
  }
  dim3 grid(gx, a->B + helper_rows, zs);
  static long long* stamps = nullptr;
  const bool want_stamps = em_sw().block_stamps;
  if (want_stamps && !stamps) hipMalloc((void**)&stamps, 256 * sizeof(long long));
  if (want_stamps) hipMemsetAsync(stamps, 0, 256 * sizeof(long long), s);
  hipLaunchKernelGGL((block_kernel<MODE, KWT, RELU>), grid, dim3(NT), SMEM_BYTES, s, *a, stamps);
  if (want_stamps) {
    long long h[256];
    hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
    const long long M56 = 0xffffffffffffffll;
    for (int wv = 0; wv < (EM_BLOCK_FINE ? 4 : grid.z > 1 ? (int)(grid.z < 4 ? grid.z : 4) : 1); ++wv) {  // (waves, or the shares of a split FFN)
      printf("[block<%d> wave %d stamps code:cycles since entry]", MODE, wv);
      for (int i = 0; i < 64 && h[wv * 64 + i]; ++i)
        printf(" %d:%lld", (int)(h[wv * 64 + i] >> 56), (h[wv * 64 + i] & M56) - (h[wv * 64] & M56));
      printf("\n");
    }
    fflush(stdout);
  }
  EM_CHECK_LAUNCH();
  return EM_OK;
}

// Position rows of the relative-position attention, fragment-major for block<ATT|C>: fragment g of head h holds the rows
// c = 16 g + T - 32 gx - 32 + lr (gx = ceil(T / 32); clamped to [0, 2T - 2]: only masked entries see a clamped row), so that the
// six fragments a (query block t0, key tile j0) rectangle reaches are g = 2 gx - t0 / 16 + j0 / 16 .. + 5 - 12 KiB contiguous.
__global__ void relpos_pack_pos_kernel(const bf16* __restrict__ pall, int ldp, int T, int gx, int npg, bf16* __restrict__ out) {
  const int g = blockIdx.x, h = blockIdx.y, l = blockIdx.z, tid = threadIdx.x;  // 128 threads: (ks, lane)
  const int ks = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
  int c = 16 * g + T - 32 * gx - 32 + lr;
  c = c < 0 ? 0 : (c > 2 * T - 2 ? 2 * T - 2 : c);
  const uint4 v = *(const uint4*)(pall + (size_t)c * ldp + l * D + h * 64 + ks * 32 + lg * 8);
  *(uint4*)(out + ((((size_t)l * 4 + h) * npg + g) * 2 + ks) * 512 + lane * 8) = v;
}

}  // namespace
extern "C" int em_relpos_pos_fragments(int32_t T) { return T > 0 ? 2 * em_cdiv(T, BM) + 4 * em_cdiv(T, 64) + 2 : 0; }
extern "C" int em_relpos_pack_pos_bf16(const void* pall, int32_t ldp, int32_t T, int32_t L, void* out, void* stream) {
  if (!pall || !out || T <= 0 || L <= 0 || ldp < L * D || ldp % 8 != 0) return EM_ERR_BAD_ARG;
  const int npg = em_relpos_pos_fragments(T);
  hipLaunchKernelGGL(relpos_pack_pos_kernel, dim3(npg, 4, L), dim3(128), 0, (hipStream_t)stream, (const bf16*)pall, ldp, T,
                     em_cdiv(T, BM), npg, (bf16*)out);
  EM_CHECK_LAUNCH();
  return EM_OK;
}
extern "C" int em_conformer_block_fused(int mode, const EmBlockArgs* a, void* stream) {
  if (!a || !a->x || !a->params || a->B <= 0 || a->T <= 0) return EM_ERR_BAD_ARG;
  const bool relu = (mode & EM_BLOCK_RELU) != 0;  // streaming layers: ReLU feed-forward, conv width 15, ff up to 4096
  mode &= ~EM_BLOCK_RELU;
  if (a->d != D || a->ff <= 0 || a->ff % 64 != 0 || a->ff > 4096) return EM_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const bool need_a = (mode & EM_BLOCK_A) != 0, need_d = (mode & EM_BLOCK_D) != 0;
  // the first FFN bias lives in the 7 KiB parameter group up to ff = 1024; wider FFNs hand it over in global memory
  if (!relu && a->ff > 1024 && (need_a || need_d)) return EM_ERR_UNSUPPORTED;  // (the C part has no FFN)
  if (relu && ((need_a && !a->ffm_b1g) || (need_d && !a->ff_b1g))) return EM_ERR_BAD_ARG;  // (these instantiations always read it there)
  if (relu && mode != EM_BLOCK_A && mode != EM_BLOCK_D && mode != (EM_BLOCK_ATT | EM_BLOCK_C)) return EM_ERR_UNSUPPORTED;  // (the instantiations that exist)
  if (a->ffn_split > 1 && (need_a || need_d)) {  // split FFN (streaming instantiations): whole pairs of 64-wide chunks per share; a launch without an FFN ignores it
    const int nch_all = ((a->ff >> 6) + 1) & ~1;
    if (!relu || a->ffn_split > 16 || nch_all % (2 * a->ffn_split) != 0 || !a->ffn_part || !a->ffn_ticket) return EM_ERR_BAD_ARG;
  }
  if (!relu && mode == EM_BLOCK_D) return EM_ERR_UNSUPPORTED;
  if (need_a) {
    if (!a->ffm_w1 || !a->ffm_w2 || !a->wqkv || !a->qh || !a->kh || !a->vt) return EM_ERR_BAD_ARG;
    if (a->Tpad % 64 != 0 || a->Tpad < em_cdiv(a->T, BM) * BM) return EM_ERR_BAD_ARG;
  }
  if (need_d) {
    if ((!a->glu && !(mode & EM_BLOCK_C)) || !a->pw2 || !a->ff_w1 || !a->ff_w2 || !a->dw_w || !a->dw_b) return EM_ERR_BAD_ARG;
    if (a->kernel != (relu ? 15 : 31)) return EM_ERR_UNSUPPORTED;
  }
  if (mode == EM_BLOCK_C && (!a->ctx || !a->glu || !a->wout || !a->pw1f)) return EM_ERR_BAD_ARG;
  if (mode & EM_BLOCK_ATT) {  // attention + the C part in one launch (round 6)
    if (mode != (EM_BLOCK_ATT | EM_BLOCK_C)) return EM_ERR_UNSUPPORTED;
    if (!a->glu || !a->wout || !a->pw1f || !a->qh || !a->kh || !a->vt) return EM_ERR_BAD_ARG;
    if (a->Tpad % 64 != 0 || a->Tpad < em_cdiv(a->T, BM) * BM) return EM_ERR_BAD_ARG;
    if (relu) {  // streaming layers: plain attention over the block's <= 64 slots, q / k / v row-major
      if (a->T > 64 || a->kv_frag) return EM_ERR_UNSUPPORTED;
    } else if (!a->pos || !a->pos_u || !a->pos_v || !a->klens || a->ldp != em_relpos_pos_fragments(a->T)) {
      return EM_ERR_BAD_ARG;
    }
  }
  if ((mode & EM_BLOCK_C) && need_d) {  // the C part folded into the launch (round 4)
    if (!a->ctx || !a->wout || !a->pw1f || !a->params_c) return EM_ERR_BAD_ARG;
    if (need_a && (!a->x_out || a->x_out == a->x)) return EM_ERR_BAD_ARG;  // neighbours read x as their halo: no in-place update
  }
  if ((mode & EM_BLOCK_FINAL) && (!a->enc_out || !a->enc_act)) return EM_ERR_BAD_ARG;
  if (mode & EM_BLOCK_CTC) {
    if ((mode & ~EM_BLOCK_C) != (EM_BLOCK_D | EM_BLOCK_FINAL | EM_BLOCK_CTC) || !a->ctc_w || !a->ctc_b || !a->ctc_ids) return EM_ERR_BAD_ARG;
    if (a->ctc_units <= 0) return EM_ERR_BAD_ARG;
  }
  // algorithmic flops of the GEMM-shaped stages (the depthwise conv and LayerNorms are VALU work)
  const double M = (double)a->B * a->T;
  double flops = 0.0;
  if (mode & EM_BLOCK_C) flops += 2.0 * M * D * (D + 2 * D);
  if (mode & EM_BLOCK_ATT) flops += 6.0 * a->B * 4 * (double)a->T * a->T * 64;
  if (mode & EM_BLOCK_D) flops += 2.0 * M * D * (D + 2.0 * a->ff);
  if (mode & EM_BLOCK_A) flops += 2.0 * M * D * (2.0 * a->ff + 3 * D);
  if (mode & EM_BLOCK_CTC) flops += 2.0 * M * D * 64.0 * a->ctc_units;
  const bool rec = em_prof_begin(stream);
  int rc = EM_ERR_BAD_ARG;
  if (relu) {
    if (mode == EM_BLOCK_A) rc = launch_block<EM_BLOCK_A, 31, true>(a, s);
    else if (mode == (EM_BLOCK_ATT | EM_BLOCK_C)) rc = launch_block<EM_BLOCK_ATT | EM_BLOCK_C, 31, true>(a, s);
    else rc = launch_block<EM_BLOCK_D, 15, true>(a, s);
    if (rec) em_prof_end(stream, flops, EM_PROF_BLOCK);
    return rc;
  }
  switch (mode) {
    case EM_BLOCK_C: rc = launch_block<EM_BLOCK_C>(a, s); break;
    case EM_BLOCK_ATT | EM_BLOCK_C: rc = launch_block<EM_BLOCK_ATT | EM_BLOCK_C>(a, s); break;
    case EM_BLOCK_A: rc = launch_block<EM_BLOCK_A>(a, s); break;
    case EM_BLOCK_D | EM_BLOCK_A: rc = launch_block<EM_BLOCK_D | EM_BLOCK_A>(a, s); break;
    case EM_BLOCK_D | EM_BLOCK_FINAL: rc = launch_block<EM_BLOCK_D | EM_BLOCK_FINAL>(a, s); break;
    case EM_BLOCK_D | EM_BLOCK_FINAL | EM_BLOCK_CTC:
      rc = launch_block<EM_BLOCK_D | EM_BLOCK_FINAL | EM_BLOCK_CTC>(a, s);
      break;
    case EM_BLOCK_C | EM_BLOCK_D | EM_BLOCK_A: rc = launch_block<EM_BLOCK_C | EM_BLOCK_D | EM_BLOCK_A>(a, s); break;
    case EM_BLOCK_C | EM_BLOCK_D | EM_BLOCK_FINAL: rc = launch_block<EM_BLOCK_C | EM_BLOCK_D | EM_BLOCK_FINAL>(a, s); break;
    case EM_BLOCK_C | EM_BLOCK_D | EM_BLOCK_FINAL | EM_BLOCK_CTC:
      rc = launch_block<EM_BLOCK_C | EM_BLOCK_D | EM_BLOCK_FINAL | EM_BLOCK_CTC>(a, s);
      break;
  }
  if (rec) em_prof_end(stream, flops, EM_PROF_BLOCK);
  return rc;
}
