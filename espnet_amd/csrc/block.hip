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
// global_load_lds ring with counted vmcnt and one raw s_barrier per one or two 32 KiB units.
//
// Orientation.  Every GEMM is computed transposed, C^T[n][m] = sum_k W[n][k] A[m][k]: the weight
// fragment is the MFMA A operand, the activation fragment the B operand, so a lane ends up with four
// CONSECUTIVE output columns n of one row m (C/D layout: row = (lane >> 4) * 4 + r, col = lane & 15).
// That makes residual / bias / LayerNorm / bf16 packing 16- or 8-byte operations and lets the
// activation fragments (32 rows x K = 256) stay in registers for a whole GEMM.
//
// Four compute waves, wave w owning the 16-column slice nf = w of every 64-column group and BOTH 16-frame
// halves, plus four loader waves that only issue the LDS-DMA (a global_load_lds costs its issuing wave 60-185
// cycles and the scalar table walk stalls it; neither belongs in the MFMA stream).  The loaders follow the
// compute waves barrier by barrier through a code word in LDS, so the compute path carries no control code
// for the ring at all (tools/experiments/barrier_bench.hip: a bare 8-wave barrier is 48 cycles; the branchy
// shared code path of the first versions spent ~500 per interval).  A "K unit" is 64 weight rows x K = 256 (per wave one weight fragment column x two frame
// fragments: 16 MFMAs); a "W2 unit" is all 256 rows x a 64-deep K slice of the FFN's second matrix
// (four weight fragments x two frame fragments x two k-steps: 16 MFMAs).  Both are 32 KiB = 256 LDS
// rows of 128 B, XOR-swizzled on the source address like csrc/gemm.hip.  The order and addresses of the
// units are a table built by the host per launch (a kernel argument, read with scalar loads), so the
// per-unit control code is a handful of scalar instructions.
#include <stdio.h>
#include <stdlib.h>

#include "em_common.h"

namespace {

constexpr int D = 256;
constexpr int BM = 32;
constexpr int NT = 512;                      // threads per workgroup: 4 compute waves + 4 loader waves
constexpr int NC = 256;                      // compute threads
constexpr int UNIT = 32768;
constexpr int NSLOT = 4;
constexpr int ABUF_OFF = NSLOT * UNIT;       // 16 KiB: LN(x) as four [32][64] k-tiles; the FFN's two H tiles alias tiles 0, 1
constexpr int RED_OFF = ABUF_OFF + 16384;    // 2 x 1 KiB: LayerNorm partials [2][4][32], alternating between consecutive LayerNorms
constexpr int PAR_OFF = RED_OFF + 2048;      // 2 x 7 KiB: bias / LayerNorm vectors, double buffered per group
constexpr int PAR_FLOATS = EM_BLOCK_PARAM_GROUP;
constexpr int PAR_BYTES = PAR_FLOATS * 4;
constexpr int SMEM_BYTES = PAR_OFF + 2 * PAR_BYTES;  // all 163 840 B of the LDS
constexpr int KW = 31, HALF = 15, TROWS = BM + KW - 1;  // depthwise conv: 62-row input tile (lives in ring slot 3)
constexpr int MAX_UNITS = 128;

// code word of a barrier (compute thread 0 -> loader waves)
constexpr int BAR_UNIT = 1;    // this barrier opens the interval of the next unit of the stream
constexpr int BAR_PARAMS = 2;  // the older parameter buffer is dead: bring in the next group
constexpr int BAR_TILE = 4;    // ring slot 3 still holds the depthwise-conv tile
constexpr int BAR_LAST = 8;    // nothing follows: the loader waves leave

// Unit g of the weight stream: byte address of its 32 contiguous KiB; bit 0 set = W2 unit (256 rows of 128 B),
// clear = K unit (64 rows of 512 B).
struct UnitTable {
  unsigned long long u[MAX_UNITS];
};
// Code of every barrier of the launch, in order (host-built, build_schedule): what the loader waves need to know
// about the compute waves' progress.  (They used to read it from an LDS word the compute waves wrote before each
// barrier; a ds_read issued behind a wave's own LDS-DMA instructions waits for those transfers -- +0.16 us per
// unit in tools/experiments/glds_bench.hip -- so the loaders must not touch LDS.)
constexpr int MAX_BARRIERS = 192;
struct Schedule {  // 4 bits per barrier, eight per word: a byte array would be read with a VECTOR load, whose
  unsigned w[MAX_BARRIERS / 8];  // completion wait (vmcnt(0)) drains the loader's whole DMA queue every interval
  __host__ __device__ int get(int k) const { return (w[k >> 3] >> ((k & 7) * 4)) & 15; }
  void set(int k, int code) { w[k >> 3] |= (unsigned)code << ((k & 7) * 4); }
};

// LDS-DMA issued from inline asm: global_load_lds_dwordx4 with a wave-uniform 64-bit base (SGPR pair)
// and a per-lane 32-bit byte offset; the LDS destination goes through M0 (saved / restored: M0 is
// compiler-reserved; s_nop 0 = the wait state an M0 write needs before an LDS-DMA).  Hidden from hipcc on
// purpose: told about an LDS-DMA (the builtin), its waitcnt pass puts s_waitcnt vmcnt(0) in front of every
// ds_read it cannot prove disjoint (here: every parameter / bias read), which drains the weight ring once
// per step.  All completion counting is by hand (vmcnt in step_begin).
__device__ __forceinline__ void glds16x4(const unsigned char* sbase, int o0, int o1, int o2, int o3, unsigned d0) {
  unsigned keep;
  const unsigned d1 = d0 + 1024, d2 = d0 + 2048, d3 = d0 + 3072;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %6\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %5\n\t"
      "s_mov_b32 m0, %7\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, %5\n\t"
      "s_mov_b32 m0, %8\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %3, %5\n\t"
      "s_mov_b32 m0, %9\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %4, %5\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(sbase), "s"(d0), "s"(d1), "s"(d2), "s"(d3)
      : "memory");
}
__device__ __forceinline__ void glds16(const unsigned char* sbase, int voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

// "s" asm operands must be provably wave-uniform: make it explicit
__device__ __forceinline__ const unsigned char* uniform_ptr(const unsigned char* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const unsigned char*)(((unsigned long long)hi << 32) | lo);
}

#define EM_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int MODE>
__global__ __launch_bounds__(NT) void block_kernel(const EmBlockArgs a, const UnitTable tab, const Schedule sched,
                                                   const int total, const int dbg, long long* __restrict__ stamps) {
  using MM = Mma<bf16>;
  constexpr bool HAS_C = (MODE & EM_BLOCK_C) != 0, HAS_D = (MODE & EM_BLOCK_D) != 0;
  constexpr bool HAS_A = (MODE & EM_BLOCK_A) != 0, FINAL = (MODE & EM_BLOCK_FINAL) != 0;
  constexpr bool CTC = (MODE & EM_BLOCK_CTC) != 0;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* const ring = smem;
  unsigned char* const abuf = smem + ABUF_OFF;
  float* const red0 = (float*)(smem + RED_OFF);
  const float* const par = (const float*)(smem + PAR_OFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nf = wave & 3;  // compute wave: 16-column slice of every 64-column group; loader wave: quarter of every unit
  const int lr = lane & 15, lg = lane >> 4;
  const int b = blockIdx.y, t0 = blockIdx.x * BM, T = a.T;

  // ======================= loader waves (4 .. 7) =================================================
  // A loader wave never computes; it follows the compute waves barrier by barrier.  Before barrier k compute
  // thread 0 publishes a code word in sync[k & 1] (BAR_* bits); the loaders read it after the barrier, so their
  // loop needs no knowledge of the stage sequence.  Invariants: `started` units have had their interval opened
  // (a barrier with BAR_UNIT); at any barrier every unit below `started` is finished, so units below
  // started + NSLOT may be in the ring (one fewer while the conv tile occupies the last slot, BAR_TILE).  Before
  // every barrier the loader makes sure unit `started` has landed, whether or not this barrier opens its
  // interval.  Parameter loads are older than the unit loads counted here, and vmcnt retires loads in order.
  if (wave >= 4) {
    __builtin_amdgcn_s_setprio(3);  // few instructions, but a late DMA stalls all eight waves
    const int gc = (lane & 7) ^ (lane >> 3);
    // glds instruction i (0..7) fills LDS rows R = 64 nf + 8 i + (lane >> 3) of the unit, lane l supplying the
    // 16-byte chunk (l & 7) ^ (R & 7) of that row's 128 B.  K unit: LDS row R = kt * 64 + n (kt = 64-deep K tile
    // = nf here, n = weight row inside the unit); W2 unit: LDS row R = n (the host packs the FFN's second matrix
    // as [ff/64][256][64], so a 64-deep K slice of all 256 rows is 32 contiguous KiB).
    int kofs[8], w2ofs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = 8 * i + (lane >> 3);
      kofs[i] = n * (D * 2) + nf * 128 + gc * 16;
      w2ofs[i] = (64 * nf + n) * 128 + gc * 16;
    }
    auto issue_unit = [&](int g, unsigned long long d) {  // d = tab.u[g], fetched one interval earlier
      if (dbg & 2) return;
      const unsigned char* base = uniform_ptr((const unsigned char*)(d & ~1ull));
      const bool w2 = (d & 1ull) != 0;
      const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)((g & (NSLOT - 1)) * UNIT + nf * 8192));
      glds16x4(base, w2 ? w2ofs[0] : kofs[0], w2 ? w2ofs[1] : kofs[1], w2 ? w2ofs[2] : kofs[2],
               w2 ? w2ofs[3] : kofs[3], dst);
      glds16x4(base, w2 ? w2ofs[4] : kofs[4], w2 ? w2ofs[5] : kofs[5], w2 ? w2ofs[6] : kofs[6],
               w2 ? w2ofs[7] : kofs[7], dst + 4096);
    };
    auto issue_params = [&](int grp) {  // 7 KiB group -> LDS buffer grp & 1; wave w moves pieces w and w + 4
      const unsigned char* src = uniform_ptr((const unsigned char*)(a.params + (size_t)grp * PAR_FLOATS));
      const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(PAR_OFF + (grp & 1) * PAR_BYTES));
      glds16(src, nf * 1024 + lane * 16, dst + nf * 1024);
      if (nf < 3) glds16(src, (nf + 4) * 1024 + lane * 16, dst + (nf + 4) * 1024);
    };
    issue_params(0);
    if (!HAS_C) issue_params(1);
    int gi = 0, started = 0, pgrp = 2;
    for (; gi < NSLOT - 1 && gi < total; ++gi) issue_unit(gi, tab.u[gi]);
    // The loader's own chain per interval is what paces the ring when it is long (measured: barrier -> LDS read
    // of the code word -> scalar load of the table entry -> 8 DMA issues = 1 100 cycles, and every interval
    // waited for it whether or not the compute waves had anything to do).  So: the table entry of the next unit
    // is fetched an interval ahead, the unit is issued IMMEDIATELY after the barrier (what may be issued depends
    // only on the codes of EARLIER barriers: every unit below `started` is finished at any barrier), and this
    // barrier's code word is read afterwards, while the DMA is already on its way.
    unsigned long long dnext = tab.u[gi < total ? gi : 0];
    int code = sched.get(0);
    for (int k = 0;; ++k) {
      const int need = started + 1 < gi ? started + 1 : gi;  // units below `need` must have landed
      const int later = (dbg & 2) ? 3 : gi - need;
      if (later >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (later == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // every unit below `started` is finished; the conv tile occupies slot 3 while BAR_TILE
      const int cap = started + ((code & BAR_TILE) ? NSLOT - 1 : NSLOT);
      if (gi < total && gi < cap) {
        issue_unit(gi, dnext);
        ++gi;
        dnext = tab.u[gi < total ? gi : 0];
      }
      for (; gi < total && gi < cap; ++gi) {  // more than one free slot: only around stage boundaries
        issue_unit(gi, tab.u[gi]);
        dnext = tab.u[gi + 1 < total ? gi + 1 : 0];
      }
      if (code & BAR_PARAMS) issue_params(pgrp++);
      if (code & BAR_UNIT) ++started;
      if (code & BAR_LAST) break;
      code = sched.get(k + 1);
    }
    return;
  }

  // ======================= compute waves (0 .. 3) ===============================================
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
  const int nch = a.ff >> 6;  // 64-wide chunks of the FFN hidden dimension
  // developer profiling (EM_BLOCK_STAMPS / EM_BLOCK_DBG, tools/block_bench.py): stage-level cycle stamps of thread 0
  // of workgroup (0, 0); dbg 1 = no MFMA / epilogue work, dbg 2 = no DMA
  int nts = 0;
  auto stamp = [&]() {
    if (stamps && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && nts < 64) stamps[nts] = (long long)__builtin_amdgcn_s_memtime();
    ++nts;
  };
  stamp();
  int gs = 0;    // units whose interval has been opened; unit g lives in ring slot g & 3
  int nbar = 0;  // barriers passed
  // every barrier goes through here: publish its code for the loaders, retire own LDS operations, synchronise
  auto bar = [&](int code) {  // `code` documents what build_schedule() tells the loaders about this barrier
    (void)code;
    ++nbar;
    EM_LGKM0();
    __builtin_amdgcn_s_barrier();
  };

  // ---- row state ------------------------------------------------------------------------------
  float4 xr[2][4];     // residual x[frame mi][64 f + ncol .. + 3], f32
  bf16x8 act[2][8];    // activation fragments of the running GEMM: frame mi*16 + lr, k = 32 ks + 8 lg ..
  auto load_x = [&]() {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int f = 0; f < 4; ++f) xr[mi][f] = *(const float4*)(a.x + mrow[mi] * D + 64 * f + ncol);
  };
  auto store_x = [&]() {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
      if (row_ok[mi]) {
#pragma unroll
        for (int f = 0; f < 4; ++f) *(float4*)(a.x + mrow[mi] * D + 64 * f + ncol) = xr[mi][f];
      }
  };
  auto load_act = [&]() {  // from abuf, after a barrier
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
  auto ln_stats = [&](float mean[2], float rstd[2], int code) {
    float* const red = red0 + (nln & 1) * 256;
    ++nln;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      float s = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f) s += (xr[mi][f].x + xr[mi][f].y) + (xr[mi][f].z + xr[mi][f].w);
      const float m0 = s * (1.0f / 16.0f);
      float q = 0.f;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const float dx = xr[mi][f].x - m0, dy = xr[mi][f].y - m0, dz = xr[mi][f].z - m0, dw = xr[mi][f].w - m0;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
      // combine equal-sized groups: n doubles, M2 = M2a + M2b + (sa - sb)^2 / (2 n)
      float so = __shfl_xor(s, 16, 64), qo = __shfl_xor(q, 16, 64);
      q = q + qo + (s - so) * (s - so) * (1.0f / 32.0f);
      s += so;
      so = __shfl_xor(s, 32, 64);
      qo = __shfl_xor(q, 32, 64);
      q = q + qo + (s - so) * (s - so) * (1.0f / 64.0f);
      s += so;
      if (lg == 0) {
        red[nf * 32 + mi * 16 + lr] = s;        // sum over this wave's 64 columns
        red[128 + nf * 32 + mi * 16 + lr] = q;  // sum of squares about their mean
      }
    }
    bar(code);
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
      rstd[mi] = 1.0f / sqrtf(qq * (1.0f / D) + a.eps);
    }
  };
  // y = LN(x; g, b) with g, b at float offsets go / bo of parameter buffer pb
  auto ln_apply = [&](const float* pb, int go, int bo, float4 y[2][4], int code) {
    float mean[2], rstd[2];
    ln_stats(mean, rstd, code);
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
  // LN(x) -> bf16 -> abuf -> activation fragments.  One more barrier, carrying `code` (the parameter reads of
  // this LayerNorm precede it).
  auto ln_to_act = [&](const float* pb, int go, int bo, int code) {
    float4 y[2][4];
    ln_apply(pb, go, bo, y, 0);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        bf16x4 pk = {(bf16)y[mi][f].x, (bf16)y[mi][f].y, (bf16)y[mi][f].z, (bf16)y[mi][f].w};
        *(bf16x4*)(abuf + f * 4096 + mi * 2048 + tile_wr) = pk;
      }
    bar(code);
    load_act();
  };
  // ---- units: fragment reads run ONE INTERVAL AHEAD of the MFMAs ---------------------------------------
  // open(): the barrier that makes the next unit of the stream visible (and releases the previous one: its
  // fragments are in registers by then); read_k / read_w2: its eight fragments LDS -> registers.  A stage
  // processes unit u from registers while the reads of unit u + 1 are in flight, so the LDS latency hides behind
  // matrix work instead of opening every interval (a single wave per SIMD has no other wave to hide it with).
  struct WF {
    bf16x8 v[8];
  };
  auto open = [&](int code) -> const unsigned char* {
    bar(BAR_UNIT | code);
    const unsigned char* su = ring + (gs & (NSLOT - 1)) * UNIT + (nf * 16 + lr) * 128;
    ++gs;
    return su;
  };
  auto read_k = [&](const unsigned char* su, WF& w) {  // K unit: fragment ks at v[ks]
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) w.v[ks] = *(const bf16x8*)(su + (ks >> 1) * 8192 + ((((ks & 1) * 4 + lg) ^ swz) << 4));
  };
  auto read_w2 = [&](const unsigned char* su, WF& w) {  // W2 unit: fragment (f, ks) at v[2 f + ks]
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) w.v[2 * f + ks] = *(const bf16x8*)(su + f * 8192 + (((ks * 4 + lg) ^ swz) << 4));
  };
  // 16 MFMAs of a K unit: out[mi] = C^T[n = nf*16 + lg*4 + r][m = mi*16 + lr]
  // (swap: C[m = mi*16 + lg*4 + r][n = nf*16 + lr])
  auto mma_k = [&](const WF& w, bool swap, f32x4 out[2]) {
    if (dbg & 1) {
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
        c[mi][ks & 1] = swap ? MM::mma(act[mi][ks], w.v[ks], c[mi][ks & 1]) : MM::mma(w.v[ks], act[mi][ks], c[mi][ks & 1]);
    out[0] = c[0][0] + c[0][1];
    out[1] = c[1][0] + c[1][1];
  };
  // LN(x) -> bf16 -> abuf; the publishing barrier also opens the first unit of the stage that consumes it
  auto ln_to_act_open = [&](const float* pb, int go, int bo, int code, WF& w0) {
    float4 y[2][4];
    ln_apply(pb, go, bo, y, 0);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        bf16x4 pk = {(bf16)y[mi][f].x, (bf16)y[mi][f].y, (bf16)y[mi][f].z, (bf16)y[mi][f].w};
        *(bf16x4*)(abuf + f * 4096 + mi * 2048 + tile_wr) = pk;
      }
    const unsigned char* su = open(code);
    read_k(su, w0);
    load_act();
  };
  // x += scale * (W2 . swish(W1 . act + b1) + b2); b1 at pb + b1o, b2 at pb + b2o.  Units in the order
  // K0 K1 W2_0 K2 W2_1 ... K_last W2_last-1 W2_last (build_units); K0 has been opened by the caller, its
  // fragments are arriving in wa.  Interval by interval (r = fragment reads of the unit just opened):
  //   r K1   | mma K0 -> h0
  //   r W2_0 | Swish(h0) -> H[0] | mma K1 -> h1
  //   r K2   | Swish(h1) -> H[1] | mma W2_0 (H[0])
  //   r W2_1 |                     mma K2 -> h2           ... and so on; the last interval has nothing to read.
  // The Swish epilogue of chunk c (bias, exp, rcp, bf16 pack, LDS store of H[c & 1]) always shares an interval
  // with the MFMAs of another unit.  2 nch barriers, the last one plain.  The two H tiles [32][64] alias abuf's
  // first two k-tiles (the activation fragments are in registers by then).
  auto ffn = [&](const float* pb, int b1o, int b2o, float scale, WF& wa) {
    f32x4 acc2[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int f = 0; f < 4; ++f) acc2[mi][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto h_store = [&](const f32x4 h[2], int c) {
      if (dbg & 1) return;
      const float4 bb = *(const float4*)(pb + b1o + c * 64 + ncol);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        bf16x4 pk = {(bf16)swishf_(h[mi][0] + bb.x), (bf16)swishf_(h[mi][1] + bb.y),
                     (bf16)swishf_(h[mi][2] + bb.z), (bf16)swishf_(h[mi][3] + bb.w)};
        *(bf16x4*)(abuf + (c & 1) * 4096 + mi * 2048 + tile_wr) = pk;  // H[c & 1][m][k = ncol ..]
      }
    };
    auto mma_w2 = [&](const WF& w, int c) {
      if (dbg & 1) return;
      const unsigned char* sh = abuf + (c & 1) * 4096 + lr * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff = ((ks * 4 + lg) ^ swz) << 4;
        const bf16x8 h0 = *(const bf16x8*)(sh + coff);
        const bf16x8 h1 = *(const bf16x8*)(sh + 2048 + coff);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          acc2[0][f] = MM::mma(w.v[2 * f + ks], h0, acc2[0][f]);
          acc2[1][f] = MM::mma(w.v[2 * f + ks], h1, acc2[1][f]);
        }
      }
    };
    // An interval that carries a Swish epilogue next to 16 MFMAs: all LDS reads first (next unit's fragments,
    // bias, H fragments), then the epilogue's VALU / transcendental instructions dealt into the issue gaps between
    // the MFMAs (left alone hipcc emits the whole epilogue, then the MFMAs: serial on a single wave per SIMD).
#define EM_INTERLEAVE(NREADS)                                 \
  do {                                                        \
    __builtin_amdgcn_sched_group_barrier(0x100, NREADS, 0);   \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {       \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      \
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      \
    }                                                         \
    __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);        \
  } while (0)
    WF wb;
    f32x4 hp[2];
    if (nch == 1) {
      read_w2(open(0), wb);
      mma_k(wa, false, hp);
      h_store(hp, 0);
      bar(0);
      mma_w2(wb, 0);
    } else {
      read_k(open(0), wb);       // K1
      mma_k(wa, false, hp);      // h0
      read_w2(open(0), wa);      // W2_0
      {
        f32x4 hn[2];
        h_store(hp, 0);
        mma_k(wb, false, hn);    // h1
        EM_INTERLEAVE(9);
        hp[0] = hn[0];
        hp[1] = hn[1];
      }
      for (int c = 0; c + 2 < nch; ++c) {
        read_k(open(0), wb);     // K_{c+2}
        h_store(hp, c + 1);
        mma_w2(wa, c);
        EM_INTERLEAVE(13);
        read_w2(open(0), wa);    // W2_{c+1}
        mma_k(wb, false, hp);    // h_{c+2}
      }
      read_w2(open(0), wb);      // W2_last
      h_store(hp, nch - 1);
      mma_w2(wa, nch - 2);
      EM_INTERLEAVE(13);
      bar(0);
      mma_w2(wb, nch - 1);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const float4 b4 = *(const float4*)(pb + b2o + 64 * f + ncol);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        xr[mi][f].x += scale * (acc2[mi][f][0] + b4.x);
        xr[mi][f].y += scale * (acc2[mi][f][1] + b4.y);
        xr[mi][f].z += scale * (acc2[mi][f][2] + b4.z);
        xr[mi][f].w += scale * (acc2[mi][f][3] + b4.w);
      }
    }
  };
  // x += (W . act + bias): four K units (N = 256), the first already opened into w[0]
  auto proj_resid = [&](const float* pb, int bo, WF& w0) {
    WF w1;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      WF& cur = (f & 1) ? w1 : w0;
      WF& nxt = (f & 1) ? w0 : w1;
      if (f + 1 < 4) read_k(open(0), nxt);
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
    }
  };

  const float* const pb0 = par;
  const float* const pb1 = par + PAR_FLOATS;
  WF w0;  // fragments of the unit a stage starts with

  if (HAS_C) {
    // linear_out over the attention context: activation fragments straight from global memory
    // (attention.py:151 linear_out; encoder_layer.py:142-147 residual)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const bf16* crow = (const bf16*)a.ctx + mrow[mi] * D + lg * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) act[mi][ks] = *(const bf16x8*)(crow + ks * 32);
    }
    load_x();
    // G0: [bout 256][norm_conv g 256][b 256][pw1 bias, fused order 512]
    read_k(open(0), w0);
    proj_resid(pb0, 0, w0);
    store_x();
    ln_to_act_open(pb0, 256, 512, 0, w0);
    // pointwise_conv1 + GLU (convolution.py:66-69): unit 2j = value rows 64j.., unit 2j+1 = their gates
    WF w1;
    f32x4 v[2], gt[2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      WF& cur = (u & 1) ? w1 : w0;
      WF& nxt = (u & 1) ? w0 : w1;
      if (u + 1 < 8) read_k(open(u + 2 == 8 ? BAR_LAST : 0), nxt);
      if (!(u & 1)) {
        mma_k(cur, false, v);
      } else {
        mma_k(cur, false, gt);
        const int j = u >> 1;
        const float4 bv = *(const float4*)(pb0 + 768 + (2 * j) * 64 + ncol);
        const float4 bg = *(const float4*)(pb0 + 768 + (2 * j + 1) * 64 + ncol);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          bf16x4 pk = {(bf16)((v[mi][0] + bv.x) * sigmoidf_(gt[mi][0] + bg.x)),
                       (bf16)((v[mi][1] + bv.y) * sigmoidf_(gt[mi][1] + bg.y)),
                       (bf16)((v[mi][2] + bv.z) * sigmoidf_(gt[mi][2] + bg.z)),
                       (bf16)((v[mi][3] + bv.w) * sigmoidf_(gt[mi][3] + bg.w))};
          if (row_ok[mi]) *(bf16x4*)((bf16*)a.glu + mrow[mi] * D + 64 * j + ncol) = pk;
        }
      }
    }
    return;
  }

  if (HAS_D) {
    // ---- depthwise conv (k = 31, zero padded) + folded BatchNorm + Swish (convolution.py:72-75):
    // the 62-row input tile of this block (frames t0 - 15 .. t0 + 46 of the utterance, zero outside
    // [0, Tv)) goes through LDS; thread c (= channel) produces the 32 frames of the block.
    unsigned char* const tile = ring + (NSLOT - 1) * UNIT;  // [62][256] bf16
    int Tv = T;
    if (a.tlens) Tv = a.tlens[b] < T ? a.tlens[b] : T;
    uint4 stage[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {  // unconditional loads from a clamped row (a load under a lane mask is
      const int q = tid + it * NC, r = q >> 5, ch = q & 31;  // waited for on the spot), zeroed afterwards
      const int t = t0 - HALF + r;
      const int tc = t < 0 ? 0 : (t < T ? t : T - 1);
      stage[it] = *(const uint4*)((const bf16*)a.glu + ((size_t)b * T + tc) * D + ch * 8);
    }
    float wk[KW];  // requested together with the tile: one global-memory latency for the whole prologue
#pragma unroll
    for (int k = 0; k < KW; ++k) wk[k] = a.dw_w[k * D + tid];
    const float bc = a.dw_b[tid];
    load_x();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int q = tid + it * NC, t = t0 - HALF + (q >> 5);
      if (q < TROWS * 32) *(uint4*)(tile + q * 16) = (t >= 0 && t < Tv) ? stage[it] : make_uint4(0u, 0u, 0u, 0u);
    }
    bar(BAR_TILE);
    {
      const bf16* col = (const bf16*)tile + tid;
      const int kt = tid >> 6, kl = tid & 63;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {  // two halves of 16 frames: 16 accumulators live at a time
        float acc[16];
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] = bc;
#pragma unroll
        for (int r = 0; r < 16 + KW - 1; ++r) {
          const float v = (float)col[(hh * 16 + r) * D];
#pragma unroll
          for (int o = 0; o < 16; ++o)
            if (r - o >= 0 && r - o < KW) acc[o] = fmaf(wk[r - o], v, acc[o]);
        }
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          const int m = hh * 16 + o;
          *(bf16*)(abuf + (kt * 32 + m) * 128 + (((kl >> 3) ^ (m & 7)) << 4) + (kl & 7) * 2) = (bf16)swishf_(acc[o]);
        }
      }
    }
    read_k(open(0), w0);  // conv output visible, the tile (slot 3) is free, pointwise_conv2's first unit open
    load_act();
    stamp();  // 1 conv prologue
    // G0: [pw2 bias 256][norm_ff g 256][b 256]; G1: [ff b1 1024][ff b2 256][norm_final g 256][b 256]
    proj_resid(pb0, 0, w0);                        // pointwise_conv2 + residual (convolution.py:77, encoder_layer.py:149-158)
    stamp();  // 2
    ln_to_act_open(pb0, 256, 512, BAR_PARAMS, w0); // norm_ff; G0 is dead after it: group 2 replaces it
    stamp();  // 3
    ffn(pb1, 0, 1024, 0.5f, w0);                   // x += 0.5 * FFN(norm_ff(x))   (encoder_layer.py:160-168)
    stamp();  // 4
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
      ln_apply(pb0, 0, 256, y, CTC ? 0 : BAR_LAST);
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
      if (!CTC) return;
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
      read_k(open(0), w0);
      load_act();
      float best[2] = {-INFINITY, -INFINITY};
      int bidx[2] = {0, 0};
      const int NU = a.ctc_units;
      // the bias of unit u + 1 is requested while unit u computes: a global load issued next to its use would
      // put its L2 latency into every one of the 79 intervals
      float4 bnext = *(const float4*)(a.ctc_b + ncol);
      auto ctc_unit = [&](const WF& w, int u) {
        const float4 bb = bnext;
        bnext = *(const float4*)(a.ctc_b + (u + 1 < NU ? u + 1 : u) * 64 + ncol);
        f32x4 c[2];
        mma_k(w, false, c);
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
      WF w1;
      for (int u = 0; u < NU; u += 2) {
        if (u + 1 < NU) read_k(open(0), w1);
        ctc_unit(w0, u);
        if (u + 1 < NU) {
          if (u + 2 < NU) read_k(open(0), w0);
          ctc_unit(w1, u + 1);
        }
      }
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
      return;
    }
  } else {
    load_x();
  }

  if (HAS_A) {
    // GA (buffer 0): [norm_ff_macaron g 256][b 256][ffm b1 1024][ffm b2 256]
    // GA+1 (buffer 1): [norm_mha g 256][b 256][bq | bk | bv 768]
    // norm_ff_macaron.  After a D part, buffer 1 still holds G1 (norm_final read it just before this
    // LayerNorm's first barrier): its successor may come in at this LayerNorm's last barrier.
    stamp();  // 5 norm_final
    ln_to_act_open(pb0, 0, 256, HAS_D ? BAR_PARAMS : 0, w0);
    stamp();  // 6 norm_ff_macaron
    ffn(pb0, 512, 1536, 0.5f, w0);                 // x += 0.5 * FFN_macaron(norm_ff_macaron(x))  (encoder_layer.py:108-121)
    stamp();  // 7 macaron FFN
    store_x();
    ln_to_act_open(pb1, 0, 256, 0, w0);            // norm_mha (encoder_layer.py:123-127)
    stamp();  // 8 norm_mha
    // q / k / v projections (attention.py:91-97), written per head: Q, K as [B][H][Tpad][64], V transposed
    // as [B][H][64][Tpad] (computed with the MFMA operands swapped so a lane holds 4 consecutive frames).
    const int H = D / 64;
    WF w1;
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      WF& cur = (u & 1) ? w1 : w0;
      WF& nxt = (u & 1) ? w0 : w1;
      if (u + 1 < 12) read_k(open(u + 2 == 12 ? BAR_LAST : 0), nxt);
      const int which = u >> 2, head = u & 3;
      const size_t bh = (size_t)b * H + head;
      f32x4 c[2];
      mma_k(cur, which == 2, c);
      if (which < 2) {
        const float4 b4 = *(const float4*)(pb1 + 512 + u * 64 + ncol);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          bf16x4 pk = {(bf16)(c[mi][0] + b4.x), (bf16)(c[mi][1] + b4.y), (bf16)(c[mi][2] + b4.z),
                       (bf16)(c[mi][3] + b4.w)};
          *(bf16x4*)((bf16*)(which ? a.kh : a.qh) + (bh * a.Tpad + t0 + mi * 16 + lr) * 64 + ncol) = pk;
        }
      } else {
        const float bvv = pb1[512 + u * 64 + nf * 16 + lr];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          bf16x4 pk = {(bf16)(c[mi][0] + bvv), (bf16)(c[mi][1] + bvv), (bf16)(c[mi][2] + bvv),
                       (bf16)(c[mi][3] + bvv)};
          *(bf16x4*)((bf16*)a.vt + (bh * 64 + nf * 16 + lr) * a.Tpad + t0 + mi * 16 + lg * 4) = pk;
        }
      }
    }
    stamp();  // 9 q k v
  }
}

// order of the weight units of one launch (see the kernel: pointwise_conv2 | FFN | macaron FFN of the next
// block | q k v;  linear_out | pointwise_conv1)
int build_units(int mode, const EmBlockArgs* a, UnitTable* t) {
  int n = 0;
  auto k_units = [&](const void* w, int count) {
    for (int u = 0; u < count; ++u) t->u[n++] = (unsigned long long)((const unsigned char*)w + (size_t)u * UNIT);
  };
  auto ffn_units = [&](const void* w1, const void* w2) {
    const int nch = a->ff / 64;  // K0, (K1, W2_0), (K2, W2_1), ..., W2_last: the kernel's software pipeline
    for (int c = 0; c <= nch; ++c) {
      if (c < nch) t->u[n++] = (unsigned long long)((const unsigned char*)w1 + (size_t)c * UNIT);
      if (c > 0) t->u[n++] = (unsigned long long)((const unsigned char*)w2 + (size_t)(c - 1) * UNIT) | 1ull;
    }
  };
  if (mode & EM_BLOCK_C) {
    k_units(a->wout, 4);
    k_units(a->pw1f, 8);
  }
  if (mode & EM_BLOCK_D) {
    k_units(a->pw2, 4);
    ffn_units(a->ff_w1, a->ff_w2);
  }
  if (mode & EM_BLOCK_A) {
    ffn_units(a->ffm_w1, a->ffm_w2);
    k_units(a->wqkv, 12);
  }
  if (mode & EM_BLOCK_CTC) k_units(a->ctc_w, a->ctc_units);
  return n;
}

// The barrier sequence of block_kernel<mode>, barrier by barrier (keep in step with the kernel: every bar(code)
// there appears here, in order).
int build_schedule(int mode, const EmBlockArgs* a, Schedule* sc) {
  int n = 0;
  const int nch = a->ff / 64;
  auto put = [&](int code, int count = 1) {
    for (int i = 0; i < count && n < MAX_BARRIERS; ++i) sc->set(n++, code);
  };
  if (mode & EM_BLOCK_C) {
    put(BAR_UNIT, 4);             // linear_out: one barrier opens each unit
    put(0);                       // norm_conv statistics
    put(BAR_UNIT);                // LN(x) published + first unit of pointwise_conv1
    put(BAR_UNIT, 6);
    put(BAR_UNIT | BAR_LAST);
    return n;
  }
  auto ffn_bars = [&]() {         // the first unit is opened by the LayerNorm before; one plain barrier at the end
    put(BAR_UNIT, 2 * nch - 1);
    put(0);
  };
  if (mode & EM_BLOCK_D) {
    put(BAR_TILE);                // conv tile staged
    put(BAR_UNIT);                // conv output published + first unit of pointwise_conv2
    put(BAR_UNIT, 3);
    put(0);                       // norm_ff statistics
    put(BAR_PARAMS | BAR_UNIT);   // norm_ff published, parameter group 0 dead, FFN's first unit
    ffn_bars();
    put(0);                       // norm_final statistics
    if (mode & EM_BLOCK_CTC) {
      put(0);                     // after_norm statistics
      put(BAR_UNIT, a->ctc_units); // rows published + first CTC unit, then one barrier per further unit
      put(BAR_LAST);              // arg-max exchange between the waves
      return n;
    }
    if (mode & EM_BLOCK_FINAL) {
      put(BAR_LAST);              // after_norm statistics
      return n;
    }
  }
  if (mode & EM_BLOCK_A) {
    put(0);                       // norm_ff_macaron statistics
    put(((mode & EM_BLOCK_D) ? BAR_PARAMS : 0) | BAR_UNIT);  // published; after a D part parameter group 1 is dead
    ffn_bars();
    put(0);                       // norm_mha statistics
    put(BAR_UNIT);                // published + first q unit
    put(BAR_UNIT, 10);
    put(BAR_UNIT | BAR_LAST);
  }
  return n;
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
  UnitTable tab = {};
  const int total = build_units(MODE, a, &tab);
  Schedule sched = {};
  build_schedule(MODE, a, &sched);
  dim3 grid(em_cdiv(a->T, BM), a->B);
  static const int dbg = getenv("EM_BLOCK_DBG") ? atoi(getenv("EM_BLOCK_DBG")) : 0;
  static long long* stamps = nullptr;
  static const bool want_stamps = getenv("EM_BLOCK_STAMPS") != nullptr;
  if (want_stamps && !stamps) hipMalloc((void**)&stamps, 64 * sizeof(long long));
  if (want_stamps) hipMemsetAsync(stamps, 0, 64 * sizeof(long long), s);
  hipLaunchKernelGGL((block_kernel<MODE>), grid, dim3(NT), SMEM_BYTES, s, *a, tab, sched, total, dbg, stamps);
  if (want_stamps) {
    long long h[64];
    hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
    printf("[block<%d> stamps, cycles since kernel start]", MODE);
    for (int i = 1; i < 64 && h[i]; ++i) printf(" %lld", h[i] - h[0]);
    printf("\n");
    fflush(stdout);
  }
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
  if ((mode & EM_BLOCK_FINAL) && (!a->enc_out || !a->enc_act)) return EM_ERR_BAD_ARG;
  if (mode & EM_BLOCK_CTC) {
    if (mode != (EM_BLOCK_D | EM_BLOCK_FINAL | EM_BLOCK_CTC) || !a->ctc_w || !a->ctc_b || !a->ctc_ids) return EM_ERR_BAD_ARG;
    // unit table / barrier schedule capacity (kernel arguments): pw2 + FFN + CTC units
    if (a->ctc_units <= 0 || 4 + 2 * (a->ff / 64) + a->ctc_units > MAX_UNITS ||
        8 + 2 * (a->ff / 64) + a->ctc_units + 2 > MAX_BARRIERS)
      return EM_ERR_UNSUPPORTED;
  }
  // algorithmic flops of the GEMM-shaped stages (the depthwise conv and LayerNorms are VALU work)
  const double M = (double)a->B * a->T;
  double flops = 0.0;
  if (mode & EM_BLOCK_C) flops += 2.0 * M * D * (D + 2 * D);
  if (mode & EM_BLOCK_D) flops += 2.0 * M * D * (D + 2.0 * a->ff);
  if (mode & EM_BLOCK_A) flops += 2.0 * M * D * (2.0 * a->ff + 3 * D);
  if (mode & EM_BLOCK_CTC) flops += 2.0 * M * D * 64.0 * a->ctc_units;
  const bool rec = em_prof_begin(stream);
  int rc = EM_ERR_BAD_ARG;
  switch (mode) {
    case EM_BLOCK_C: rc = launch_block<EM_BLOCK_C>(a, s); break;
    case EM_BLOCK_A: rc = launch_block<EM_BLOCK_A>(a, s); break;
    case EM_BLOCK_D | EM_BLOCK_A: rc = launch_block<EM_BLOCK_D | EM_BLOCK_A>(a, s); break;
    case EM_BLOCK_D | EM_BLOCK_FINAL: rc = launch_block<EM_BLOCK_D | EM_BLOCK_FINAL>(a, s); break;
    case EM_BLOCK_D | EM_BLOCK_FINAL | EM_BLOCK_CTC:
      rc = launch_block<EM_BLOCK_D | EM_BLOCK_FINAL | EM_BLOCK_CTC>(a, s);
      break;
  }
  if (rec) em_prof_end(stream, flops, EM_PROF_BLOCK);
  return rc;
}
