// Relative-position multi-head self-attention, LDS-resident form (bf16 MFMA, d_k = 64) for the fused
// Conformer path (csrc/block.hip writes its operands per head).
//
// Reference: RelPositionMultiHeadedAttention.forward
// (espnet2/legacy/nets/pytorch_backend/transformer/attention.py:416-459), rel_shift (:391-408),
// forward_attention (:121-151):
//     AC[i][j] = (q_i + u) . k_j        BD[i][j] = (q_i + v) . p[T-1-i+j]
//     P = softmax_j((AC + BD) / sqrt(d_k)) over keys j < klens[b],   ctx_i = sum_j P[i][j] v_j
//
// One workgroup = (utterance b, head, 128 queries), 8 waves x 16 queries.  For up to 256 keys at a time
// everything a workgroup touches sits in LDS, loaded ONCE with global_load_lds (K 32 KiB, V^T 32 KiB, the
// 383 position rows p[T-1-i+j] the (128 query, 256 key) rectangle can reach: 48 KiB): at T = 249 (10 s
// of audio) that is the whole problem, the key loop has no barrier in it and no operand is fetched twice
// (csrc/attention.hip re-stages K, V and 128 position rows per 64 x 64 tile behind four barriers each).
// Longer inputs walk 256-key super-tiles with the online softmax carried in registers.
//
// Orientation: scores are computed TRANSPOSED, S^T[key][query] = K . (Q+u)^T, so that in the MFMA C/D
// layout a lane holds 4 consecutive keys of ONE query: the softmax statistics are per-lane scalars
// (reduced over the 4 lane groups with two shuffles), and the probabilities are already in the
// B-operand layout of O^T[dk][query] = V^T . P^T -- they never go through LDS.  (The 8 contraction
// slots of lane group g then carry keys {4g..4g+3} and {16+4g..16+4g+3} of a 32-key step; V^T is read to
// match with two 8-byte LDS loads.)  rel_shift is index arithmetic: the dense window
// D[c][i] = p[c] . (q_i + v) over the 80 position rows a (16 query, 64 key) tile can reach is written to
// a per-wave scratch and read back skewed, BD^T[j][i] = D[15 - i + j][i].
//
// Round 5, measured and NOT kept (git: the commit "attention: two 16-query fragments per wave ..."; profiles/r05e_attention_two_
// fragments_ab.txt): two query fragments per wave with the key tiles dealt to the two halves of the workgroup - every K / V^T
// operand fragment read from LDS feeding two MFMAs, the two fragments' position windows sharing 96 rows, the halves'
// online-softmax states merged once at the end - cuts the LDS operand traffic by a third and is SLOWER: 12.15 against
// 11.19 us (T = 249, 4 heads, B = 32), 42.5 against 40.1 us (8 heads, B = 64), same call, same box.  The tile loop is not
// bound by LDS bandwidth; what it waits for is the dependent chain of a tile (operand reads -> MFMAs -> scratch round trip
// -> row maximum -> exponentials -> P . V) with two waves per SIMD to cover it, and half as many, twice as fat waves cover
// it worse.
#include <stdio.h>
#include <stdlib.h>

#include "em_common.h"
#include "switches.h"

namespace {

constexpr int QB = 128;                       // queries per workgroup
constexpr int KSUP = 256;                     // keys per LDS super-tile
constexpr int SK_OFF = 0;                     // [256 keys][128 B], chunks XOR-swizzled by (row & 7)
constexpr int SV_OFF = 32768;                 // [4 key tiles][64 dk][128 B], chunks XOR-swizzled by (row >> 1) & 7
constexpr int SP_OFF = 65536;                 // [384 position rows][128 B]
constexpr int SBD_OFF = SP_OFF + 384 * 128;   // [8 waves][16 queries][84] f32
constexpr int LDB = 84;
constexpr int SMEM_BYTES = SBD_OFF + 8 * 16 * LDB * 4;  // 157 696 B

// LDS-DMA from inline asm (see csrc/block.hip: hidden from hipcc's waitcnt pass on purpose)
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

__global__ __launch_bounds__(512) void relpos_attn2_kernel(
    const bf16* __restrict__ qh, const bf16* __restrict__ kh, const bf16* __restrict__ vt,
    const bf16* __restrict__ p, int ldp, const float* __restrict__ pos_u, const float* __restrict__ pos_v,
    const int* __restrict__ klens, int T, int Tpad, int H, bf16* __restrict__ ctx, long long* __restrict__ stamps) {
  using MM = Mma<bf16>;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4, swz = lr & 7;
  const int hh = blockIdx.y, b = blockIdx.z, i0 = blockIdx.x * QB, iw0 = i0 + 16 * wave;
  const int klen = klens[b] < T ? klens[b] : T;
  const size_t bh = (size_t)b * H + hh;
  // developer timing (EM_ATTN2_STAMPS): cycle stamps of wave 0 of workgroup (0, 1, 3)
  int nts = 0;
  auto stamp = [&]() {
    if (stamps && blockIdx.x == 0 && blockIdx.y == 1 && blockIdx.z == 3 && tid == 0 && nts < 32)
      stamps[nts] = (long long)__builtin_amdgcn_s_memtime();
    ++nts;
  };
  stamp();

  const unsigned char* k_base = uniform_ptr((const unsigned char*)(kh + bh * Tpad * 64));
  const unsigned char* v_base = uniform_ptr((const unsigned char*)(vt + bh * 64 * Tpad));
  const unsigned char* p_base = uniform_ptr((const unsigned char*)(p + hh * 64));
  // ---- staging, in KEY-TILE order (round 3).  A super-tile is four 64-key tiles; the pieces a tile's MFMAs need - its
  // 64 K rows, its 64 columns of V^T and the position rows its (128 query x 64 key) rectangle adds to the window (192
  // for the first tile, 64 for each later one) - are requested tile by tile, 1 KiB per wave-instruction, so that the
  // first tile can be computed once ITS 40 KiB are in LDS while the other 72 KiB are still on their way.  Until round
  // 3 the kernel waited for all 112 KiB (plus the queries) behind one vmcnt(0): at the ~13 B/clk a CU gets while
  // every CU's prologue hits memory at once, that was ~4 us of a 12.8 us launch with the matrix cores idle.
  // A tile's pieces are requested while the tile before it is computed.  V^T is laid out per tile,
  // [tile][64 dk][128 B], 16-byte chunks XOR-swizzled by (row >> 1) & 7 (conflict-free 8-byte operand reads).
  auto stage_tile = [&](int js, int t) {
    const int r8 = lane >> 3, c8 = lane & 7;
    {  // K rows js + 64 t + 8 wave ..
      int row = js + 64 * t + 8 * wave + r8;
      row = row < Tpad ? row : Tpad - 1;
      glds16(k_base, row * 128 + ((c8 ^ r8) << 4), SK_OFF + (8 * t + wave) * 1024);
    }
    const int cbase = T - 1 - (i0 + QB - 1) + js;
    auto prow = [&](int pj) {  // position rows cbase + 8 pj .. (clamped: only masked entries see the clamp)
      int c = cbase + 8 * pj + r8;
      c = c < 0 ? 0 : (c > 2 * T - 2 ? 2 * T - 2 : c);
      glds16(p_base, c * (ldp * 2) + ((c8 ^ r8) << 4), SP_OFF + pj * 1024);
    };
    if (t == 0) {
      prow(3 * wave);
      prow(3 * wave + 1);
      prow(3 * wave + 2);
    } else {
      prow(24 + 8 * (t - 1) + wave);
    }
    {  // V^T: dk rows 8 wave .., keys js + 64 t ..
      const int row = 8 * wave + r8;
      int col = js + 64 * t;
      col = col < Tpad ? col : Tpad - 64;
      glds16(v_base, (row * Tpad + col) * 2 + ((c8 ^ ((row >> 1) & 7)) << 4), SV_OFF + t * 8192 + wave * 1024);
    }
  };

  // ---- query fragments (B operand: column = query iw0 + lr, k-slice lg) and the two position biases: requested
  // BEFORE the staging, from inline asm like it - a load hipcc knows about would be waited for with the count it
  // keeps, which does not include the 14 staging requests issued behind it: s_waitcnt vmcnt(0), i.e. everything.
  typedef __attribute__((ext_vector_type(4))) float f4;
  f4 qraw[2], ur[4], vr[4];
  {
    const bf16* qrow = qh + (bh * Tpad + iw0 + lr) * 64 + lg * 8;
    const float* pu = pos_u + hh * 64 + lg * 8;
    const float* pv = pos_v + hh * 64 + lg * 8;
    asm volatile(
        "global_load_dwordx4 %0, %10, off\n\t"
        "global_load_dwordx4 %1, %10, off offset:64\n\t"
        "global_load_dwordx4 %2, %11, off\n\t"
        "global_load_dwordx4 %3, %11, off offset:16\n\t"
        "global_load_dwordx4 %4, %11, off offset:128\n\t"
        "global_load_dwordx4 %5, %11, off offset:144\n\t"
        "global_load_dwordx4 %6, %12, off\n\t"
        "global_load_dwordx4 %7, %12, off offset:16\n\t"
        "global_load_dwordx4 %8, %12, off offset:128\n\t"
        "global_load_dwordx4 %9, %12, off offset:144"
        : "=&v"(qraw[0]), "=&v"(qraw[1]), "=&v"(ur[0]), "=&v"(ur[1]), "=&v"(ur[2]), "=&v"(ur[3]), "=&v"(vr[0]),
          "=&v"(vr[1]), "=&v"(vr[2]), "=&v"(vr[3])
        : "v"(qrow), "v"(pu), "v"(pv)
        : "memory");
  }
  stage_tile(0, 0);
  // the queries and the biases have landed (tile 0's five requests are newer).  ONE statement, no branch around it: the
  // "+v" ties keep the values in the registers the loads write - with the wait in two branches (a first version of the
  // two-tiles-ahead staging) hipcc merged the branches through register copies placed IN FRONT of the waits, i.e. copies of
  // registers whose loads were still in flight (NaNs in test_relpos_attention2[49]).
  // (5 = tile 0's requests of this wave: K, three position pieces, V^T - see stage_tile)
  asm volatile("s_waitcnt vmcnt(5)"
               : "+v"(qraw[0]), "+v"(qraw[1]), "+v"(ur[0]), "+v"(ur[1]), "+v"(ur[2]), "+v"(ur[3]), "+v"(vr[0]),
                 "+v"(vr[1]), "+v"(vr[2]), "+v"(vr[3])
               :
               : "memory");
  if (64 < klen) stage_tile(0, 1);  // staging runs two tiles ahead, see the key loop (tile 0 is waited for there)
  stamp();
  bf16x8 qu[2], qv[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const bf16x8 raw = __builtin_bit_cast(bf16x8, qraw[ks]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float q = (float)raw[e];
      // 1 / sqrt(64) goes into the operands: a power of two, so (AC + BD) / 8 comes out bit for bit the same and the
      // sixteen multiplications per lane and tile are gone
      qu[ks][e] = (bf16)((q + ur[2 * ks + (e >> 2)][e & 3]) * 0.125f);
      qv[ks][e] = (bf16)((q + vr[2 * ks + (e >> 2)][e & 3]) * 0.125f);
    }
  }

  f32x4 acc_o[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) acc_o[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float row_m = -INFINITY;
  // the softmax denominator rides on the matrix core: a fifth "V^T fragment" of ones makes O^T's extra rows the sum of the
  // (bf16-rounded) probabilities over ALL the keys of a query - no per-element accumulation, no cross-lane reduction
  f32x4 acc_l = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bf16x8 ones = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
  constexpr float LOG2E = 1.4426950408889634f;
  float* const bd = (float*)(smem + SBD_OFF) + wave * 16 * LDB;

  // (Round 5, measured and not kept: the S^T / window MFMAs of tile kt + 1 issued in front of tile kt's rel-shift, softmax and
  // P . V - two sets of score / window accumulators, 172 registers - is as fast as this order once the staging runs two tiles
  // ahead: 11.15 against 11.12 us, 40.2 against 39.8 us; profiles/r05g_attention_pipelined_ab.txt.)
  f32x4 scb[2][4], ddb[2][5];
  auto s_phase = [&](int kt, f32x4 (&sc)[4], f32x4 (&dd)[5]) __attribute__((always_inline)) {
    const int jl = kt * 64;
#pragma unroll
    for (int n = 0; n < 4; ++n) sc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 5; ++n) dd[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned char* sk = smem + SK_OFF + (jl + lr) * 128;
    const unsigned char* sp = smem + SP_OFF + (112 - 16 * wave + jl + lr) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + lg) ^ swz) << 4;
#pragma unroll
      for (int n = 0; n < 4; ++n) sc[n] = MM::mma(*(const bf16x8*)(sk + n * 2048 + coff), qu[ks], sc[n]);
#pragma unroll
      for (int n = 0; n < 5; ++n) dd[n] = MM::mma(*(const bf16x8*)(sp + n * 2048 + coff), qv[ks], dd[n]);
    }
  };
  auto rest_phase = [&](int kt, int j0, f32x4 (&sc)[4], f32x4 (&dd)[5]) __attribute__((always_inline)) {
    if (stamps) asm volatile("s_nop 0" : "+v"(sc[3]), "+v"(dd[4]));  // (developer timing: the MFMAs have finished)
    stamp();
    // ---- rel_shift: D[c][i] -> scratch[i][c], read back at c = 15 - i + j
#pragma unroll
    for (int n = 0; n < 5; ++n)
      *(float4*)(bd + lr * LDB + 16 * n + 4 * lg) = make_float4(dd[n][0], dd[n][1], dd[n][2], dd[n][3]);
    float tm = -INFINITY;
    const float* bdr = bd + lr * LDB + 15 - lr + 4 * lg;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[n][r] += bdr[16 * n + r];
    if (j0 + 64 > klen) {  // (uniform: only the tile that holds the utterance's end masks anything)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[n][r] = (j0 + 16 * n + 4 * lg + r < klen) ? sc[n][r] : -INFINITY;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) tm = fmaxf(tm, sc[n][r]);
    // ---- online softmax; this lane's query is iw0 + lr, its keys the 16 (n, r) of lane group lg
    if (stamps) asm volatile("s_nop 0" : "+v"(tm));
    stamp();
    tm = wave_xor16_max(tm);  // the four lane groups of a query: two register swaps (gfx950), no LDS round trip
    tm = wave_xor32_max(tm);
    const float mn = fmaxf(row_m, tm);
    const float alpha = __expf(row_m - mn);
    row_m = mn;
    const float mnl = mn * LOG2E;
    unsigned pbu[2][4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // exp(s - m) = 2^(s log2e - m log2e): one fma + v_exp_f32 (masked: 2^-inf = 0); two values per conversion
        const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[n][2 * h], LOG2E, -mnl));
        const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[n][2 * h + 1], LOG2E, -mnl));
        // (as a vector conversion: one v_cvt_pk_bf16_f32, and hipcc places the wait state a v_exp_f32 result needs
        // before a VALU read - from inline asm it does not, and the conversion read stale registers)
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        pbu[n >> 1][(n & 1) * 2 + h] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){e0, e1}, bf16x2));
      }
    bf16x8 pb[2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      pb[jp] = __builtin_bit_cast(bf16x8, (u32x4){pbu[jp][0], pbu[jp][1], pbu[jp][2], pbu[jp][3]});
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      acc_o[f][0] *= alpha; acc_o[f][1] *= alpha; acc_o[f][2] *= alpha; acc_o[f][3] *= alpha;
    }
    acc_l[0] *= alpha;  // (the other three rows of the ones fragment carry the same sum; only this one is read)
    if (stamps) asm volatile("s_nop 0" : "+v"(acc_o[3]));
    stamp();
    // ---- O^T += V^T . P^T   (V^T tile kt: [64 dk][128 B], chunk c holds keys 8 c .. 8 c + 7 of the tile)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      acc_l = MM::mma(ones, pb[jp], acc_l);
      const int c0 = 4 * jp + (lg >> 1);
      const int sw = (lr >> 1) & 7;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const unsigned char* sv = smem + SV_OFF + kt * 8192 + (16 * f + lr) * 128 + (lg & 1) * 8;
        const bf16x4 a0 = *(const bf16x4*)(sv + ((c0 ^ sw) << 4));
        const bf16x4 a1 = *(const bf16x4*)(sv + (((c0 + 2) ^ sw) << 4));
        const bf16x8 vf = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        acc_o[f] = MM::mma(vf, pb[jp], acc_o[f]);
      }
    }
  };
  // Staging runs TWO tiles ahead (round 5; one ahead until then): tile kt + 2's pieces are requested in front of tile kt's
  // chain and the wait in front of tile kt + 1's products counts them out (`vmcnt(3)`: a later tile is three requests per
  // wave - K, position rows, V^T - and requests complete in order).
  auto exists = [&](int js, int t) { return t < KSUP / 64 && js + 64 * t < klen; };
  // (The counts are exact only while nothing else of this wave is in the memory pipeline between a tile's requests and
  // its wait: stores count in vmcnt on gfx9, so the developer stamps - one store each - would let a wait pass before its
  // tile has landed.  With stamps on, wait for everything: timing of a diagnostic build, never a race.  ADVICE r05;
  // tools/isa_waits.py checks the product build for stray VMEM instructions between the staging asm and the waits.)
  constexpr int TILE_REQS = 3;  // K rows, position rows, V^T rows: one request per wave each for a tile after the first
  auto wait_but_newest_tile = [&](bool newer) __attribute__((always_inline)) {
    if (newer && !stamps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TILE_REQS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  for (int js = 0; js < klen; js += KSUP) {
    if (js > 0) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave is done with the previous super-tile
      stage_tile(js, 0);
      if (exists(js, 1)) stage_tile(js, 1);
    }
    // tile 0 of the super-tile has landed: this wave's own pieces, then everybody's (barrier).  (Requested all at once up
    // front, the tiles' pieces of different waves interleave in the memory pipeline and the first tile is complete only when
    // most of the 112 KiB are: 12.8 -> 12.2 us; tile by tile: see profiles/r03m.)
    wait_but_newest_tile(exists(js, 1));
    __builtin_amdgcn_s_barrier();
    stamp();
    if (exists(js, 2)) stage_tile(js, 2);
    s_phase(0, scb[0], ddb[0]);
#pragma unroll
    for (int kt = 0; kt < KSUP / 64; ++kt) {
      const int j0 = js + kt * 64;
      if (j0 >= klen) break;
      if (exists(js, kt + 1)) {  // (uniform) the next tile: landed -> its products are issued
        wait_but_newest_tile(exists(js, kt + 2));
        __builtin_amdgcn_s_barrier();
        stamp();
        if (exists(js, kt + 3)) stage_tile(js, kt + 3);
      }
      if (kt > 0) s_phase(kt, scb[kt & 1], ddb[kt & 1]);
      rest_phase(kt, j0, scb[kt & 1], ddb[kt & 1]);
    }
  }
  // (a workgroup that stopped early - klen short of the staged keys - still has requests in flight into ITS LDS)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (stamps) asm volatile("s_nop 0" : "+v"(acc_o[3]));
  stamp();

  // ---- normalise and store ctx[b*T + i][hh*64 + 16 f + 4 lg + r]
  const int i = iw0 + lr;
  if (i < T) {
    const float row_l = acc_l[0];
    const float inv = row_l > 0.f ? 1.0f / row_l : 0.f;
    bf16* o = ctx + ((size_t)b * T + i) * (H * 64) + hh * 64 + 4 * lg;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      bf16x4 pk = {(bf16)(acc_o[f][0] * inv), (bf16)(acc_o[f][1] * inv), (bf16)(acc_o[f][2] * inv),
                   (bf16)(acc_o[f][3] * inv)};
      *(bf16x4*)(o + 16 * f) = pk;
    }
  }
}


}  // namespace

extern "C" int em_relpos_attention2_bf16(const void* qh, const void* kh, const void* vt, const void* p,
                                         int32_t ldp, const float* pos_u, const float* pos_v,
                                         const int32_t* klens, int32_t B, int32_t T, int32_t Tpad, int32_t h,
                                         void* ctx, void* stream) {
  if (!qh || !kh || !vt || !p || !pos_u || !pos_v || !klens || !ctx) return EM_ERR_BAD_ARG;
  if (B <= 0 || T <= 0 || h <= 0) return EM_ERR_BAD_ARG;
  if (Tpad % KSUP != 0 || Tpad < T || ldp % 8 != 0) return EM_ERR_UNSUPPORTED;
  static EmLdsCap cap = {};
  if (em_raise_lds_cap((const void*)relpos_attn2_kernel, SMEM_BYTES, &cap) != EM_OK) return EM_ERR_LAUNCH;
  dim3 grid(em_cdiv(T, QB), h, B);
  static long long* stamps = nullptr;
  const bool want_stamps = em_sw().attn2_stamps;
  if (want_stamps && !stamps && hipMalloc((void**)&stamps, 32 * sizeof(long long)) != hipSuccess) return EM_ERR_LAUNCH;
  if (want_stamps && hipMemsetAsync(stamps, 0, 32 * sizeof(long long), (hipStream_t)stream) != hipSuccess) return EM_ERR_LAUNCH;
  const bool rec = em_prof_begin(stream);
  hipLaunchKernelGGL(relpos_attn2_kernel, grid, dim3(512), SMEM_BYTES, (hipStream_t)stream, (const bf16*)qh,
                     (const bf16*)kh, (const bf16*)vt, (const bf16*)p, ldp, pos_u, pos_v, klens, T, Tpad, h,
                     (bf16*)ctx, want_stamps ? stamps : nullptr);
  if (want_stamps) {
    long long hs[32];
    if (hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost) == hipSuccess) {
      printf("[attn2 stamps, cycles since entry]");
      for (int i = 1; i < 32 && hs[i]; ++i) printf(" %lld", hs[i] - hs[0]);
      printf("\n");
      fflush(stdout);
    }
  }
  if (rec) em_prof_end(stream, 6.0 * B * h * (double)T * T * 64, EM_PROF_ATTN);
  EM_CHECK_LAUNCH();
  return EM_OK;
}
