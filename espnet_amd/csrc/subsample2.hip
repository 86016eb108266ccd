// Conv2dSubsampling's two convolutions in ONE kernel (bf16, d = 256, 3 x 3 stride 2 twice):
//     feats [B][T_f][n_mels] f32  --conv1 + ReLU-->  (never in memory)  --conv2 + ReLU-->  c2 [B][T2][F2][256] bf16
// Reference: Conv2dSubsampling.forward, espnet2/legacy/nets/pytorch_backend/transformer/subsampling.py:432-447
// (torch.nn.Conv2d(1, d, 3, 2), ReLU, Conv2d(d, d, 3, 2), ReLU), utterance MVN (layers/utterance_mvn.py:45-88) folded
// into the input rows as in csrc/frontend.hip.
//
// Why.  Unfused, conv1 writes a [B][500][39][256] bf16 map (319 MB at B = 32: 72 us at HBM speed) that conv2's
// implicit GEMM reads back twice (two 128-column tiles), 211 us at 0.34 of the bf16 MFMA peak: together 22 % of the
// greedy step (profiles/r03e_bench_small_b32_kernel_stats.csv).  conv1 is 9 MACs per map element - cheap enough to
// recompute next to conv2's MFMAs, so the map need not exist.
//
// Shape of the kernel.  One workgroup = 8 output time rows x all F2 frequency bins (152 positions at 80 mels, padded
// to 10 MFMA fragments of 16) x all 256 output channels; four waves (one per SIMD: the 160 accumulator registers of
// a wave plus its weight ring need more than the 256 registers two waves per SIMD would leave), each in two roles:
//   * conv2: wave w owns output channels 64 w .. 64 w + 63 (4 fragments) x all 10 position fragments.  K is walked in
//     8 chunks of 32 input channels x 9 taps, a tap being ONE 32-deep MFMA step: the tap's four weight fragments come
//     straight from global memory (fragment-major packing by the host, 1 KiB per wave-wide load, requested two taps
//     ahead through a register ring), the position fragments from an LDS "patch" [map position][32 ch] of the conv1
//     output (80-byte rows: conflict-free 16-byte reads at the stride-2 positions of a fragment) at (2 t2 + kt,
//     2 f2 + kf): the im2col is index arithmetic on LDS addresses.  The weights are the MFMA A operand (C^T[n][m]), and
//     the host packs their rows so that row lr of a wave's fragment j is output channel 64 w + 16 (lr / 4) + 4 j + lr % 4:
//     a lane ends up with 16 consecutive output channels of one position, i.e. two 16-byte stores into the channel-last c2.
//   * conv1, ALSO on the matrix cores: for 16 map positions x 16 channels it is a [16 x 9] x [9 x 16] product, and
//     with operands split into bf16 high and low parts (x = xh + xl, w = wh + wl) one v_mfma_f32_16x16x32_bf16
//     computes  xh.wh + xl.wh + xh.wl + bias  in f32 (k = 0-8 | 9-17 | 18-26 | 27, 28 for the two halves of the
//     bias): 16+ significant bits per operand, i.e. f32-class conv1 results (error ~1e-5 of the terms, against the
//     4e-3 of the bf16 rounding they get anyway).  The position operands (the tile's inputs in that im2col form,
//     MVN applied in f32 first) are built ONCE per tile into LDS - they do not depend on the channel - and a chunk's
//     conv1 is then 22 MFMAs per wave against the 360 of its conv2.  (A VALU conv1 - 9 packed FMAs per position and
//     channel pair with the weights in scalar registers - spilled 1 090 SGPRs and cost as many issue slots as the
//     MFMAs; tools/isa_waits.py.)
// One barrier per chunk: the conv1 patch of chunk c + 1 is produced, a fragment per half tap, while chunk c is consumed.
// Per workgroup and chunk: 382 MFMAs per wave (6 100 cycles of matrix-core time), 144 KiB of weights (64 B/clk would
// take 2 300 cycles), 400 KiB of LDS fragment reads (1 600 - 3 100 cycles): the kernel is built to be MFMA-bound, and
// measures 7 400 cycles per chunk (EM_SUB2_STAMPS; 8 850 before everything that is not a conv2 MFMA was dealt out one
// piece per MFMA slot, see the chunk loop).  The workgroups are persistent: tile loop, next tile's inputs and operand
// build overlapped with the current tile's last chunk and stores (see the kernel body); DESIGN.md section 4c.
#include <stdio.h>
#include <stdlib.h>

#include "em_common.h"
#include "switches.h"

namespace {

constexpr int D = 256;           // OUTPUT channels of one launch; wider models run one launch per 256-channel half (round 4)
constexpr int TT = 8;          // output time rows per workgroup
constexpr int MF = 10;         // position fragments of conv2: TT * F2 <= 160
constexpr int CC = 32;         // input channels per chunk = one MFMA k-step per tap
// (chunks of 32 INPUT channels: a template argument of the kernel - 8 for d = 256, 16 for d = 512)
constexpr int T1R = 2 * TT + 1;   // conv1 rows a tile reaches: 17
constexpr int INR = 2 * T1R + 1;  // feature rows: 35
constexpr int F1MAX = 40, F2MAX = 19, MELMAX = 2 * F1MAX + 2;
constexpr int NPF = (T1R * F1MAX + 15) / 16;  // map-position fragments of conv1: 43
constexpr int PSTRIDE = 2 * CC + 16;          // bytes per patch position: 64 of channels + 16 of padding
// A patch is [T1R map rows][F1 positions][PSTRIDE] with a row pitch of ROWP bytes, chosen so that the step from the last
// position of an output row's window to the first of the next one (2 ROWP - (F2 - 1) * 2 * PSTRIDE) is the same number of
// LDS banks as the step between neighbouring positions (2 * PSTRIDE = 40 banks) for F2 = 19, the 80-mel case: a 16-position
// fragment that wraps over a row end then reads as conflict-free as one inside a row.  With the dense pitch F1 * PSTRIDE
// the wrap shifted the rest of the fragment by 32 banks, onto the banks of its first lanes: SQ_LDS_BANK_CONFLICT was
// 39 % of SQ_LDS_IDX_ACTIVE (profiles/r03ah_greedy_pmc_sq.json), seven of the ten fragments of a tile wrap.
constexpr int ROWP = F1MAX * PSTRIDE + 112;      // 3 312: >= F1 * PSTRIDE, = 19 * 80 (mod 128)
static_assert(ROWP % 16 == 0 && (2 * ROWP - 18 * 2 * PSTRIDE) % 256 == (2 * PSTRIDE) % 256, "bank-neutral row wrap at F2 = 19");
constexpr int PATCH_BYTES = 55 * 1024;           // >= T1R * ROWP = 56 304, a multiple of 1 KiB
constexpr int XIN_OFF = 0;                       // [NPF][64 lanes][16 B]: the conv1 position operands
constexpr int PATCH_OFF = NPF * 1024;
constexpr int IN_OFF = PATCH_OFF;  // f32 [INR][n_mels] (contiguous rows: filled by LDS-DMA), aliasing patch 0: free during a tile's LAST chunk (which reads patch 1 and produces nothing)
constexpr int MAXK = 16;                              // tiles per persistent workgroup at most (the host sizes the grid)
constexpr int MEAN_OFF = PATCH_OFF + 2 * PATCH_BYTES;  // f32 [MAXK][MELMAX]: per-bin means of the utterances of this workgroup's tiles
constexpr int BIAS_OFF = MEAN_OFF + MAXK * MELMAX * 4;  // f32 [D]: conv2's bias (read in the epilogue without touching vmcnt)
constexpr int SMEM_BYTES = BIAS_OFF + D * 4;
static_assert(INR * MELMAX * 4 + 1024 <= PATCH_BYTES && SMEM_BYTES <= 160 * 1024, "LDS");
template <bool V> struct Flag { static constexpr bool value = V; };
constexpr int PF_PER_WAVE = (NPF + 3) / 4;  // 11
static_assert(PF_PER_WAVE <= 17, "a conv1 fragment per group of a chunk at most, finished in the group after");

typedef const __attribute__((address_space(1))) bf16x8* GFRAG;

// LDS-DMA from inline asm (see csrc/block.hip: hidden from hipcc's waitcnt pass on purpose): 64 lanes x 16 bytes from
// sbase + voff to LDS at lds_dst + 16 * lane
__device__ __forceinline__ void glds16(const unsigned char* sbase, unsigned voff, unsigned lds_dst) {
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

struct Sub2Args {
  const float* feats;     // [B][T_f][n_mels]
  const float* partial;   // [B][8][n_mels] MVN partial sums, or NULL
  const int32_t* flens;   // [B]
  const void* w1f;        // conv1 weights + bias as MFMA operands, [8 chunks][2 fragments][64 lanes][8] bf16
  const void* w2f;        // conv2 weights, fragment-major [chunk][tap][wave][frag][lane][8] bf16
  const float* b2;        // [256]
  void* out;              // [B][T2][F2][ldo] bf16, already offset to this launch's first output channel
  int ldo;                // channels per output position (= d of the model)
  int B, T_f, n_mels, F1, T2, F2;
  long long* stamps;      // developer timing (EM_SUB2_STAMPS): cycle stamps of workgroup (3, 5), wave 0
};

// conv2's MFMA with the accumulator pinned to the AGPR half of the register file.  Through the builtin hipcc kept a
// third of the 160 accumulator registers in VGPRs and copied them in and out around the MFMAs (four v_accvgpr_write
// plus hazard nops per MFMA, as much issue time as the MFMA itself: a chunk took 9 000 cycles for 5 800 of MFMA).
// Nothing reads an accumulator within hundreds of cycles of its last MFMA here (a barrier and an LDS wait lie
// before the epilogue), so the hazard nops hipcc cannot insert for asm are not needed.
__device__ __forceinline__ void mfma_acc(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ bf16 hi_of(float x) { return (bf16)x; }
__device__ __forceinline__ bf16 lo_of(float x) { return (bf16)(x - (float)(bf16)x); }

template <int NCHUNK>
__global__ __launch_bounds__(256, 1) void sub2_kernel(const Sub2Args a) {
  static_assert(NCHUNK % 2 == 0, "the last chunk must read patch 1");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  float* const s_in = (float*)(smem + IN_OFF);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int n_mels = a.n_mels, F1 = a.F1, F2 = a.F2;
  // Persistent workgroups (one per CU): workgroup g walks the tiles g, g + gridDim.x, ...  With one workgroup per
  // tile (round 3's first version, EM_SUB2_STAMPS) every tile paid 13 K cycles of input latency + operand building
  // before its first conv2 MFMA and 10 K cycles to get its stores accepted at the end - all 256 CUs store their 80 KiB
  // at the same moment, four times per launch - of 100 K.  Here the NEXT tile's 35 feature rows are requested by
  // LDS-DMA at the start of a tile's last chunk (into patch 0, which that chunk neither reads nor produces), and its
  // conv1 operands are built BETWEEN the output stores of the epilogue: LDS and VALU work that needs no memory, done
  // while the stores wait for the fabric.  (Prefetching the rows into registers spilled: the kernel lives at 255 VGPRs.)
  const int tiles_t = (a.T2 + TT - 1) / TT, ntiles = tiles_t * a.B;
  int tile = blockIdx.x;
  int nts = 0;
  auto stamp = [&]() {
    if (a.stamps && blockIdx.x == 7 && tid == 0 && nts < 32) a.stamps[nts] = (long long)__builtin_amdgcn_s_memtime();
    ++nts;
  };
  stamp();
  // weight fragments of conv2: tap g = cc * 9 + tap of the whole K walk, four 1 KiB lines per wave and tap, requested
  // two taps (1 280 cycles of MFMA) ahead through a ring of three sets - 9 taps per chunk, so the slot g % 3 = tap % 3
  // is a compile-time index.  Unconditional: past the end the walk wraps around (the next tile's first taps).
  // (through a buffer resource: the tap's offset is a scalar register, the fragment's an immediate - no per-load 64-bit
  // vector address arithmetic in a loop whose issue slots are what it runs out of, see below)
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w2f), 0, NCHUNK * 9 * 16384, 0x00020000);
  const unsigned wvoff = (unsigned)(wave * 4096 + lane * 16);
  constexpr int NTAP = NCHUNK * 9;
  static_assert(NTAP % 3 == 0, "the ring slot of a tap must not depend on the tile");
  bf16x8 wr[3][4];
  auto wload = [&](int g, bf16x8 (&w)[4]) {
    const unsigned so = (unsigned)(g < NTAP ? g : g - NTAP) * 16384u;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      w[j] = __builtin_bit_cast(bf16x8, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + j * 1024, so, 0));
  };
  // ---- a tile's inputs.  Rows past the utterance's end repeat the last row (they only reach output rows past T2,
  // which are never stored).  The rows are contiguous in memory and in LDS: 1 KiB per wave-instruction.
  const int rowb = n_mels * 4, tileb = INR * rowb, npiece = (tileb + 1023) >> 10;
  auto request_inputs = [&](int tl) {
    const int bb = tl / tiles_t, tt0 = (tl - bb * tiles_t) * TT;
    if (4 * tt0 + INR <= a.T_f) {
      // the tile's rows are one contiguous block of the utterance: base in scalar registers, lane * 16 the only VGPR
      // (at this point of the kernel there are no free ones: computed per lane, the addresses were spilled and every
      // request waited for a scratch reload - and with it for the request before it)
      const unsigned char* fb = uniform_ptr((const unsigned char*)(a.feats + ((size_t)bb * a.T_f + 4 * tt0) * n_mels));
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        int j = wave + 4 * i;
        j = j < npiece ? j : npiece - 1;  // (surplus requests repeat the last piece: same bytes, no branch)
        int jb = j * 1024;
        jb = jb + 1024 <= tileb ? jb : tileb - 1024;  // (the last piece is shifted back inside the tile: same bytes twice)
        glds16(fb + jb, (unsigned)(lane * 16), IN_OFF + jb);
      }
    } else {  // the utterance's last tile: rows past its end repeat the last row (one tile in 32)
      const unsigned char* fb = uniform_ptr((const unsigned char*)(a.feats + (size_t)bb * a.T_f * n_mels));
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        int j = wave + 4 * i;
        j = j < npiece ? j : npiece - 1;
        int jb = j * 1024;
        jb = jb + 1024 <= tileb ? jb : tileb - 1024;
        const int o = jb + lane * 16;
        const int r = o / rowb, col = o - r * rowb;
        int row = 4 * tt0 + r;
        row = row < a.T_f ? row : a.T_f - 1;
        glds16(fb, (unsigned)(row * rowb + col), IN_OFF + jb);
      }
    }
  };
  static_assert(3 * 4 * 1024 >= INR * MELMAX * 4, "three pieces per wave cover a tile");
  request_inputs(tile);
  // the per-bin means (MVN) of the utterances of ALL this workgroup's tiles, once, while registers are plentiful
  float* const s_means = (float*)(smem + MEAN_OFF);
  float* const s_bias = (float*)(smem + BIAS_OFF);
  s_bias[tid] = a.b2[tid];  // (256 threads = D channels)
  {
    // (tile, bin) pairs dealt over all 256 threads: one or two round trips for the usual four tiles per workgroup
    int kn = (ntiles - tile + (int)gridDim.x - 1) / (int)gridDim.x;
    kn = kn < MAXK ? kn : MAXK;
    for (int item = tid; item < kn * n_mels; item += 256) {
      const int k = item / n_mels, bin = item - k * n_mels;
      const int bb = (tile + k * (int)gridDim.x) / tiles_t;
      float mean = 0.f;
      if (a.partial) {
        const float* pp = a.partial + (size_t)bb * 8 * n_mels + bin;
        float pv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) pv[q] = pp[q * n_mels];
        float sm = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) sm += pv[q];
        mean = sm / (float)a.flens[bb];
      }
      s_means[k * MELMAX + bin] = mean;
    }
  }
  const int npos = T1R * F1;
  const int M = TT * F2;
  // ---- conv1's position operands: fragment pf covers map positions 16 pf .. 16 pf + 15 (p = t1l * F1 + f1); lane
  // (lr, lg) holds k = 8 lg .. 8 lg + 7 of position 16 pf + lr in the split layout of the header:
  //   k: 0-8 xh[tap] | 9-17 xl[tap] | 18-26 xh[tap] | 27, 28: 1 | 29-31: 0
  // MVN: the utterance's per-bin mean comes off in f32 first.
  const int f1_inv = (65536 + F1 - 1) / F1;
  auto build_xin = [&](int pf, const float* s_mean) {
    int p = pf * 16 + lr;
    p = p < npos ? p : npos - 1;
    const int t1l = (p * f1_inv) >> 16, f1 = p - t1l * F1;  // (exact: p < 688, F1 <= 40)
    // (columns 2 f1, 2 f1 + 1 as one 8-byte read - n_mels is a multiple of 4 - and 2 f1 + 2 as a third dword)
    const float2 mu01 = *(const float2*)(s_mean + 2 * f1);
    const float mu[3] = {mu01.x, mu01.y, s_mean[2 * f1 + 2]};
    float x[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float* row = s_in + (2 * t1l + i) * n_mels + 2 * f1;
      const float2 x01 = *(const float2*)row;
      x[i * 3 + 0] = x01.x - mu[0];
      x[i * 3 + 1] = x01.y - mu[1];
      x[i * 3 + 2] = row[2] - mu[2];
    }
    bf16 h[9], l[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      h[t] = hi_of(x[t]);
      l[t] = lo_of(x[t]);
    }
    const bf16 one = (bf16)1.0f, zero = (bf16)0.0f;
    bf16x8 v;
    if (lg == 0) v = (bf16x8){h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};
    else if (lg == 1) v = (bf16x8){h[8], l[0], l[1], l[2], l[3], l[4], l[5], l[6]};
    else if (lg == 2) v = (bf16x8){l[7], l[8], h[0], h[1], h[2], h[3], h[4], h[5]};
    else v = (bf16x8){h[6], h[7], h[8], one, one, zero, zero, zero};
    *(bf16x8*)(smem + XIN_OFF + pf * 1024 + lane * 16) = v;
  };
  // ---- conv1 of a chunk for this wave's j-th position fragment (pf = wave + 4 j), both 16-channel fragments of the
  // chunk: 2 MFMAs; the lane gets channels 4 lg .. 4 lg + 3 (+ 16 for the second fragment) of position 16 pf + lr
  bf16x8 w1[2], w1b[2];
  auto load_w1_into = [&](int cc, bf16x8 (&w)[2]) {
#pragma unroll
    for (int f = 0; f < 2; ++f) w[f] = *(GFRAG)((const unsigned char*)a.w1f + (size_t)(cc * 2 + f) * 1024 + lane * 16);
  };
  auto load_w1 = [&](int cc) { load_w1_into(cc, w1); };
  auto conv1_pf = [&](int j) {
    const int pf = wave + 4 * j;
    return pf < NPF ? pf : NPF - 1;  // (the last waves redo a fragment: same bytes, no branch)
  };
  auto conv1_read = [&](int j) { return *(const bf16x8*)(smem + XIN_OFF + conv1_pf(j) * 1024 + lane * 16); };
  // Inside the chunk loop conv1 is split over two groups: the two MFMAs of a fragment are issued in one group (asm,
  // results in VGPRs), their ReLU / bf16 / LDS write in the NEXT one - with one wave per SIMD an MFMA hides three other
  // instructions at most, and waiting for a just-issued MFMA's result stalls the twenty behind it.  (No hazard nops
  // needed: 20 MFMAs lie between the write and the read.)
  // ReLU of two values AFTER the conversion, on the packed pair: as 16-bit integers negative bf16 values (sign bit
  // set, -0 included) are negative, so max(., 0) is the ReLU - one v_pk_max_i16 for what fmaxf does in four v_max_f32
  // (each fmaxf brings a canonicalising second one)
  auto relu_pack = [](float x, float y) {
    // (asm: from `(bf16)x, (bf16)y` hipcc makes two conversions and a v_perm, and as plain arithmetic the pieces drift out
    // of the slots they are dealt into below)
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_pk_max_i16 %0, %0, 0" : "=v"(r) : "v"(x), "v"(y));
    return r;
  };
  // byte offset inside a patch of this lane's map position of conv1 fragment pf (position 16 pf + lr, clamped like the
  // operand build: the lanes past the map repeat its last position - same value, same address)
  auto patch_pos = [&](int pf) {
    int p = pf * 16 + lr;
    p = p < npos ? p : npos - 1;
    const int t1l = (p * f1_inv) >> 16;
    return t1l * ROWP + (p - t1l * F1) * PSTRIDE;
  };
  auto conv1_frag = [&](int buf, int j, const bf16x8& xv) {
    unsigned char* const dst = smem + PATCH_OFF + buf * PATCH_BYTES + patch_pos(conv1_pf(j)) + lg * 8;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[f], xv, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const bf16x4 pk = {(bf16)fmaxf(c[0], 0.f), (bf16)fmaxf(c[1], 0.f), (bf16)fmaxf(c[2], 0.f), (bf16)fmaxf(c[3], 0.f)};
      *(bf16x4*)(dst + f * 32) = pk;
    }
  };
  // "use" of everything requested ahead for the next tile - conv1 weights of chunks 0 and 1, conv2 weights of taps 0 and
  // 1: after it hipcc has nothing to wait for at the top of the tile loop, where a wait would be for the forty stores
  // of the tile before (memory operations complete in issue order)
  auto settle = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(w1[0]), "+v"(w1[1]), "+v"(w1b[0]), "+v"(w1b[1]));
    asm volatile("" : "+v"(wr[0][0]), "+v"(wr[0][1]), "+v"(wr[0][2]), "+v"(wr[0][3]), "+v"(wr[1][0]), "+v"(wr[1][1]),
                 "+v"(wr[1][2]), "+v"(wr[1][3]));
  };
  // ---- the first tile: rows landed, operands built
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // every wave's pieces have landed; the means and the bias are in LDS
  wload(0, wr[0]);
  wload(1, wr[1]);
  load_w1(0);
  load_w1_into(1, w1b);
#pragma unroll 1
  for (int j = 0; j < PF_PER_WAVE; ++j) build_xin(conv1_pf(j), s_means);
  settle();
  int kt = 0;  // index of the tile among this workgroup's
  // conv2: byte offset of this lane's column of position fragment i inside a patch, at tap (0, 0): position
  // (2 t2l, 2 f2), channels 8 lg ..; positions past the tile repeat the last one (computed, never stored)
  int pb[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    int m = i * 16 + lr;
    m = m < M ? m : M - 1;
    const int t2l = m / F2, f2 = m - t2l * F2;
    pb[i] = (2 * t2l) * ROWP + (2 * f2) * PSTRIDE + lg * 16;
  }
  const size_t out_bytes = ((size_t)a.B * a.T2 * F2 * a.ldo - (a.ldo - D)) * 2;  // (from this launch's first channel to the end of the last position's 256)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)(unsigned)out_bytes, 0x00020000);

  for (;;) {
  const int b = tile / tiles_t, t2_0 = (tile - b * tiles_t) * TT;
  const int next = tile + gridDim.x;
  __syncthreads();  // the operands are complete (and the rows they were built from dead: patch 0 may be written)
  stamp();
  // (rolled: unrolled, the eleven patch addresses are tile invariants - hoisted, then spilled; two fragments per trip so
  // that one's LDS read -> MFMA -> convert -> LDS write chain runs in the shadow of the other's)
#pragma unroll 1
  for (int j = 0; j < PF_PER_WAVE; j += 2) {
    const bf16x8 x0 = conv1_read(j), x1 = conv1_read(j + 1);  // (j + 1 past the end: conv1_pf clamps, a harmless repeat)
    conv1_frag(0, j, x0);
    conv1_frag(0, j + 1, x1);
  }
  // chunk 1's conv1 weights were requested BEFORE the previous tile's stores (memory operations complete in issue
  // order: a load issued here would be a wait for all forty of them)
#pragma unroll
  for (int f = 0; f < 2; ++f) w1[f] = w1b[f];
  f32x4 acc[4][MF];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();  // patch 0 is complete
  stamp();
  // The position fragments are read HALF A TAP ahead: a group = five fragments = 20 MFMAs (320 cycles of matrix-core
  // time, more than an LDS round trip), requested into the other half of a register double buffer before the current
  // group's MFMAs are issued.  Read next to their use - what hipcc does with the plain loop - every fragment exposed
  // its LDS latency to four MFMAs' worth of work.  18 groups per chunk: the buffer parity of a group is a compile-time
  // index.  (A whole tap ahead, with two chunks per trip for the parity, ran out of registers: the xin / patch
  // addresses of the conv1 slices were spilled, and a scratch reload is a vmcnt(0) wait in front of the weight ring.)
  constexpr int GF = MF / 2;  // fragments per group
  static_assert(MF % 2 == 0, "two groups per tap");
  bf16x8 bfr[2][GF];
  auto read_group = [&](const unsigned char* patch, int g, bf16x8 (&dst)[GF]) {
    const int tap = g >> 1, i0 = (g & 1) * GF;
    const int toff = (tap / 3) * ROWP + (tap % 3) * PSTRIDE;
#pragma unroll
    for (int i = 0; i < GF; ++i) dst[i] = *(const bf16x8*)(patch + pb[i0 + i] + toff);
  };
  read_group(smem + PATCH_OFF, 0, bfr[0]);
  bf16x8 xv[2];
  xv[0] = conv1_read(0);
  // a chunk: 9 taps x 2 groups; all but the last also produce the next chunk's patch, a conv1 fragment per group
  // With one wave per SIMD nothing else issues while this wave waits, and an MFMA (16 cycles of the matrix core) hides
  // three other instructions at most - so everything that is not a conv2 MFMA is dealt out BETWEEN them, one small piece
  // per MFMA, in exactly this order (sched_barrier after every slot; left to hipcc the pieces end up in one block
  // between the groups, ~120 cycles per group during which the matrix core has nothing to do):
  //   slots 0-4    the next group's position fragments (LDS)          slots 10-13  the previous conv1 fragment: ReLU, bf16
  //   slot  5      the next conv1 fragment's position operand (LDS)   slots 14-15  ... its patch address, its LDS write
  //   slots 6-9    the weights two taps ahead (even groups)           slots 16-17  this group's two conv1 MFMAs
  auto chunk = [&](int cc, auto last) {
    constexpr bool LAST = decltype(last)::value;
    const unsigned char* const patch = smem + PATCH_OFF + (LAST ? 1 : (cc & 1)) * PATCH_BYTES;
    const int nbuf = (cc + 1) & 1;
    f32x4 c1[2];
    unsigned pk1[2][2];
    int wdst = 0;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#pragma unroll
    for (int g = 0; g < 18; ++g) {
      const int tap = g >> 1, i0 = (g & 1) * GF;
      const bool fin = !LAST && g >= 1 && g - 1 < PF_PER_WAVE;  // a conv1 fragment issued in the previous group
      const bool iss = !LAST && g < PF_PER_WAVE;
#pragma unroll
      for (int k = 0; k < 4 * GF; ++k) {
        const int i = k >> 2, j = k & 3;
        mfma_acc(acc[j][i0 + i], wr[tap % 3][j], bfr[g & 1][i]);
        if (k < GF && g < 17) {
          const int g1 = g + 1, tap1 = g1 >> 1;
          bfr[g1 & 1][k] = *(const bf16x8*)(patch + pb[(g1 & 1) * GF + k] + (tap1 / 3) * ROWP + (tap1 % 3) * PSTRIDE);
        }
        if (k == 5 && !LAST && g + 1 < PF_PER_WAVE) xv[(g + 1) & 1] = conv1_read(g + 1);
        if (k >= 6 && k < 10 && !(g & 1)) {
          const int gt = cc * 9 + tap + 2;  // (past the end the walk wraps: the next tile's first taps)
          const unsigned so = (unsigned)(gt < NTAP ? gt : gt - NTAP) * 16384u;
          wr[(tap + 2) % 3][k - 6] = __builtin_bit_cast(bf16x8, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + (k - 6) * 1024, so, 0));
        }
        if (fin && k >= 10 && k < 14) {
          const int f = (k - 10) >> 1, h = (k - 10) & 1;
          pk1[f][h] = relu_pack(c1[f][2 * h], c1[f][2 * h + 1]);
        }
        if (fin && k == 14) wdst = PATCH_OFF + nbuf * PATCH_BYTES + patch_pos(conv1_pf(g - 1)) + lg * 8;  // (its address: a slot of its own)
        if (fin && k == 15) {
#pragma unroll
          for (int f = 0; f < 2; ++f) *(u32x2*)(smem + wdst + f * 32) = (u32x2){pk1[f][0], pk1[f][1]};
        }
        if (iss && (k == 16 || k == 17))
          // (early-clobber: an MFMA's destination must not overlap its A / B operands, and xv is dead after slot 17)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c1[k - 16]) : "v"(w1[k - 16]), "v"(xv[g & 1]));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
#pragma unroll 1
  for (int cc = 0; cc < NCHUNK - 1; ++cc) {
    chunk(cc, Flag<false>{});
    load_w1(cc + 2 < NCHUNK ? cc + 2 : 0);  // conv1 weights of the chunk produced during the next trip (at the end: the next tile's chunk 0)
    stamp();
    __syncthreads();
    stamp();
    // the next chunk's first group, from the patch that has just been completed
    read_group(smem + PATCH_OFF + ((cc + 1) & 1) * PATCH_BYTES, 0, bfr[0]);
    xv[0] = conv1_read(0);
  }
  // ---- the last chunk; the next tile's rows are requested first (patch 0 is dead: every wave is past the barrier)
  request_inputs(next < ntiles ? next : tile);
  {
    // (the chunk index through an opaque copy: as a constant, the 36 weight addresses of this chunk are invariants of the
    // tile loop, which hipcc hoists out of it - and spills: 72 registers the kernel does not have)
    int cc_last = NCHUNK - 1;
    asm volatile("" : "+s"(cc_last));
    chunk(cc_last, Flag<true>{});
  }
  stamp();
  // ---- epilogue: + bias, ReLU, bf16; lane = 16 consecutive channels n = 64 w + 16 lg + 4 j + r of position 16 i + lr
  // (the host packs the weights' rows in that order): two 16-byte stores per fragment row.
  float4 bias[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bias[j] = *(const float4*)(s_bias + wave * 64 + lg * 16 + j * 4);
  // (from LDS: a global load here would be counted by hipcc in front of the stores, with the asm requests it does
  // not see in between)
  load_w1_into(1, w1b);
  // this wave's pieces of the next tile's rows (requested a chunk ago)
  // (... and the conv1 weights of the next tile's chunks 0 and 1: "used" here, so that hipcc has nothing left to wait
  // for at the top of the tile loop, where its wait would be for the stores below)
  settle();
  __syncthreads();                                   // ... and every other wave's
  stamp();
  // position m = t2l * F2 + f2 of the tile IS its row offset in the output, so a lane's byte offsets are linear in the
  // fragment index (one VGPR + constants; per-fragment quotients would be hoisted out of the tile loop and spilled:
  // a scratch reload in front of every group of stores, each one a wait for all the stores before it)
  int mlim = (a.T2 - t2_0) * F2;
  mlim = mlim < M ? mlim : M;
  const unsigned base0 = (unsigned)(((((size_t)b * a.T2 + t2_0) * F2 + lr) * a.ldo + wave * 64 + lg * 16) * 2);
  ++kt;
  const float* const s_mean_next = s_means + (next < ntiles ? kt : kt - 1) * MELMAX;
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    const bool ok = i * 16 + lr < mlim;
    const unsigned off = ok ? base0 + (unsigned)(i * 16 * a.ldo * 2) : 0xffffffffu;
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // fragments 2 h, 2 h + 1: channels 8 h .. 8 h + 7 of the lane's sixteen
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
      u32x4 pk;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * h + jj;
        pk[2 * jj + 0] = relu_pack(acc[j][i][0] + bias[j].x, acc[j][i][1] + bias[j].y);
        pk[2 * jj + 1] = relu_pack(acc[j][i][2] + bias[j].z, acc[j][i][3] + bias[j].w);
      }
      __builtin_amdgcn_raw_buffer_store_b128(pk, rs, ok ? off + h * 16 : 0xffffffffu, 0, 0);
    }
    // a fragment of the next tile's conv1 operands between the stores (after the last tile: the same tile's again)
    // (the fragment index through an opaque copy, for the same reason: its quotients and addresses are tile invariants)
    int pf = conv1_pf(i);
    asm volatile("" : "+s"(pf));
    build_xin(pf, s_mean_next);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll 1
  for (int j = MF; j < PF_PER_WAVE; ++j) build_xin(conv1_pf(j), s_mean_next);
  stamp();
  if (next >= ntiles) break;
  tile = next;
  }
}

}  // namespace

// feats -> c2 (the input of the embedding Linear), see the header.  EM_ERR_UNSUPPORTED for shapes outside the kernel
// (the caller then runs conv1 and the implicit-GEMM conv2 as separate launches).
extern "C" int em_conv2d_sub12_bf16(const float* feats, const float* mvn_partial, const int32_t* flens, int32_t B,
                                    int32_t T_f, int32_t n_mels, const void* conv1_wf, const void* conv2_wf,
                                    const float* conv2_b, int32_t d, void* c2, void* stream) {
  if (!feats || !flens || !conv1_wf || !conv2_wf || !conv2_b || !c2 || B <= 0) return EM_ERR_BAD_ARG;
  if (T_f < 7) return EM_ERR_TOO_SHORT;
  const int T1 = (T_f - 3) / 2 + 1, F1 = (n_mels - 3) / 2 + 1;
  const int T2 = (T1 - 3) / 2 + 1, F2 = (F1 - 3) / 2 + 1;
  if (n_mels % 4 != 0) return EM_ERR_UNSUPPORTED;  // 16-byte LDS-DMA pieces of whole rows
  if ((d != D && d != 2 * D) || n_mels < 7 || n_mels > MELMAX || F1 > F1MAX || F2 > F2MAX || TT * F2 > 16 * MF) return EM_ERR_UNSUPPORTED;
  if ((size_t)B * T2 * F2 * d * 2 >= ((size_t)1 << 32) - 64) return EM_ERR_UNSUPPORTED;  // one buffer resource
  const int nhalf = d / D;  // d = 512 (round 4): one launch per 256 output channels, 16 chunks of input channels each;
                            // conv1 is recomputed by both (22 of a chunk's 382 MFMAs per wave)
  static EmLdsCap cap8 = {}, cap16 = {};
  if (nhalf == 1 && em_raise_lds_cap((const void*)sub2_kernel<8>, SMEM_BYTES, &cap8) != EM_OK) return EM_ERR_LAUNCH;
  if (nhalf == 2 && em_raise_lds_cap((const void*)sub2_kernel<16>, SMEM_BYTES, &cap16) != EM_OK) return EM_ERR_LAUNCH;
  Sub2Args a;
  a.feats = feats; a.partial = mvn_partial; a.flens = flens;
  a.w1f = conv1_wf; a.w2f = conv2_wf; a.b2 = conv2_b; a.out = c2; a.ldo = d;
  a.B = B; a.T_f = T_f; a.n_mels = n_mels; a.F1 = F1; a.T2 = T2; a.F2 = F2;
  static long long* stamps = nullptr;
  const bool want_stamps = em_sw().sub2_stamps;
  if (want_stamps && !stamps && hipMalloc((void**)&stamps, 32 * sizeof(long long)) != hipSuccess) return EM_ERR_LAUNCH;
  if (want_stamps && hipMemsetAsync(stamps, 0, 32 * sizeof(long long), (hipStream_t)stream) != hipSuccess) return EM_ERR_LAUNCH;
  a.stamps = want_stamps ? stamps : nullptr;
  const bool rec = em_prof_begin(stream);
  static int ncu_of[64];  // compute units per device (asked once: no runtime call on later launches, legal under stream capture)
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
    if (ncu_of[dev] == 0) {
      hipDeviceProp_t prop;
      ncu_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    ncu = ncu_of[dev];
  }
  const int ntiles = em_cdiv(T2, TT) * B;
  int grid = ntiles < ncu ? ntiles : ncu;
  if (em_cdiv(ntiles, grid) > MAXK) grid = em_cdiv(ntiles, MAXK);  // (very large batches: more than one workgroup per CU in sequence)
  if (nhalf == 1) {
    hipLaunchKernelGGL(sub2_kernel<8>, dim3(grid), dim3(256), SMEM_BYTES, (hipStream_t)stream, a);
  } else {
    for (int hf = 0; hf < 2; ++hf) {
      a.w2f = (const unsigned char*)conv2_wf + (size_t)hf * 16 * 9 * 16384;
      a.b2 = conv2_b + hf * D;
      a.out = (unsigned char*)c2 + (size_t)hf * D * 2;
      hipLaunchKernelGGL(sub2_kernel<16>, dim3(grid), dim3(256), SMEM_BYTES, (hipStream_t)stream, a);
    }
  }
  if (want_stamps) {
    long long h[32];
    if (hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
      printf("[sub2 stamps, cycles since entry]");
      for (int i = 1; i < 32 && h[i]; ++i) printf(" %lld", h[i] - h[0]);
      printf("\n");
      fflush(stdout);
    }
  }
  if (rec) em_prof_end(stream, 2.0 * (double)B * T2 * F2 * d * 9.0 * d, EM_PROF_GEMM);
  EM_CHECK_LAUNCH();
  return EM_OK;
}
