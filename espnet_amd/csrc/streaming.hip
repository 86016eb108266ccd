// Streaming (contextual block) Conformer encoder step — SURVEY.md §8(a) A16, BASELINE config 5.
//
// Reference: ContextualBlockConformerEncoder.forward_infer
// (espnet2/asr/encoder/contextual_block_conformer_encoder.py:386-600) and
// ContextualBlockEncoderLayer.forward_infer
// (espnet2/legacy/nets/pytorch_backend/conformer/contextual_block_encoder_layer.py:197-310).
//
// The per-call unit of work is a tensor of n_blk blocks x L = block_size + 2 slots x d (slot 0 =
// context inherited from the previous block, slots 1..block_size = frames, slot L-1 = this block's
// context).  All shapes of a steady-state call are static, so the whole step (≈170 launches for 12
// layers) is captured once into a hipGraph by the host layer and replayed per chunk.
//
// Kernels here: block assembly (StreamPositionalEncoding + block means, embedding.py:376-389,
// encoder :512-536), plain multi-head attention inside a block with the contextual mask
// (attention.py:121-151, encoder :539-544), context propagation between layers (layer :292-304).
// The dense parts reuse gemm.hip / norm.hip / conv.hip.
#include <math.h>

#include <stdlib.h>

#include "em_common.h"
#include "switches.h"

namespace {

// One workgroup per block i.  xs [total][d] f32 (subsampled frames of this call incl. the carried
// buffer), pe [>= ...][d] absolute sinusoid table, x out [n_blk][L][d] f32.
//   slot L-1 = addin_i   = mean(frames of block i) * sqrt(d) + pe[i + n_proc]
//   slot 0   = addin_{i-1} (i > 0), else prev_addin carried from the previous call, else addin_0
//   slot 1+j = xs[cur + j] * sqrt(d) + pe[cur + hop * n_proc + j]; unused slots are zero
__global__ __launch_bounds__(256) void cb_build_blocks_kernel_v1(const float* __restrict__ xs,
                                                              const float* __restrict__ pe,
                                                              const float* __restrict__ prev_addin,
                                                              int n_proc_host,
                                                              const int* __restrict__ n_proc_dev, int np_stride,
                                                              int total, int bs, int hs,
                                                              int d, float xscale,
                                                              float* __restrict__ x,
                                                              float* __restrict__ addin_out) {
  const int i = blockIdx.x, L = bs + 2;
  // blockIdx.y = stream of a batch of lock-step streams (em_cb_build_blocks_batch_f32): its frames, its carried
  // context, its blocks
  const int sidx = blockIdx.y;
  xs += (size_t)sidx * total * d;
  if (prev_addin) prev_addin += (size_t)sidx * d;
  x += (size_t)sidx * gridDim.x * L * d;
  addin_out += (size_t)sidx * d;
  // the number of blocks already processed feeds the positional-encoding offsets; a hipGraph-
  // captured step reads it from device memory (kernel arguments are frozen at capture)
  // (np_stride 1: one count PER STREAM - streams of a batch that joined at different times, em_cb_build_blocks_rows_f32)
  const int n_proc = n_proc_dev ? n_proc_dev[(size_t)sidx * np_stride] : n_proc_host;
  float* xb = x + (size_t)i * L * d;
  const int cur = i * hs;
  const int clen = (total - cur) < bs ? (total - cur) : bs;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float sum = 0.f;
    for (int j = 0; j < clen; ++j) {
      const float v = xs[(size_t)(cur + j) * d + c];
      sum += v;
      xb[(size_t)(1 + j) * d + c] = v * xscale + pe[(size_t)(cur + hs * n_proc + j) * d + c];
    }
    for (int j = clen; j < bs; ++j) xb[(size_t)(1 + j) * d + c] = 0.f;
    const float addin = (sum / (float)clen) * xscale + pe[(size_t)(i + n_proc) * d + c];
    xb[(size_t)(L - 1) * d + c] = addin;
    float first;
    if (i > 0) {  // the previous block's context: recomputed here, no inter-workgroup dependency
      const int pcur = (i - 1) * hs;
      const int plen = (total - pcur) < bs ? (total - pcur) : bs;
      float ps = 0.f;
      for (int j = 0; j < plen; ++j) ps += xs[(size_t)(pcur + j) * d + c];
      first = (ps / (float)plen) * xscale + pe[(size_t)(i - 1 + n_proc) * d + c];
    } else {
      first = prev_addin ? prev_addin[c] : addin;
    }
    xb[c] = first;
    if (i == gridDim.x - 1) addin_out[c] = addin;
  }
}

// Round 6: the same block, rows dealt to four thread groups (a thread: four columns of every fourth row) instead of a thread
// per column walking all 40 rows - 13.4 us of a serial walk per tick for 1.4 MB (profiles/r06w_stream_tick_order.txt).  The
// block's mean is the sum of the four groups' sums in a fixed order ((g0 + g1) + (g2 + g3)): deterministic, the same for one
// stream and for a batch; against the serial sum of the first version it differs by f32 round-off only.  d % 4 == 0, d <= 1024.
__global__ __launch_bounds__(256) void cb_build_blocks_kernel(const float* __restrict__ xs,
                                                              const float* __restrict__ pe,
                                                              const float* __restrict__ prev_addin,
                                                              int n_proc_host,
                                                              const int* __restrict__ n_proc_dev, int np_stride,
                                                              int total, int bs, int hs,
                                                              int d, float xscale,
                                                              float* __restrict__ x,
                                                              float* __restrict__ addin_out) {
  __shared__ float4 part[2][4][256];
  const int i = blockIdx.x, L = bs + 2, nq = d >> 2;
  const int sidx = blockIdx.y;
  xs += (size_t)sidx * total * d;
  if (prev_addin) prev_addin += (size_t)sidx * d;
  x += (size_t)sidx * gridDim.x * L * d;
  addin_out += (size_t)sidx * d;
  const int n_proc = n_proc_dev ? n_proc_dev[(size_t)sidx * np_stride] : n_proc_host;
  float* xb = x + (size_t)i * L * d;
  const int cur = i * hs;
  const int clen = (total - cur) < bs ? (total - cur) : bs;
  const int pcur = (i - 1) * hs;
  const int plen = i > 0 ? ((total - pcur) < bs ? (total - pcur) : bs) : 0;
  const int g = threadIdx.x >> 6, q0 = threadIdx.x & 63;
  auto add4 = [](float4 a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; return a; };
  for (int q = q0; q < nq; q += 64) {
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f), ps = sum;
    for (int j = g; j < bs; j += 4) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < clen) {
        const float4 v = *(const float4*)(xs + (size_t)(cur + j) * d + 4 * q);
        const float4 p = *(const float4*)(pe + (size_t)(cur + hs * n_proc + j) * d + 4 * q);
        sum = add4(sum, v);
        o = make_float4(v.x * xscale + p.x, v.y * xscale + p.y, v.z * xscale + p.z, v.w * xscale + p.w);
      }
      *(float4*)(xb + (size_t)(1 + j) * d + 4 * q) = o;
    }
    for (int j = g; j < plen; j += 4) ps = add4(ps, *(const float4*)(xs + (size_t)(pcur + j) * d + 4 * q));
    part[0][g][q] = sum;
    part[1][g][q] = ps;
  }
  __syncthreads();
  if (g == 0) {
    for (int q = q0; q < nq; q += 64) {
      const float4 s4 = add4(add4(part[0][0][q], part[0][1][q]), add4(part[0][2][q], part[0][3][q]));
      const float4 pa = *(const float4*)(pe + (size_t)(i + n_proc) * d + 4 * q);
      const float fc = (float)clen;
      const float4 addin = make_float4((s4.x / fc) * xscale + pa.x, (s4.y / fc) * xscale + pa.y, (s4.z / fc) * xscale + pa.z,
                                       (s4.w / fc) * xscale + pa.w);
      *(float4*)(xb + (size_t)(L - 1) * d + 4 * q) = addin;
      float4 first;
      if (i > 0) {  // the previous block's context: recomputed here, no inter-workgroup dependency
        const float4 p4 = add4(add4(part[1][0][q], part[1][1][q]), add4(part[1][2][q], part[1][3][q]));
        const float4 pp = *(const float4*)(pe + (size_t)(i - 1 + n_proc) * d + 4 * q);
        const float fp = (float)plen;
        first = make_float4((p4.x / fp) * xscale + pp.x, (p4.y / fp) * xscale + pp.y, (p4.z / fp) * xscale + pp.z,
                            (p4.w / fp) * xscale + pp.w);
      } else {
        first = prev_addin ? *(const float4*)(prev_addin + 4 * q) : addin;
      }
      *(float4*)(xb + 4 * q) = first;
      if (i == (int)gridDim.x - 1) *(float4*)(addin_out + 4 * q) = addin;
    }
  }
}

// Plain multi-head attention inside each block: one 256-thread workgroup per (head, block); four
// threads share a query slot, each owning DK/4 channels (partial dot products are combined with two
// shuffles, the context accumulates per channel slice).  qkv [n_blk*L][3d] act (q | k | v).
// mask_mode 1 = contextual mask (encoder :539-544): query slot 0 attends to nothing (its output is
// 0), every other slot attends to slots 0..L-2; mask_mode 0 = no mask (short-utterance path).
template <typename T, int DK>
__global__ __launch_bounds__(256) void block_mha_kernel(const T* __restrict__ qkv, int L, int d,
                                                        int mask_mode, T* __restrict__ ctx) {
  constexpr int DKP = DK / 4;
  extern __shared__ float sm[];
  float* Ks = sm;                 // [L][DK]
  float* Vs = Ks + L * DK;        // [L][DK]
  float* S = Vs + L * DK;         // [64][L + 1] scaled scores
  const int h = blockIdx.x, blk = blockIdx.y, tid = threadIdx.x;
  const int r = tid >> 2, part = tid & 3;
  const T* base = qkv + (size_t)blk * L * 3 * d + h * DK;
  for (int e = tid; e < L * DK; e += 256) {
    const int j = e / DK, c = e - j * DK;
    Ks[e] = to_f32(base[(size_t)j * 3 * d + d + c]);
    Vs[e] = to_f32(base[(size_t)j * 3 * d + 2 * d + c]);
  }
  __syncthreads();
  const int nkeys = mask_mode ? L - 1 : L;
  const bool active = r < L;
  const int rq = active ? r : L - 1;
  float q[DKP];
#pragma unroll
  for (int c = 0; c < DKP; ++c) q[c] = to_f32(base[(size_t)rq * 3 * d + part * DKP + c]);
  const float scale = rsqrtf((float)DK);
  float mx = -INFINITY;
  for (int j = 0; j < nkeys; ++j) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DKP; ++c) s = fmaf(q[c], Ks[j * DK + part * DKP + c], s);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s *= scale;
    mx = fmaxf(mx, s);
    if (part == 0) S[r * (L + 1) + j] = s;
  }
  __syncthreads();
  float acc[DKP];
#pragma unroll
  for (int c = 0; c < DKP; ++c) acc[c] = 0.f;
  float sum = 0.f;
  for (int j = 0; j < nkeys; ++j) {
    const float p = expf(S[r * (L + 1) + j] - mx);
    sum += p;
#pragma unroll
    for (int c = 0; c < DKP; ++c) acc[c] = fmaf(p, Vs[j * DK + part * DKP + c], acc[c]);
  }
  if (!active) return;
  T* out = ctx + ((size_t)blk * L + r) * d + h * DK + part * DKP;
  const float inv = (mask_mode && r == 0) ? 0.f : 1.0f / sum;
#pragma unroll
  for (int c = 0; c < DKP; ++c) out[c] = from_f32<T>(acc[c] * inv);
}

// The same attention over the PER-HEAD operands the fused block kernels write (csrc/block.hip, EM_BLOCK_A): q / k
// [n_blk][H][Tpad][64], V transposed [n_blk][H][64][Tpad], bf16; ctx [n_blk*L][H*64] bf16.  (round 4: the fused streaming layer)
__global__ __launch_bounds__(256) void cb_mha_heads_kernel(const bf16* __restrict__ qh, const bf16* __restrict__ kh,
                                                          const bf16* __restrict__ vt, int L, int Tpad, int H,
                                                          int mask_mode, bf16* __restrict__ ctx) {
  constexpr int DK = 64, DKP = DK / 4;
  extern __shared__ float sm[];
  float* Ks = sm;                 // [L][DK]
  float* Vs = Ks + L * DK;        // [L][DK]
  float* S = Vs + L * DK;         // [64][L + 1] scaled scores
  const int h = blockIdx.x, blk = blockIdx.y, tid = threadIdx.x;
  const int r = tid >> 2, part = tid & 3;
  const size_t bh = (size_t)blk * H + h;
  const bf16* qb = qh + bh * Tpad * DK;
  const bf16* kb = kh + bh * Tpad * DK;
  const bf16* vb = vt + bh * DK * Tpad;
  for (int e = tid; e < L * DK; e += 256) {
    const int j = e / DK, c = e - j * DK;
    Ks[e] = (float)kb[(size_t)j * DK + c];
  }
  for (int e = tid; e < L * DK; e += 256) {  // (V^T rows are contiguous in j: read along j, store transposed)
    const int c = e / L, j = e - c * L;
    Vs[j * DK + c] = (float)vb[(size_t)c * Tpad + j];
  }
  __syncthreads();
  const int nkeys = mask_mode ? L - 1 : L;
  const bool active = r < L;
  const int rq = active ? r : L - 1;
  float q[DKP];
#pragma unroll
  for (int c = 0; c < DKP; ++c) q[c] = (float)qb[(size_t)rq * DK + part * DKP + c];
  const float scale = rsqrtf((float)DK);
  float mx = -INFINITY;
  for (int j = 0; j < nkeys; ++j) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DKP; ++c) s = fmaf(q[c], Ks[j * DK + part * DKP + c], s);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s *= scale;
    mx = fmaxf(mx, s);
    if (part == 0) S[r * (L + 1) + j] = s;
  }
  __syncthreads();
  float acc[DKP];
#pragma unroll
  for (int c = 0; c < DKP; ++c) acc[c] = 0.f;
  float sum = 0.f;
  for (int j = 0; j < nkeys; ++j) {
    const float p = expf(S[r * (L + 1) + j] - mx);
    sum += p;
#pragma unroll
    for (int c = 0; c < DKP; ++c) acc[c] = fmaf(p, Vs[j * DK + part * DKP + c], acc[c]);
  }
  if (!active) return;
  bf16* out = ctx + ((size_t)blk * L + r) * (H * DK) + h * DK + part * DKP;
  const float inv = (mask_mode && r == 0) ? 0.f : 1.0f / sum;
#pragma unroll
  for (int c = 0; c < DKP; ++c) out[c] = (bf16)(acc[c] * inv);
}

// Round 5: the same attention on the matrix cores, without LDS and without a barrier.  One wave per (block, head, 16 query
// slots): Q and K rows are MFMA operands as they lie in memory ([slot][64] bf16, a lane reads 16 bytes of its row); the
// scores are computed transposed, S^T[key][query] = K . Q^T, so a lane holds 16 keys of ONE query - the softmax statistics
// are per lane plus two register swaps - and the probabilities, rounded to bf16, are already the B operand of
// O^T = V^T . P^T, whose A operand is V^T as block<A> wrote it ([64][Tpad]: two 8-byte reads per fragment, the keys of the
// lane's accumulator slots).  1 / sqrt(64) is folded into the query operands (a power of two: exact); the denominator is the
// sum of the ROUNDED probabilities (a fragment of ones as a fifth A operand), as in relpos_attn2_kernel.  Slots past the
// block (rows of the 64-slot operand tiles that no workgroup of block<A> wrote, or that it filled with copies of the last
// slot) never count: their scores are replaced, not scaled, and their V columns are zeroed before the product.
// The first version (above: f32 LDS copies of K and V, four threads per query, 2 x 41 dependent LDS round trips) took
// 21.8 us per launch at 32 blocks, a fifth of the streaming layer (profiles/r04zs_stream_batch32_kernel_stats.csv).
// Reference: MultiHeadedAttention.forward (transformer/attention.py:121-151) under the contextual mask
// (contextual_block_conformer_encoder.py:539-544).  L <= 64 <= Tpad.
__global__ __launch_bounds__(256) void cb_mha_heads_mfma_kernel(const bf16* __restrict__ qh, const bf16* __restrict__ kh,
                                                               const bf16* __restrict__ vt, int L, int Tpad, int H,
                                                               int mask_mode, bf16* __restrict__ ctx) {
  using MM = Mma<bf16>;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const int h = blockIdx.x, blk = blockIdx.y, lane = threadIdx.x & 63;
  const int qt = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // this wave's 16 query slots
  if (16 * qt >= L) return;
  const int lr = lane & 15, lg = lane >> 4;
  const size_t bh = (size_t)blk * H + h;
  const bf16* const qb = qh + bh * Tpad * 64;
  const bf16* const kb = kh + bh * Tpad * 64;
  const bf16* const vb = vt + bh * 64 * Tpad;
  const int nkeys = mask_mode ? L - 1 : L;
  // ---- every operand requested up front: one round trip
  bf16x8 qraw[2], kf[4][2];
  uint2 vraw[4][2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) qraw[ks] = *(const bf16x8*)(qb + (size_t)(16 * qt + lr) * 64 + ks * 32 + lg * 8);
#pragma unroll
  for (int nf = 0; nf < 4; ++nf)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) kf[nf][ks] = *(const bf16x8*)(kb + (size_t)(16 * nf + lr) * 64 + ks * 32 + lg * 8);
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
        vraw[f][jp][hf] = *(const uint2*)(vb + (size_t)(16 * f + lr) * Tpad + 32 * jp + 16 * hf + 4 * lg);
  // ---- S^T (64 keys x 16 queries): A = K rows (key 16 nf + lr, k-slice lg), B = Q^T (query lr, k-slice lg)
  bf16x8 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[ks][e] = (bf16)((float)qraw[ks][e] * 0.125f);
  f32x4 sc[4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) sc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) sc[nf] = MM::mma(kf[nf][ks], qf[ks], sc[nf]);
  // ---- this lane: keys 16 nf + 4 lg + r of query lr
  float tm = -INFINITY;
#pragma unroll
  for (int nf = 0; nf < 4; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[nf][r] = (16 * nf + 4 * lg + r) < nkeys ? sc[nf][r] : -INFINITY;
      tm = fmaxf(tm, sc[nf][r]);
    }
  tm = wave_xor16_max(tm);  // the four lane groups of a query
  tm = wave_xor32_max(tm);
  constexpr float LOG2E = 1.4426950408889634f;
  const float mnl = tm * LOG2E;
  unsigned pbu[2][4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[nf][2 * hh], LOG2E, -mnl));
      const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[nf][2 * hh + 1], LOG2E, -mnl));
      typedef __attribute__((ext_vector_type(2))) float f32x2;
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
      pbu[nf >> 1][(nf & 1) * 2 + hh] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){e0, e1}, bf16x2));
    }
  // ---- O^T += V^T . P^T: contraction slot e of the 32-key step jp is key 32 jp + 16 (e >> 2) + 4 lg + (e & 3) on both sides
  const bf16x8 ones = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
  f32x4 acc_o[4], acc_l = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < 4; ++f) acc_o[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
    const bf16x8 pb = __builtin_bit_cast(bf16x8, (u32x4){pbu[jp][0], pbu[jp][1], pbu[jp][2], pbu[jp][3]});
    acc_l = MM::mma(ones, pb, acc_l);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      unsigned w[4];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int nv = nkeys - (32 * jp + 16 * hf + 4 * lg);  // how many of the piece's four keys exist
        w[2 * hf] = vraw[f][jp][hf].x & (nv >= 2 ? 0xffffffffu : (nv == 1 ? 0xffffu : 0u));
        w[2 * hf + 1] = vraw[f][jp][hf].y & (nv >= 4 ? 0xffffffffu : (nv == 3 ? 0xffffu : 0u));
      }
      acc_o[f] = MM::mma(__builtin_bit_cast(bf16x8, (u32x4){w[0], w[1], w[2], w[3]}), pb, acc_o[f]);
    }
  }
  // ---- lane (lr, lg): query 16 qt + lr, channels 16 f + 4 lg .. + 3
  const int q = 16 * qt + lr;
  if (q < L) {
    const bool none = mask_mode && q == 0;  // (the context slot of the contextual mask attends to nothing: zeros, selected - not scaled)
    const float inv = 1.0f / acc_l[0];
    bf16* const dst = ctx + ((size_t)blk * L + q) * (H * 64) + h * 64 + 4 * lg;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const bf16x4 pk = {(bf16)(none ? 0.f : acc_o[f][0] * inv), (bf16)(none ? 0.f : acc_o[f][1] * inv),
                         (bf16)(none ? 0.f : acc_o[f][2] * inv), (bf16)(none ? 0.f : acc_o[f][3] * inv)};
      *(bf16x4*)(dst + 16 * f) = pk;
    }
  }
}

// Context hand-over after a layer (layer :292-304), in place on x [n_blk][L][d] f32:
//   x[0][0] = past_ctx (or x[0][L-1] for the first block of an utterance); x[b][0] = x[b-1][L-1];
//   next_ctx = x[n_blk-1][L-1].
__global__ __launch_bounds__(256) void cb_propagate_ctx_kernel(float* __restrict__ x,
                                                               const float* __restrict__ past_ctx,
                                                               float* __restrict__ next_ctx,
                                                               int n_blk, int L, int d, int ctx_stride) {
  const int b = blockIdx.x;
  // blockIdx.y = stream of a batch of lock-step streams: its blocks, its context vectors (ctx_stride floats apart)
  x += (size_t)blockIdx.y * n_blk * L * d;
  if (past_ctx) past_ctx += (size_t)blockIdx.y * ctx_stride;
  if (next_ctx) next_ctx += (size_t)blockIdx.y * ctx_stride;
  float* xb = x + (size_t)b * L * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float v;
    if (b == 0) v = past_ctx ? past_ctx[c] : xb[(size_t)(L - 1) * d + c];
    else v = x[((size_t)(b - 1) * L + (L - 1)) * d + c];
    xb[c] = v;
    if (b == n_blk - 1 && next_ctx) next_ctx[c] = xb[(size_t)(L - 1) * d + c];
  }
}

// out[j] = xs[j] * sqrt(d) + pe[start + j]   (StreamPositionalEncoding.forward, embedding.py:376-389)
__global__ __launch_bounds__(256) void stream_pos_enc_kernel(const float* __restrict__ xs,
                                                             const float* __restrict__ pe, int start,
                                                             int d, float xscale,
                                                             float* __restrict__ out) {
  const int j = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += blockDim.x)
    out[(size_t)j * d + c] = xs[(size_t)j * d + c] * xscale + pe[(size_t)(start + j) * d + c];
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct CbWs {
  size_t xn, big, g, g2, ctx, qh, kh, vt, part, ticket, total;
  int Tpad;
};
inline CbWs cb_layout(int dtype, const EmConformerWeights* w, int M) {
  const size_t es = dtype == EM_BF16 ? 2 : 4;
  const size_t d = w->d;
  const size_t wide = (size_t)w->ff > 3 * d ? w->ff : 3 * d;
  CbWs s;
  size_t o = 0;
  s.xn = o; o += align_up((size_t)M * d * es);
  s.big = o; o += align_up((size_t)M * wide * es);
  s.g = o; o += align_up((size_t)M * d * es);
  s.g2 = o; o += align_up((size_t)M * d * es);
  s.ctx = o; o += align_up((size_t)M * d * es);
  s.total = o;
  return s;
}
// ... plus the per-head operands of the fused layer (bf16, d = 256: csrc/block.hip): q / k [n_blk][4][Tpad][64],
// V^T [n_blk][4][64][Tpad] with Tpad = the 32-row workgroups' reach
// Shares of an FFN's hidden dimension per 32-row block in the fused streaming layers (EmBlockArgs.ffn_split, round 6).  A tick
// of a few streams runs every launch on a few of the chip's 256 CUs, each pushing the whole 2 MiB of an FFN through ONE CU's
// MFMA pipes (~8 us of a ~20 us launch); S workgroups per row block take 1 / S of it each and meet once (a store, a ticket,
// the last one sums: 3 - 5 us, growing with S).  Measured per call of 12 layers (profiles/r06t_stream_split_sweep.txt):
//   streams (row blocks)   1 (2)   8 (16)   16 (32)   32 (64)   64 (128)
//   S = 1                  926     941      941       945       946   us
//   S = 2                  788     807      846       874      1029
//   S = 4                  720     772      840       996      1430
//   S = 8                  742     826     1030      1429      2184
// so: 4 up to 32 row blocks, 2 up to 64, 1 beyond; whole pairs of 64-wide chunks per share.
// ESPNET_AMD_STREAM_FFN_SPLIT=n: developer switch (1 = off, n = forced).
inline int cb_ffn_split(const EmConformerWeights* w, int row_blocks) {
  const int forced = em_sw().stream_ffn_split;
  const int pairs = (((w->ff >> 6) + 1) & ~1) / 2;
  int S = forced > 0 ? forced : row_blocks <= 32 ? 4 : row_blocks <= 64 ? 2 : 1;
  if (S > 16) S = 16;
  while (S > 1 && pairs % S != 0) --S;
  return S;
}
inline CbWs cb_layout_fused(int dtype, const EmConformerWeights* w, int n_blk, int L) {
  CbWs s = cb_layout(dtype, w, n_blk * L);
  s.Tpad = (L + 63) / 64 * 64;
  const size_t per_head = (size_t)n_blk * w->d * s.Tpad * 2;
  size_t o = s.total;
  s.qh = o; o += align_up(per_head);
  s.kh = o; o += align_up(per_head);
  s.vt = o; o += align_up(per_head);
  // split FFN of the row-block launches (round 6, EmBlockArgs.ffn_split): partial sums [row blocks][S][8192] f32 + tickets
  const int nrb = n_blk * ((L + 31) / 32), S = cb_ffn_split(w, nrb);
  s.part = o; o += S > 1 ? align_up((size_t)nrb * S * 8192 * 4) : 0;
  s.ticket = o; o += S > 1 ? align_up((size_t)nrb * 4) : 0;
  s.total = o;
  return s;
}
// Which streaming layers take the fused launch sequence: bf16, 256 wide, 4 heads, conv width 15, ff <= 4096 in whole
// chunk pairs, blocks of at most 64 slots, every layer packed for it by the host.
// Rounds 4 and 5 also asked for at least 8 blocks in the call: a workgroup of the fused kernels pushes ALL of a layer's 5 MB of
// weights through one CU, and with one block (one stream, one call) that was two workgroups on an otherwise empty chip - 1.47 ms
// per call against 1.15 for the per-operator sequence, whose GEMMs spread every weight matrix over a hundred CUs
// (profiles/r04i_stream_fused_ab.txt; equal in round 5 with the helper workgroups).  Round 6: with each FFN dealt to four
// workgroups per row block (cb_ffn_split) the fused layers win from one block on - 720 us against 870 per call
// (profiles/r06t_stream_split_sweep.txt) - and the floor is gone.  ESPNET_AMD_STREAM_FUSED_MIN=n: developer A/B switch.
inline bool cb_fusable(int dtype, const EmConformerWeights* w, int L, int n_blk) {
  const bool off = em_sw().stream_no_fused;  // developer A/B switch
  const int min_blk = em_sw().stream_fused_min;
  if (off || n_blk < min_blk || dtype != EM_BF16 || w->d != 256 || w->heads != 4 || w->kernel != 15 || w->ff > 4096 || w->ff % 128 != 0 ||
      L > 64 || !w->layers)
    return false;
  for (int l = 0; l < w->num_blocks; ++l) {
    const EmConformerLayer& q = w->layers[l];
    if (!q.fp_a || !q.fp_c || !q.fp_da || !q.ffm_w1p || !q.ffm_w2p || !q.wqkvp || !q.woutp || !q.pw1f || !q.pw2p ||
        !q.ff_w1p || !q.ff_w2p || !q.ffm_b1 || !q.ff_b1)
      return false;
  }
  return true;
}

inline int gemm(int dtype, int epi, const void* A, const void* W, void* C, const float* bias, int M,
                int N, int K, int lda, int ldc, float scale, void* stream) {
  EmGemmArgs a = {};
  a.A = A; a.W = W; a.C = C; a.bias = bias;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.scale = scale;
  return em_gemm(dtype, epi, EM_A_PLAIN, &a, stream);
}

#define EM_TRY(expr)                \
  do {                              \
    int rc__ = (expr);              \
    if (rc__ != EM_OK) return rc__; \
  } while (0)

constexpr float LN_EPS = 1e-12f;

}  // namespace

extern "C" int em_cb_build_blocks_f32(const float* xs, const float* pe, const float* prev_addin,
                                      int32_t n_proc, const int32_t* n_proc_dev, int32_t n_blk,
                                      int32_t total, int32_t bs,
                                      int32_t hs, int32_t d, float* x, float* addin_out,
                                      void* stream) {
  if (!xs || !pe || !x || !addin_out || n_blk <= 0 || total <= 0 || bs <= 0 || hs <= 0 || d <= 0)
    return EM_ERR_BAD_ARG;
  if ((n_blk - 1) * hs >= total) return EM_ERR_BAD_ARG;  // every block holds at least one frame
  if (d % 4 == 0 && d <= 1024)
    hipLaunchKernelGGL(cb_build_blocks_kernel, dim3(n_blk), dim3(256), 0, (hipStream_t)stream, xs, pe,
                     prev_addin, n_proc, n_proc_dev, 0, total, bs, hs, d, sqrtf((float)d), x, addin_out);
  else
    hipLaunchKernelGGL(cb_build_blocks_kernel_v1, dim3(n_blk), dim3(256), 0, (hipStream_t)stream, xs, pe,
                     prev_addin, n_proc, n_proc_dev, 0, total, bs, hs, d, sqrtf((float)d), x, addin_out);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_cb_build_blocks_batch_f32(const float* xs, const float* pe, const float* prev_addin, int32_t n_proc,
                                            int32_t n_streams, int32_t n_blk, int32_t total, int32_t bs, int32_t hs,
                                            int32_t d, float* x, float* addin_out, void* stream) {
  if (!xs || !pe || !x || !addin_out || n_streams <= 0 || n_blk <= 0 || total <= 0 || bs <= 0 || hs <= 0 || d <= 0)
    return EM_ERR_BAD_ARG;
  if ((n_blk - 1) * hs >= total) return EM_ERR_BAD_ARG;  // every block holds at least one frame
  if (d % 4 == 0 && d <= 1024)
    hipLaunchKernelGGL(cb_build_blocks_kernel, dim3(n_blk, n_streams), dim3(256), 0, (hipStream_t)stream, xs, pe,
                     prev_addin, n_proc, (const int*)nullptr, 0, total, bs, hs, d, sqrtf((float)d), x, addin_out);
  else
    hipLaunchKernelGGL(cb_build_blocks_kernel_v1, dim3(n_blk, n_streams), dim3(256), 0, (hipStream_t)stream, xs, pe,
                     prev_addin, n_proc, (const int*)nullptr, 0, total, bs, hs, d, sqrtf((float)d), x, addin_out);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_cb_build_blocks_rows_f32(const float* xs, const float* pe, const float* prev_addin,
                                           const int32_t* n_proc_rows, int32_t n_streams, int32_t n_blk, int32_t total,
                                           int32_t bs, int32_t hs, int32_t d, float* x, float* addin_out, void* stream) {
  if (!xs || !pe || !x || !addin_out || !n_proc_rows || n_streams <= 0 || n_blk <= 0 || total <= 0 || bs <= 0 ||
      hs <= 0 || d <= 0)
    return EM_ERR_BAD_ARG;
  if ((n_blk - 1) * hs >= total) return EM_ERR_BAD_ARG;  // every block holds at least one frame
  if (d % 4 == 0 && d <= 1024)
    hipLaunchKernelGGL(cb_build_blocks_kernel, dim3(n_blk, n_streams), dim3(256), 0, (hipStream_t)stream, xs, pe,
                     prev_addin, 0, n_proc_rows, 1, total, bs, hs, d, sqrtf((float)d), x, addin_out);
  else
    hipLaunchKernelGGL(cb_build_blocks_kernel_v1, dim3(n_blk, n_streams), dim3(256), 0, (hipStream_t)stream, xs, pe,
                     prev_addin, 0, n_proc_rows, 1, total, bs, hs, d, sqrtf((float)d), x, addin_out);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_stream_pos_enc_f32(const float* xs, const float* pe, int32_t start, int32_t n,
                                     int32_t d, float* out, void* stream) {
  if (!xs || !pe || !out || n <= 0 || d <= 0 || start < 0) return EM_ERR_BAD_ARG;
  hipLaunchKernelGGL(stream_pos_enc_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, xs, pe, start,
                     d, sqrtf((float)d), out);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_block_mha(int dtype, const void* qkv, int32_t n_blk, int32_t L, int32_t d,
                            int32_t heads, int32_t mask_mode, void* ctx, void* stream) {
  if (!qkv || !ctx || n_blk <= 0 || L <= 0 || L > 64 || heads <= 0) return EM_ERR_BAD_ARG;
  const int dk = d / heads;
  const size_t lds = ((size_t)2 * L * dk + (size_t)64 * (L + 1)) * sizeof(float);
  dim3 grid(heads, n_blk);
  hipStream_t s = (hipStream_t)stream;
#define EM_MHA_CASE(TT, DKK)                                                                   \
  hipLaunchKernelGGL((block_mha_kernel<TT, DKK>), grid, dim3(256), lds, s, (const TT*)qkv, L, d, \
                     mask_mode, (TT*)ctx)
  if (dtype == EM_F32 && dk == 64) EM_MHA_CASE(float, 64);
  else if (dtype == EM_F32 && dk == 32) EM_MHA_CASE(float, 32);
  else if (dtype == EM_BF16 && dk == 64) EM_MHA_CASE(bf16, 64);
  else if (dtype == EM_BF16 && dk == 32) EM_MHA_CASE(bf16, 32);
  else return EM_ERR_UNSUPPORTED;
#undef EM_MHA_CASE
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_cb_propagate_ctx_f32(float* x, const float* past_ctx, float* next_ctx,
                                       int32_t n_blk, int32_t L, int32_t d, void* stream) {
  if (!x || n_blk <= 0 || L < 2 || d <= 0) return EM_ERR_BAD_ARG;
  hipLaunchKernelGGL(cb_propagate_ctx_kernel, dim3(n_blk), dim3(256), 0, (hipStream_t)stream, x,
                     past_ctx, next_ctx, n_blk, L, d, 0);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" size_t em_cb_workspace_bytes(int dtype, const EmConformerWeights* w, int32_t n_blk,
                                        int32_t L) {
  if (!w || n_blk <= 0 || L <= 0) return 0;
  return cb_fusable(dtype, w, L, n_blk) ? cb_layout_fused(dtype, w, n_blk, L).total : cb_layout(dtype, w, n_blk * L).total;
}

// All layers of the block encoder on x [n_blk][L][d] f32 (in place).  past_ctx / next_ctx:
// [num_blocks][d] f32 context vectors carried across calls (past_ctx NULL = first call of an
// utterance; both NULL with mask_mode 0 = short-utterance path without context slots).
// EmConformerLayer fields are reused: norm_mha = norm1, norm_ff = norm2, no pos_u / pos_v; the
// feed-forward activation is ReLU (contextual_block_conformer_encoder.py:148-154).
static int cb_encode_blocks_impl(int dtype, const EmConformerWeights* w, float* x, int32_t n_streams, int32_t n_blk_s,
                                 int32_t L, int32_t mask_mode, const float* past_ctx, float* next_ctx, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (!w || !x || !workspace || n_streams <= 0 || n_blk_s <= 0 || L <= 0) return EM_ERR_BAD_ARG;
  if (dtype != EM_F32 && dtype != EM_BF16) return EM_ERR_BAD_ARG;
  // n_streams lock-step streams of n_blk_s blocks each: the dense operators see n_streams * n_blk_s independent blocks;
  // only the context hand-over between layers knows which blocks follow each other
  const int n_blk = n_streams * n_blk_s;
  const int d = w->d, h = w->heads, ff = w->ff, NL = w->num_blocks, M = n_blk * L;
  if (d % 64 != 0 || ff % 64 != 0 || (d / h != 64 && d / h != 32)) return EM_ERR_UNSUPPORTED;
  const bool fused = cb_fusable(dtype, w, L, n_blk);
  const CbWs s = fused ? cb_layout_fused(dtype, w, n_blk, L) : cb_layout(dtype, w, M);
  if (workspace_bytes < s.total) return EM_ERR_WORKSPACE;
  unsigned char* ws = (unsigned char*)workspace;
  void *xn = ws + s.xn, *big = ws + s.big, *gl = ws + s.g, *g2 = ws + s.g2, *ctx = ws + s.ctx;
  if (fused) {
    // ---- round 4: the layer as FIVE launches instead of thirteen, on the row-block kernels of the Conformer path
    // (csrc/block.hip; a block of L <= 64 slots = an "utterance" of two 32-row workgroups); round 6: THREE -
    //   block<A | RELU>        norm_ff_macaron + macaron FFN (ReLU, ff up to 4096) + norm_mha + q / k / v per head
    //   block<ATT | C | RELU>  plain multi-head attention over the block's slots (contextual mask) in front of
    //                          linear_out + residual, norm_conv, pointwise_conv1 + GLU   (rounds 4 - 5: cb_mha_heads + block<C>)
    //   block<D | RELU>        depthwise conv (k = 15) + BN + Swish, pointwise_conv2 + residual, norm_ff, FFN, norm_final
    //   (propagate_ctx: the context hand-over between blocks, why D and the next layer's A stay separate launches; with one
    //   block per stream and call it is folded into A and D: fold_ctx below)
    // On a chip the launch does not fill, the FFNs of A and D are dealt to 2 - 4 workgroups per row block (cb_ffn_split).
    // contextual_block_encoder_layer.py:197-310.  One stream, one block: 1.18 ms (round 4) -> 0.64 ms per call of 12 layers.
    EmBlockArgs ba = {};
    ba.B = n_blk; ba.T = L; ba.Tpad = s.Tpad; ba.d = d; ba.ff = ff; ba.kernel = w->kernel; ba.eps = LN_EPS;
    ba.x = x; ba.ctx = ctx; ba.glu = gl; ba.qh = ws + s.qh; ba.kh = ws + s.kh; ba.vt = ws + s.vt;
    const size_t mha_lds = ((size_t)2 * L * 64 + (size_t)64 * (L + 1)) * sizeof(float);
    const bool mha_v1 = em_sw().stream_mha_v1;  // developer A/B switch: the LDS / VALU attention kernel of round 4
    // round 6: the block attention in front of block<C>, in its launch (csrc/block.hip, EM_BLOCK_ATT | EM_BLOCK_C | EM_BLOCK_RELU).
    // ESPNET_AMD_STREAM_SPLIT_ATT: developer A/B switch - the attention launch of rounds 4 - 5 (bit for bit the same rows)
    const bool att_c = !mha_v1 && !em_sw().stream_split_att;
    ba.att_mask = mask_mode;
    // One block per stream (the steady-state tick): the hand-over after layer l is "slot 0 := the previous call's context
    // vector of layer l; this call's := the last slot" - block<A> of layer l + 1 reads its slot 0 from past_ctx and
    // block<D> of layer l writes its last slot to next_ctx as well: twelve launches less per call (4.9 us each, 59 of a
    // 32-stream tick's 1 170 us; profiles/r05x_stream_batch32_kernel_stats.csv).  What x[.][0] holds after the LAST layer
    // is never read (slot 0 is not an output frame).  ESPNET_AMD_STREAM_NO_CTX_FOLD: developer A/B switch (read per call:
    // tests/test_gpu_streaming.py compares the two bit for bit).
    // (past_ctx == next_ctx: layer l + 1 would read THIS call's vector where it wants the previous call's - the hand-over launch reads
    // before it writes and is safe in place, so an aliased call keeps it)
    const bool fold_ctx = mask_mode && n_blk_s == 1 && past_ctx && next_ctx && past_ctx != next_ctx && !em_sw().stream_no_ctx_fold;
    ba.row_stride = NL * d;
    // Round 6: a tick that leaves CUs idle deals each FFN's hidden dimension to S workgroups per row block (cb_ffn_split;
    // csrc/block.hip, "the S shares of this row block meet").  The tickets start at zero and every launch leaves them there.
    const int ffn_s = cb_ffn_split(w, n_blk * ((L + 31) / 32));
    if (ffn_s > 1) {
      ba.ffn_split = ffn_s; ba.ffn_part = (float*)(ws + s.part); ba.ffn_ticket = (int32_t*)(ws + s.ticket);
      if (hipMemsetAsync(ba.ffn_ticket, 0, (size_t)n_blk * ((L + 31) / 32) * 4, (hipStream_t)stream) != hipSuccess) return EM_ERR_LAUNCH;
    }
    for (int l = 0; l < NL; ++l) {
      const EmConformerLayer& q = w->layers[l];
      ba.ffm_w1 = q.ffm_w1p; ba.ffm_w2 = q.ffm_w2p; ba.wqkv = q.wqkvp; ba.ffm_b1g = q.ffm_b1; ba.params = q.fp_a;
      ba.row0_src = (fold_ctx && l > 0) ? past_ctx + (size_t)(l - 1) * d : nullptr;
      ba.last_dst = nullptr;
      EM_TRY(em_conformer_block_fused(EM_BLOCK_A | EM_BLOCK_RELU, &ba, stream));
      ba.row0_src = nullptr;
      ba.wout = q.woutp; ba.pw1f = q.pw1f; ba.params = q.fp_c;
      if (att_c) {  // round 6: the attention in front of the C part, one launch
        EM_TRY(em_conformer_block_fused(EM_BLOCK_ATT | EM_BLOCK_C | EM_BLOCK_RELU, &ba, stream));
      } else {
        if (mha_v1)
          hipLaunchKernelGGL(cb_mha_heads_kernel, dim3(h, n_blk), dim3(256), mha_lds, (hipStream_t)stream,
                             (const bf16*)ba.qh, (const bf16*)ba.kh, (const bf16*)ba.vt, L, s.Tpad, h, mask_mode, (bf16*)ctx);
        else
          hipLaunchKernelGGL(cb_mha_heads_mfma_kernel, dim3(h, n_blk), dim3(256), 0, (hipStream_t)stream,
                             (const bf16*)ba.qh, (const bf16*)ba.kh, (const bf16*)ba.vt, L, s.Tpad, h, mask_mode, (bf16*)ctx);
        EM_CHECK_LAUNCH();
        EM_TRY(em_conformer_block_fused(EM_BLOCK_C, &ba, stream));
      }
      ba.pw2 = q.pw2p; ba.ff_w1 = q.ff_w1p; ba.ff_w2 = q.ff_w2p; ba.dw_w = q.dw_w; ba.dw_b = q.dw_b; ba.ff_b1g = q.ff_b1;
      ba.params = q.fp_da;
      ba.last_dst = fold_ctx ? next_ctx + (size_t)l * d : nullptr;
      EM_TRY(em_conformer_block_fused(EM_BLOCK_D | EM_BLOCK_RELU, &ba, stream));
      ba.last_dst = nullptr;
      if (mask_mode && !fold_ctx) {
        hipLaunchKernelGGL(cb_propagate_ctx_kernel, dim3(n_blk_s, n_streams), dim3(256), 0, (hipStream_t)stream, x,
                           past_ctx ? past_ctx + (size_t)l * d : nullptr, next_ctx ? next_ctx + (size_t)l * d : nullptr,
                           n_blk_s, L, d, NL * d);
        EM_CHECK_LAUNCH();
      }
    }
    return EM_OK;
  }
  // A pre-norm LayerNorm rides in the prologue of the projection that consumes it (csrc/ln_gemm.hip) where that kernel
  // has the epilogue (plain / ReLU): three launches less per layer.  A step is ~200 dependent launches of ~5.6 us for
  // 42 rows - launch latency, nothing else - so the count is what matters (round 3: 1.30 -> see profiles/r03p).
  const bool no_lng = em_sw().stream_no_ln_gemm;  // developer A/B switch
  const bool lng = !no_lng && d % 64 == 0 && d <= 1024;
  auto ln_proj = [&](int epi, const float* g, const float* be, const void* W, const float* bias, void* C, int N) {
    if (lng) return em_ln_gemm(dtype, epi, x, g, be, LN_EPS, W, bias, C, M, N, d, N, stream);
    int rc = em_layernorm(dtype, x, g, be, M, d, LN_EPS, xn, nullptr, stream);
    if (rc != EM_OK) return rc;
    return gemm(dtype, epi, xn, W, C, bias, M, N, d, d, N, 1.f, stream);
  };
  for (int l = 0; l < NL; ++l) {
    const EmConformerLayer& q = w->layers[l];
    EM_TRY(ln_proj(EM_EPI_RELU, q.norm_ff_mac_g, q.norm_ff_mac_b, q.ffm_w1, q.ffm_b1, big, ff));
    EM_TRY(gemm(dtype, EM_EPI_RESID_F32, big, q.ffm_w2, x, q.ffm_b2, M, d, ff, ff, d, 0.5f, stream));
    EM_TRY(ln_proj(EM_EPI_STORE, q.norm_mha_g, q.norm_mha_b, q.wqkv, q.bqkv, big, 3 * d));
    EM_TRY(em_block_mha(dtype, big, n_blk, L, d, h, mask_mode, ctx, stream));
    EM_TRY(gemm(dtype, EM_EPI_RESID_F32, ctx, q.wout, x, q.bout, M, d, d, d, d, 1.f, stream));
    EM_TRY(em_layernorm(dtype, x, q.norm_conv_g, q.norm_conv_b, M, d, LN_EPS, xn, nullptr, stream));
    EM_TRY(gemm(dtype, EM_EPI_GLU, xn, q.pw1, gl, q.pw1_b, M, 2 * d, d, d, d, 1.f, stream));
    EM_TRY(em_dwconv_bn_swish(dtype, gl, q.dw_w, q.dw_b, nullptr, n_blk, L, d, w->kernel, g2, stream));
    EM_TRY(gemm(dtype, EM_EPI_RESID_F32, g2, q.pw2, x, q.pw2_b, M, d, d, d, d, 1.f, stream));
    EM_TRY(ln_proj(EM_EPI_RELU, q.norm_ff_g, q.norm_ff_b, q.ff_w1, q.ff_b1, big, ff));
    EM_TRY(gemm(dtype, EM_EPI_RESID_F32, big, q.ff_w2, x, q.ff_b2, M, d, ff, ff, d, 0.5f, stream));
    EM_TRY(em_layernorm_inplace_f32(x, q.norm_final_g, q.norm_final_b, M, d, LN_EPS, stream));
    if (mask_mode) {
      hipLaunchKernelGGL(cb_propagate_ctx_kernel, dim3(n_blk_s, n_streams), dim3(256), 0, (hipStream_t)stream, x,
                         past_ctx ? past_ctx + (size_t)l * d : nullptr, next_ctx ? next_ctx + (size_t)l * d : nullptr,
                         n_blk_s, L, d, NL * d);
      EM_CHECK_LAUNCH();
    }
  }
  return EM_OK;
}

extern "C" int em_cb_encode_blocks(int dtype, const EmConformerWeights* w, float* x, int32_t n_blk,
                                   int32_t L, int32_t mask_mode, const float* past_ctx,
                                   float* next_ctx, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  return cb_encode_blocks_impl(dtype, w, x, 1, n_blk, L, mask_mode, past_ctx, next_ctx, workspace, workspace_bytes,
                               stream);
}

// The same for a batch of n_streams lock-step streams: x [n_streams][n_blk][L][d], past_ctx / next_ctx
// [n_streams][num_blocks][d]; workspace for n_streams * n_blk blocks (em_cb_workspace_bytes).
extern "C" int em_cb_encode_blocks_batch(int dtype, const EmConformerWeights* w, float* x, int32_t n_streams,
                                         int32_t n_blk, int32_t L, int32_t mask_mode, const float* past_ctx,
                                         float* next_ctx, void* workspace, size_t workspace_bytes, void* stream) {
  return cb_encode_blocks_impl(dtype, w, x, n_streams, n_blk, L, mask_mode, past_ctx, next_ctx, workspace,
                               workspace_bytes, stream);
}
