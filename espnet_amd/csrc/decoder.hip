// Attention-decoder step kernels for the label-synchronous beam search.
//
// Reference: TransformerDecoder.batch_score / forward_one_step
// (espnet2/asr/decoder/transformer_decoder.py:191-311), DecoderLayer.forward with `cache`
// (espnet2/legacy/nets/pytorch_backend/transformer/decoder_layer.py:73-179),
// MultiHeadedAttention.forward (transformer/attention.py:121-151, 263-265),
// PositionalEncoding.forward (transformer/embedding.py:84-95).
//
// The reference re-projects linear_k/linear_v of the whole prefix (self-attention) and of the whole
// encoder memory (source attention) for every hypothesis, layer and step.  Here both are true
// caches with identical values:
//   * self-attention K/V of prefix position j depend only on y_0..y_j, so they are written once
//     into kc/vc[layer][j][slot] when slot's hypothesis consumes position j; hypotheses are paths in
//     a token tree and `anc[row][j]` names the slot holding position j of row's prefix.  A beam
//     reorder therefore permutes only the small `anc` table, never the K/V cache;
//   * source-attention K / V^T of the memory are computed once per utterance (em_search_init) and
//     shared by the W hypotheses of that utterance (one workgroup per (utterance, head): MFMA
//     QK^T and PV over a 16-row query tile).
#include <stdlib.h>

#include "em_common.h"
#include "switches.h"

namespace {

// x[r] = embed[tok[r]] * sqrt(d) + pe[pos]      (embedding.py:93; f32 residual stream)
// pos_dev != NULL (hipGraph-captured search step): the position comes from device memory and
// tok_row is then the BASE of the [pos][n] token table.
__global__ __launch_bounds__(128) void dec_embed_kernel(const float* __restrict__ embed,
                                                        const float* __restrict__ pe,
                                                        const int* __restrict__ tok_row, int V,
                                                        int d, int pos, const int* __restrict__ pos_dev,
                                                        int pe_len, float xscale,
                                                        float* __restrict__ x) {
  const int r = blockIdx.x;
  if (pos_dev) {
    pos = *pos_dev;
    if (pos >= pe_len) return;  // a replayed graph may run past the last useful step
    tok_row += (size_t)pos * gridDim.x;
  }
  int t = tok_row[r];
  t = t < 0 ? 0 : (t >= V ? V - 1 : t);  // rows of ended hypotheses hold stale ids
  const float* e = embed + (size_t)t * d;
  const float* p = pe + (size_t)pos * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x) x[(size_t)r * d + c] = e[c] * xscale + p[c];
}

template <typename T>
__device__ __forceinline__ void load8(const T* __restrict__ p, float o[8]);
template <>
__device__ __forceinline__ void load8<bf16>(const bf16* __restrict__ p, float o[8]) {
  const bf16x8 v = *(const bf16x8*)p;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
}
template <>
__device__ __forceinline__ void load8<float>(const float* __restrict__ p, float o[8]) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
  o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <typename T>
__device__ __forceinline__ void store8(T* __restrict__ p, const float o[8]);
template <>
__device__ __forceinline__ void store8<bf16>(bf16* __restrict__ p, const float o[8]) {
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (bf16)o[e];
  *(bf16x8*)p = v;
}
template <>
__device__ __forceinline__ void store8<float>(float* __restrict__ p, const float o[8]) {
  *(float4*)p = make_float4(o[0], o[1], o[2], o[3]);
  *(float4*)(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
}

template <typename T>
__device__ __forceinline__ typename Mma<T>::frag zero_frag();
template <>
__device__ __forceinline__ bf16x8 zero_frag<bf16>() {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  return __builtin_bit_cast(bf16x8, (u32x4){0u, 0u, 0u, 0u});
}
template <>
__device__ __forceinline__ float zero_frag<float>() { return 0.f; }

// 8 consecutive elements kept in their storage form (16 B for bf16, 32 B for f32) so that many
// row loads can be in flight without spending 8 VGPRs each on converted floats.
template <typename T> struct Raw8;
template <> struct Raw8<bf16> {
  bf16x8 v;
  __device__ __forceinline__ void load(const bf16* p) { v = *(const bf16x8*)p; }
  __device__ __forceinline__ float at(int e) const { return (float)v[e]; }
};
template <> struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *(const float4*)p; b = *(const float4*)(p + 4); }
  __device__ __forceinline__ float at(int e) const {
    return e == 0 ? a.x : e == 1 ? a.y : e == 2 ? a.z : e == 3 ? a.w : e == 4 ? b.x : e == 5 ? b.y : e == 6 ? b.z : b.w;
  }
};

// One wave per (hypothesis row, head); the `group` rows of one utterance (its beam) share a
// workgroup: hypotheses of a beam share almost all of their ancestors, so the same K/V cache rows
// are requested by the waves of one CU (L1 / one L2) instead of by up to `group` CUs on 8 XCDs.
// qkv [n][3d] (q | k | v of the token at position `pos`); kc/vc [Lmax][n][d] (this layer);
// anc [n][Lmax].  ctx [n][d].
// Lane = (position sub-index jsub, 8-channel chunk ch): one wave-wide load instruction covers
// NJ = 64/(DK/8) prefix positions x the whole head (16-byte loads for bf16), so a 250-token prefix
// is 32 position groups.
//
// Round 3: ONE pass with an online softmax.  Until then the kernel ran QK^T over the whole prefix, a softmax through
// LDS and two barriers, and then P.V over the whole prefix again: two sequences of dependent gather round trips (the
// V rows were only requested after the softmax) and a ds_bpermute chain of three per position for the dot products -
// 12.5 us per launch, six launches per label step (profiles/r03f_search_kernel_stats.csv).  Now a batch requests the K
// AND the V rows of its UN position groups together (half the round trips), every position slot jsub keeps its own
// running (max, sum, context) over the positions j = jsub (mod NJ) - no LDS, no barrier - and the NJ partial states are
// merged once at the end.  The dot product's cross-lane sum runs on DPP.  The result differs from the two-pass form by
// f32 rounding of the rescaling only.
//
// Round 3, later: SPLIT waves per (row, head).  A wave's batch - 64 positions, K and V rows of 128 bytes each scattered
// over the cache - is one dependent round trip of ~3.5 us whatever the grid looks like (rows per workgroup 10 -> 1:
// 0.368 -> 0.363 ms per label step, profiles/r03v), and a 190-token prefix was four of them in sequence.  Now the
// batches of a prefix are dealt round-robin to SPLIT waves of the same workgroup, each carrying its own online-softmax
// state, merged through LDS at the end: one round trip up to 64 SPLIT positions.  (Measured, profiles/r03w: with four
// waves per row the launch time no longer depends on the prefix length - 10.7 +- 0.9 us against 5.3 .. 19.4 - but every
// wave brings ~1.5 us of fixed cost, query load to merge; two waves per row is the best trade at 160 rows.)
constexpr int SA_MAXL = 4096;  // Lmax ancestor slots per row must fit the 64 KiB default LDS
constexpr int SA_MERGE_FLOATS = 16 * 8 * 10;  // [wave][channel chunk][m, l, acc[8]]
template <typename T, int DK>
__global__ __launch_bounds__(1024) void dec_self_attn_kernel(const T* __restrict__ qkv,
                                                           T* __restrict__ kc, T* __restrict__ vc,
                                                           const int* __restrict__ anc,
                                                           const int* __restrict__ anc_odd, int n,
                                                           int d, int Lmax, int pos,
                                                           const int* __restrict__ pos_dev,
                                                           const int* __restrict__ tok_tab,
                                                           T* __restrict__ ctx, int SA_SPLIT) {
  // tok_tab != NULL ([Lmax][n] token table): keys whose token id is 0 are masked
  // (TransformerLM._target_mask, espnet2/lm/transformer_lm.py:54-57).
  // pos_dev != NULL (hipGraph-captured search step): position from device memory; the ancestor
  // table is double-buffered by step parity (anc at even steps, anc_odd at odd steps)
  if (pos_dev) {
    pos = *pos_dev;
    if (pos >= Lmax) return;
    if (pos & 1) anc = anc_odd;
  }
  constexpr int NCH = DK / 8;   // lanes per position
  constexpr int NJ = 64 / NCH;  // positions per group
  static_assert(NCH == 8 || NCH == 4, "d_k 64 or 32");
  extern __shared__ int sa_lds[];  // merge buffer, then per row: Lmax ancestor slots
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rloc = wave / SA_SPLIT, slice = wave % SA_SPLIT;  // the row of the workgroup, and which of its batches
  float* const mrg = (float*)sa_lds;
  int* a_s = sa_lds + SA_MERGE_FLOATS + (size_t)rloc * Lmax;
  const int h = blockIdx.x;
  int r = blockIdx.y * ((blockDim.x >> 6) / SA_SPLIT) + rloc;
  const bool live = r < n;  // the last group may be partial
  r = live ? r : n - 1;
  const int ch = lane % NCH, jsub = lane / NCH;
  const T* row = qkv + (size_t)r * 3 * d + h * DK;
  float q[8];
  load8<T>(row + ch * 8, q);
  const float scale = rsqrtf((float)DK);
#pragma unroll
  for (int e = 0; e < 8; ++e) q[e] *= scale;
  // append this position's K/V to the cache (read back by later steps only)
  if (lane < NCH && live && slice == 0) {
    const size_t o = ((size_t)pos * n + r) * d + h * DK + lane * 8;
    float t8[8];
    load8<T>(row + d + lane * 8, t8);
    store8<T>(kc + o, t8);
    load8<T>(row + 2 * d + lane * 8, t8);
    store8<T>(vc + o, t8);
  }
  // ancestor slots of the prefix: one coalesced read into LDS, so the K/V row addresses of the
  // loop below do not hang off a second dependent global load.  (Filled by all the row's waves together, read after the
  // barrier.)
  for (int j = slice * 64 + lane; j < pos; j += 64 * SA_SPLIT) a_s[j] = anc[(size_t)r * Lmax + j];
  __syncthreads();
  const int niter = (pos + NJ) / NJ;  // ceil((pos+1)/NJ)
  // The gathers are latency-bound (one 128-byte row per position, scattered over the cache): UN K rows and UN V rows
  // per lane are in flight before the first use.
  constexpr int UN = sizeof(T) == 2 ? 8 : 4;
  float m_run = -INFINITY, l_run = 0.f;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int* const tt = tok_tab ? tok_tab : anc;
  for (int it0 = slice * UN; it0 < niter; it0 += UN * SA_SPLIT) {
    Raw8<T> k8[UN], v8[UN];
    int tkv[UN];
    // UNCONDITIONAL gathers from a clamped position (positions past the prefix repeat the current row and are masked
    // below): a load under `if (j <= pos)` is followed by s_waitcnt vmcnt(0) - hipcc's wait counting gives up at a
    // branch - and the UN loads "in flight" become UN dependent round trips (tools/isa_waits.py)
    int slot[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {  // the batch's ancestor slots first, back to back (one LDS latency, not UN)
      const int j = (it0 + u) * NJ + jsub;
      slot[u] = a_s[j < pos ? j : 0];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = (it0 + u) * NJ + jsub;
      const bool cur = j >= pos;  // the current token's own row (and the clamp for positions past it)
      const size_t o = ((size_t)(cur ? pos : j) * n + slot[u]) * d + h * DK + ch * 8;
      k8[u].load(cur ? row + d + ch * 8 : kc + o);
      v8[u].load(cur ? row + 2 * d + ch * 8 : vc + o);
      // the key's token id (LM mask), requested with the rows; without a table the load repeats anc[0] and is ignored
      tkv[u] = tt[tok_tab ? (size_t)(cur ? pos : j) * n + (cur ? r : slot[u]) : 0];
    }
    __builtin_amdgcn_sched_barrier(0);  // every request of the batch is out before its first use (left alone hipcc sinks the loads next to their uses)
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = (it0 + u) * NJ + jsub;
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) dot = fmaf(q[e], k8[u].at(e), dot);
      dot += dpp_f32<DPP_XOR1>(dot);
      dot += dpp_f32<DPP_XOR2>(dot);
      if constexpr (NCH == 8) dot += dpp_f32<DPP_HALF_MIRROR>(dot);  // the quads are uniform: lane 7 - i holds the other quad's sum
      const bool valid = (j <= pos) & ((tok_tab == nullptr) | (tkv[u] != 0));  // (bitwise: no branch for hipcc to sink a load into)
      // online softmax of this position slot: (m, l, acc) <- merge with (dot, 1, v)
      const float m_new = valid ? fmaxf(m_run, dot) : m_run;
      const float alpha = (m_run > -INFINITY) ? __expf(m_run - m_new) : 0.f;  // m_new == m_run == -inf: nothing yet
      const float p = valid ? __expf(dot - m_new) : 0.f;
      l_run = l_run * alpha + p;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, v8[u].at(e), acc[e] * alpha);
      m_run = m_new;
    }
  }
  // merge the NJ position slots: rotations inside the 16-lane rows on DPP, then across the rows
  float M = m_run;
  if constexpr (NCH == 4) M = fmaxf(M, dpp_f32<DPP_ROR4>(M));
  M = fmaxf(M, dpp_f32<DPP_ROR8>(M));
  M = fmaxf(M, __shfl_xor(M, 16, 64));
  M = fmaxf(M, __shfl_xor(M, 32, 64));
  const float f = (m_run > -INFINITY) ? __expf(m_run - M) : 0.f;  // all keys masked: every slot is empty -> zeros
  float sum = l_run * f;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] *= f;
  if constexpr (NCH == 4) {
    sum += dpp_f32<DPP_ROR4>(sum);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += dpp_f32<DPP_ROR4>(acc[e]);
  }
  sum += dpp_f32<DPP_ROR8>(sum);
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] += dpp_f32<DPP_ROR8>(acc[e]);
#pragma unroll
  for (int o = 16; o < 64; o <<= 1) {
    sum += __shfl_xor(sum, o, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  }
  // ---- the SPLIT partial states of the row -> its first wave (every lane of a wave holds the wave's totals of its chunk)
  if (lane < NCH) {
    float* o = mrg + (wave * 8 + lane) * 10;
    o[0] = M;
    o[1] = sum;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[2 + e] = acc[e];
  }
  __syncthreads();
  if (lane < NCH && live && slice == 0) {
    float Mx = -INFINITY;
    for (int q = 0; q < SA_SPLIT; ++q) Mx = fmaxf(Mx, mrg[((wave + q) * 8 + lane) * 10]);
    float tot = 0.f, out[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < SA_SPLIT; ++q) {
      const float* o = mrg + ((wave + q) * 8 + lane) * 10;
      const float fq = (o[0] > -INFINITY) ? __expf(o[0] - Mx) : 0.f;  // (a slice without positions, or all of them masked)
      tot += o[1] * fq;
#pragma unroll
      for (int e = 0; e < 8; ++e) out[e] += o[2 + e] * fq;
    }
    const float inv = tot > 0.f ? 1.0f / tot : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] *= inv;
    store8<T>(ctx + (size_t)r * d + h * DK + lane * 8, out);
  }
}

// Source attention of the W hypotheses of one utterance over its encoder memory.
// grid (heads, B, ceil(W/16)), 256 threads.  qs [n][d]; kmem rows at stride ldk (K part of the
// per-layer [B*T][2d] projection); vT [B][d][Tpad] (V transposed, zero padded); klens [B].
//
// Round 3: the kernel's global loads are requested up front.  Before, the QK^T loop fetched one key tile per trip
// (load, wait, MFMA, LDS store: Tpad / 64 dependent round trips per wave) and the P.V loop one V fragment per trip
// (Tpad / 32 more), with a shuffle-chain softmax between them: 10.3 us per launch for 64 KB of operands, six launches
// per label step.  Now a wave requests the key tiles of a batch (UT tiles x KS fragments) AND its first UV fragments
// of V^T - which do not depend on the softmax - in one burst before its first MFMA, the softmax reductions run on DPP,
// and only the rows that exist (W of the 16) are normalised, dealt round-robin to the waves.
//
// Round 4, NVQ > 0: the pre-norm LayerNorm and the query projection (decoder_layer.py:119-121 norm2 -> src_attn's linear_q,
// attention.py:94) run in THIS kernel's prologue - the label step's LayerNorm + q launch (ln_gemm, 4.7 us at the
// single-round-trip floor, six per step) is gone.  A workgroup needs only ITS head's 64 query columns of its 16 rows:
// the workgroup normalises the 16 rows (f32 residual stream x, K = 64 NVQ) into LDS exactly as ln_gemm_kernel does
// (16 lanes per row, DPP statistics), wave w contracts them with rows h DK + 16 w .. + 15 of W_q (fragments straight
// from global memory, requested first), the result (+ bias, rounded to the act dtype like the q tensor it replaces) goes
// through a [16][DK] LDS tile into the A-operand fragments.  Same summation order as ln_gemm: the same q bit for bit.
struct SrcLnQ {
  const float *x, *g, *be, *bq;
  const void* wq;
  float eps;
  int wq_frag;  // round 6: wq is FRAGMENT-MAJOR (espnet_amd.lib.pack_frag16): the wave's 16 x K operand rows are K / 32 contiguous KiB
  // round 6: the memory's K and V^T FRAGMENT-MAJOR (dec_pack_mem_frag_kernel below; both or neither): every operand load of the
  // kernel is then one contiguous KiB.  kf [B][heads][Tpad / 16][DK / 32][lane][8], vf [B][heads][DK / 16][Tpad / 32][lane][8],
  // zero for keys >= T.
  const void *kf, *vf;
};
template <typename T, int DK, int NVQ>
__global__ __launch_bounds__(256) void dec_src_attn_kernel(const T* __restrict__ qs,
                                                           const T* __restrict__ kmem, int ldk,
                                                           const T* __restrict__ vT,
                                                           const int* __restrict__ klens, int W,
                                                           int d, int Tn, int Tpad,
                                                           T* __restrict__ ctx, SrcLnQ lq) {
  using M = Mma<T>;
  constexpr int KS = DK / M::K;
  constexpr int UT = sizeof(T) == 2 ? 4 : 2;  // key tiles per wave and batch (UT * KS operand loads in flight)
  constexpr int UV = 8;                       // V^T fragments per batch
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int LDS_S = Tpad + 4;                       // f32 score row stride
  const int LDS_P = Tpad + 16 / (int)sizeof(T);     // probability row stride (elements)
  float* S = (float*)smem_raw;                      // [16][LDS_S]
  float* sums = S + 16 * LDS_S;                     // [16]
  T* P = (T*)(sums + 16);                           // [16][LDS_P]

  const int h = blockIdx.x, b = blockIdx.y, rg = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int row0 = b * W + rg * 16;
  const int nrows = (W - rg * 16) < 16 ? (W - rg * 16) : 16;
  const int klen = klens[b] < Tn ? klens[b] : Tn;
  const float scale = rsqrtf((float)DK);
  const int ntile = Tpad / 16, nkk = Tpad / M::K;
  const bool pv_wave = wave < DK / 16;

  typename M::frag qf[KS];
  if constexpr (NVQ == 0) {
    const int rr = lr < nrows ? lr : nrows - 1;
    const T* qrow = qs + (size_t)(row0 + rr) * d + h * DK;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = M::load(qrow + ks * M::K + lg * M::EPL);
  }
  const bool mfrag = lq.kf != nullptr;  // (uniform) fragment-major memory: one contiguous KiB per operand load
  const int H = gridDim.x;
  const T* kb = mfrag ? (const T*)lq.kf + ((size_t)(b * H + h) * ntile * KS * 64 + lane) * M::EPL : kmem + (size_t)b * Tn * ldk + h * DK;
  const T* vb = mfrag ? (const T*)lq.vf + (((size_t)(b * H + h) * (DK / 16) + (pv_wave ? wave : 0)) * nkk * 64 + lane) * M::EPL
                      : vT + ((size_t)b * d + h * DK + (pv_wave ? wave : 0) * 16 + lr) * Tpad + lg * M::EPL;
  const size_t vstep = mfrag ? (size_t)64 * M::EPL : (size_t)M::K;  // elements between consecutive 32-key fragments of V^T
  // the first batch of V^T fragments: requested before the scores exist
  typename M::frag vf[UV];
#pragma unroll
  for (int u = 0; u < UV; ++u) vf[u] = M::load(vb + (size_t)(u < nkk ? u : nkk - 1) * vstep);
  typename M::frag kf[UT][KS];
  auto load_k = [&](int nt0) {  // unconditional, from clamped rows / tiles (tiles past the end are computed and dropped)
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const int nt = nt0 + wave + 4 * u;
      const int key = nt * 16 + lr;
      const int kc_ = key < Tn ? key : Tn - 1;
      const int tl = nt < ntile ? nt : ntile - 1;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        kf[u][ks] = M::load(mfrag ? kb + (size_t)(tl * KS + ks) * 64 * M::EPL : kb + (size_t)kc_ * ldk + ks * M::K + lg * M::EPL);
    }
  };
  if constexpr (NVQ > 0) load_k(0);  // the first key tiles do not depend on the queries: under the LayerNorm + projection
  if constexpr (NVQ > 0) {
    static_assert(DK == 64 && sizeof(T) == 2, "fused query projection: bf16, d_k = 64");
    constexpr int K = 64 * NVQ, LDA = K + 8, LDQ = DK + 8, NST = K / M::K;
    T* const sA = (T*)(smem_raw + (size_t)16 * LDS_S * 4 + 64 + (size_t)16 * LDS_P * sizeof(T));  // behind S, sums, P
    T* const sQ = sA + 16 * LDA;
    // W_q rows of this wave's 16 query columns: requested before anything of the LayerNorm
    // (row-major: 16 rows x 64 bytes per wave-wide load; fragment-major: one contiguous KiB - the same values either way)
    const T* wrow = lq.wq_frag ? (const T*)lq.wq + ((size_t)(h * (DK / 16) + wave) * NST * 64 + lane) * M::EPL
                               : (const T*)lq.wq + (size_t)(h * DK + wave * 16 + lr) * K + lg * M::EPL;
    const size_t wstep = lq.wq_frag ? (size_t)64 * M::EPL : (size_t)M::K;
    typename M::frag fw[NST];
#pragma unroll
    for (int u = 0; u < NST; ++u) fw[u] = M::load(wrow + (size_t)u * wstep);
    const float bv = lq.bq[h * DK + wave * 16 + lr];
    {
      const int grp = lane >> 4, li = lane & 15;
      const int rq = wave * 4 + grp;  // the row of the 16 this lane group normalises (rows past nrows: copies of the last)
      const float4* xr = (const float4*)(lq.x + (size_t)(row0 + (rq < nrows ? rq : nrows - 1)) * K);
      float4 v[NVQ], g4[NVQ], b4[NVQ];
#pragma unroll
      for (int j = 0; j < NVQ; ++j) v[j] = xr[j * 16 + li];
#pragma unroll
      for (int j = 0; j < NVQ; ++j) {
        g4[j] = *(const float4*)(lq.g + (j * 16 + li) * 4);
        b4[j] = *(const float4*)(lq.be + (j * 16 + li) * 4);
      }
      asm volatile("" ::: "memory");  // every request of the prologue is out before the first reduction (as ln_gemm.hip)
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NVQ; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
      s += dpp_f32<DPP_XOR1>(s); s += dpp_f32<DPP_XOR2>(s); s += dpp_f32<DPP_HALF_MIRROR>(s); s += dpp_f32<DPP_MIRROR>(s);
      const float mean = s / (float)K;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < NVQ; ++j) {
        const float a0 = v[j].x - mean, a1 = v[j].y - mean, a2 = v[j].z - mean, a3 = v[j].w - mean;
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
      q += dpp_f32<DPP_XOR1>(q); q += dpp_f32<DPP_XOR2>(q); q += dpp_f32<DPP_HALF_MIRROR>(q); q += dpp_f32<DPP_MIRROR>(q);
      const float rstd = 1.0f / sqrtf(q / (float)K + lq.eps);
      T* dst = sA + (size_t)rq * LDA;
#pragma unroll
      for (int j = 0; j < NVQ; ++j) {
        const int c0 = (j * 16 + li) * 4;
        __attribute__((aligned(8))) T o[4];
        o[0] = from_f32<T>((v[j].x - mean) * rstd * g4[j].x + b4[j].x);
        o[1] = from_f32<T>((v[j].y - mean) * rstd * g4[j].y + b4[j].y);
        o[2] = from_f32<T>((v[j].z - mean) * rstd * g4[j].z + b4[j].z);
        o[3] = from_f32<T>((v[j].w - mean) * rstd * g4[j].w + b4[j].w);
        *(uint2*)(dst + c0) = *(const uint2*)o;
      }
    }
    __syncthreads();
    f32x4 qa = (f32x4){0.f, 0.f, 0.f, 0.f};
    const T* a0p = sA + (size_t)lr * LDA + lg * M::EPL;
#pragma unroll
    for (int u = 0; u < NST; ++u) qa = M::mma(M::load(a0p + (size_t)u * M::K), fw[u], qa);
    // C/D layout: col = lr (query column 16 wave + lr), row = lg * 4 + r
#pragma unroll
    for (int r = 0; r < 4; ++r) sQ[(lg * 4 + r) * LDQ + wave * 16 + lr] = from_f32<T>(qa[r] + bv);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = M::load(sQ + lr * LDQ + ks * M::K + lg * M::EPL);
  }
  for (int nt0 = 0; nt0 < ntile; nt0 += 4 * UT) {
    if (nt0 > 0 || NVQ == 0) load_k(nt0);  // (NVQ > 0: the first batch was requested in front of the query prologue)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const int nt = nt0 + wave + 4 * u;
      const int key = nt * 16 + lr;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = M::mma(qf[ks], kf[u][ks], acc);
      if (nt < ntile) {
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(lg * 4 + r) * LDS_S + key] = key < klen ? acc[r] * scale : -INFINITY;
      }
    }
  }
  __syncthreads();
  // softmax numerators of the rows that exist, dealt round-robin to the waves; P holds exp(s - max) in the act dtype
  // and sums[] the sum of the ROUNDED values, so the normalisation below is exact for what PV sees.  Rows past nrows
  // (W < 16) get zero probabilities: their P.V lanes are computed and never stored.
  for (int rr = wave; rr < 16; rr += 4) {
    if (rr >= nrows) {
      for (int j = lane; j < Tpad; j += 64) P[rr * LDS_P + j] = from_f32<T>(0.f);
      continue;
    }
    float mx = -INFINITY;
    for (int j = lane; j < Tpad; j += 64) mx = fmaxf(mx, S[rr * LDS_S + j]);
    mx = wave_allmax_dpp(mx);
    float sum = 0.f;
    for (int j = lane; j < Tpad; j += 64) {
      T pt = from_f32<T>(__expf(S[rr * LDS_S + j] - mx));
      P[rr * LDS_P + j] = pt;
      sum += to_f32(pt);
    }
    sum = wave_allsum_dpp(sum);
    if (lane == 0) sums[rr] = sum;
  }
  __syncthreads();
  if (pv_wave) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kk0 = 0; kk0 < nkk; kk0 += UV) {
      if (kk0 > 0) {
#pragma unroll
        for (int u = 0; u < UV; ++u) vf[u] = M::load(vb + (size_t)(kk0 + u < nkk ? kk0 + u : nkk - 1) * vstep);
      }
#pragma unroll
      for (int u = 0; u < UV; ++u) {
        const int kk = kk0 + u < nkk ? kk0 + u : nkk - 1;
        typename M::frag pf = M::load(P + lr * LDS_P + kk * M::K + lg * M::EPL);
        if (kk0 + u >= nkk) pf = zero_frag<T>();
        acc = M::mma(pf, vf[u], acc);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = lg * 4 + r;
      if (rr < nrows)
        ctx[(size_t)(row0 + rr) * d + h * DK + wave * 16 + lr] = from_f32<T>(acc[r] / sums[rr]);
    }
  }
}

// Round 6: the source-attention memory FRAGMENT-MAJOR (one thread per 16-byte chunk of either output).
//   kf[b][h][tile][ks][lane = lg * 16 + lr][e] = K[b * T + 16 tile + lr][h DK + 32 ks + 8 lg + e]      (K = kv columns 0 .. d - 1)
//   vf[b][h][df][kk][lane = lg * 16 + lr][e]   = V[b * T + 32 kk + 8 lg + e][h DK + 16 df + lr]        (V = kv columns d .. 2 d - 1)
// zero for keys >= T.  Tpad % 32 == 0.
template <typename T>
__global__ __launch_bounds__(256) void dec_pack_mem_frag_kernel(const T* __restrict__ kv, int B, int Tn, int Tpad, int d, int H,
                                                                T* __restrict__ kf, T* __restrict__ vf) {
  const int DK = d / H;
  const size_t nchunk = (size_t)B * d * Tpad / 8;  // 16-byte chunks of one output
  const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= 2 * nchunk) return;
  const bool is_v = id >= nchunk;
  size_t c = is_v ? id - nchunk : id;
  const int lane = (int)(c & 63), lr = lane & 15, lg = lane >> 4;
  c >>= 6;
  typedef __attribute__((ext_vector_type(8))) T vec8;
  vec8 out;
  if (!is_v) {
    const int KS = DK / 32, ntile = Tpad / 16;
    const int ks = (int)(c % KS); c /= KS;
    const int tile = (int)(c % ntile); c /= ntile;
    const int h = (int)(c % H), b = (int)(c / H);
    const int key = tile * 16 + lr;
    const T* src = kv + ((size_t)b * Tn + (key < Tn ? key : 0)) * 2 * d + h * DK + ks * 32 + lg * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = key < Tn ? src[e] : (T)0.f;
    *(vec8*)(kf + id * 8) = out;
  } else {
    const int nkk = Tpad / 32, DF = DK / 16;
    const int kk = (int)(c % nkk); c /= nkk;
    const int df = (int)(c % DF); c /= DF;
    const int h = (int)(c % H), b = (int)(c / H);
    const int col = d + h * DK + df * 16 + lr;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int key = kk * 32 + lg * 8 + e;
      out[e] = key < Tn ? kv[((size_t)b * Tn + key) * 2 * d + col] : (T)0.f;
    }
    *(vec8*)(vf + (id - nchunk) * 8) = out;
  }
}

// vT[b][c][t] = kv[b*T + t][d + c]   (t < T; the padding t in [T, Tpad) stays zero)
template <typename T>
__global__ __launch_bounds__(256) void transpose_v_kernel(const T* __restrict__ kv, int Tn, int d,
                                                          int Tpad, T* __restrict__ vT) {
  __shared__ T tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    int t = t0 + k, c = c0 + tx;
    tile[k][tx] = (t < Tn && c < d) ? kv[((size_t)b * Tn + t) * 2 * d + d + c] : from_f32<T>(0.f);
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    int c = c0 + k, t = t0 + tx;
    if (c < d && t < Tn) vT[((size_t)b * d + c) * Tpad + t] = tile[tx][k];
  }
}

// TransformerLM input: e[r] = embed[tok[r]] in the act dtype (the A operand of the input Linear).
template <typename T>
__global__ __launch_bounds__(128) void lm_embed_kernel(const float* __restrict__ embed,
                                                       const int* __restrict__ tok_row, int V, int eu,
                                                       const int* __restrict__ pos_dev, int Lmax,
                                                       T* __restrict__ e) {
  const int r = blockIdx.x;
  if (pos_dev) {
    const int pos = *pos_dev;
    if (pos >= Lmax) return;
    tok_row += (size_t)pos * gridDim.x;
  }
  int t = tok_row[r];
  t = t < 0 ? 0 : (t >= V ? V - 1 : t);
  for (int c = threadIdx.x; c < eu; c += blockDim.x) e[(size_t)r * eu + c] = from_f32<T>(embed[(size_t)t * eu + c]);
}

// TransformerLM input layer tail (legacy/nets/pytorch_backend/transformer/encoder.py:132-139) on
// x [n][d] f32 in place: torch.nn.LayerNorm(eps 1e-5) -> ReLU -> optional PositionalEncoding
// (x * sqrt(d) + pe[pos]).  One wave per row.
__global__ __launch_bounds__(256) void lm_input_norm_kernel(float* __restrict__ x,
                                                            const float* __restrict__ g,
                                                            const float* __restrict__ b,
                                                            const float* __restrict__ pe, int n, int d,
                                                            int pos, const int* __restrict__ pos_dev,
                                                            int Lmax) {
  if (pos_dev) {
    pos = *pos_dev;
    if (pos >= Lmax) return;
  }
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  float* xr = x + (size_t)row * d;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float t = xr[c] - mean;
    q += t * t;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
  const float xs = sqrtf((float)d);
  for (int c = lane; c < d; c += 64) {
    float v = fmaxf((xr[c] - mean) * rstd * g[c] + b[c], 0.f);
    if (pe) v = v * xs + pe[(size_t)pos * d + c];
    xr[c] = v;
  }
}

// ---- round 5: self-attention over the UNION of a beam's ancestors, on the matrix cores ---------------------------------
// The W hypotheses of an utterance are paths in one token tree: at position j they sit on a handful of distinct cache rows
// (their common prefix on ONE), yet dec_self_attn_kernel above gathers every row's own copy of the prefix - W x pos K and V
// rows per utterance, layer and step, 164 MB per layer at configs[3]'s per-GPU shape (640 rows, 8 heads, position ~125) -
// and at 640 rows it is the label step's largest kernel (6 x 32 us of 0.72 ms, profiles/r05j_search640_kernel_stats.csv).
// Here one workgroup owns (utterance, head): phase 1 finds, per position, the distinct cache rows under the beam (`first
// [u][j]`: the lowest beam row that shares row u's ancestor at j; a key is kept where first[u][j] == u) and compacts them
// into a key list by a block-wide prefix sum; the beam's own new tokens follow as W keys each visible to its own row.
// Phase 2 walks the list in 64-key tiles (a wave per tile): the keys' K rows are gathered ONCE (buffer loads; lanes past
// the list read nothing), S^T[key][row] = K . Q^T runs on MFMA for all W rows together, a score counts where
// first[row][j(key)] == u(key), the online softmax is per lane as in relpos_attn2_kernel (scores transposed: the
// probabilities are already the B operand of O^T = V^T . P^T), and V^T comes out of the tile by 2-byte reads.  The waves'
// partial (m, l, O) meet once through LDS.  Reference: MultiHeadedAttention.forward over the cached prefix
// (transformer/attention.py:121-151 via decoder_layer.py:96-117 with cache), identical for every hypothesis that shares
// a prefix - which is what is exploited.  bf16, d_k = 64, W <= 16, Lmax <= 512 and a key list that fits 64 KiB of LDS (configs[2] / [3]: W = 10, Lmax = 258); other
// shapes keep the per-row kernel.  (Four waves and <= 64 KiB of LDS on purpose: two to three workgroups share a CU and the
// launch's B x heads workgroups - 512 at configs[3]'s per-GPU shape - are all resident at once.  With eight waves, a tile each,
// and 82 KiB the workgroups run in two rounds: 23.4 against 15.5 us per launch, profiles/r05n_tree_self_attention_ab.txt.)
constexpr int TREE_LMAX = 512;  // positions (two per thread of phase 1 at most)
// dynamic LDS: [tile 4 x 8 KiB][kj u16 NK][ks u16 NK][ku u8 NK][first u8 16 x LP][wsum 4 + nk], NK = W Lmax rounded up to 64,
// LP = Lmax rounded up to 64; the launcher refuses shapes beyond 64 KiB (configs[2] / [3]: W = 10, Lmax = 258: 50 KiB)
__host__ __device__ inline int tree_nk(int W, int Lmax) { return (W * Lmax + 63) & ~63; }
__host__ __device__ inline int tree_lp(int Lmax) { return (Lmax + 63) & ~63; }
__host__ __device__ inline size_t tree_lds_bytes(int W, int Lmax) {
  return 32768 + (size_t)tree_nk(W, Lmax) * 5 + 16 * (size_t)tree_lp(Lmax) + 32;
}

template <int WT>  // beam width bound (the key-list pass is a triangle of WT (WT - 1) / 2 comparisons per position)
__global__ __launch_bounds__(256) void dec_self_attn_tree_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ kc,
                                                                  bf16* __restrict__ vc, const int* __restrict__ anc,
                                                                  const int* __restrict__ anc_odd, int n, int d, int Lmax,
                                                                  int pos, const int* __restrict__ pos_dev, int W,
                                                                  bf16* __restrict__ ctx) {
  using MM = Mma<bf16>;
  extern __shared__ __attribute__((aligned(16))) unsigned char tree_lds[];
  if (pos_dev) {
    pos = *pos_dev;
    if (pos >= Lmax) return;
    if (pos & 1) anc = anc_odd;
  }
  const int NK = tree_nk(W, Lmax), LP = tree_lp(Lmax);
  unsigned short* const s_kj = (unsigned short*)(tree_lds + 32768);  // key -> position
  unsigned short* const s_ks = s_kj + NK;                             // key -> cache row (position pos: the beam row's own index)
  unsigned char* const s_ku = (unsigned char*)(s_ks + NK);            // key -> beam row that owns it (first occurrence)
  unsigned char* const s_first = s_ku + NK;                           // [16][LP]
  int* const s_ws = (int*)(s_first + 16 * LP);                        // wsum[4], nk
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int h = blockIdx.x, r0 = blockIdx.y * W;

  // ---- the new token's K / V go into the cache (read back by later steps only)
  if (tid < W * 16) {
    const int u = tid >> 4, which = (tid >> 3) & 1, ch = tid & 7;
    const uint4 v = *(const uint4*)(qkv + (size_t)(r0 + u) * 3 * d + (1 + which) * d + h * 64 + ch * 8);
    *(uint4*)((which ? vc : kc) + ((size_t)pos * n + r0 + u) * d + h * 64 + ch * 8) = v;
  }
  // ---- phase 1: positions tid and tid + 256 of the prefix
  {
    int a[2][WT];
    unsigned own[2] = {0u, 0u};  // bit u: row u owns a key at this position
    int cnt = 0;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const int j = tid + 256 * pp;
      if (j < pos) {
#pragma unroll
        for (int u = 0; u < WT; ++u) a[pp][u] = u < W ? anc[(size_t)(r0 + u) * Lmax + j] : -1 - u;
#pragma unroll
        for (int u = 0; u < WT; ++u) {
          int f = u;
#pragma unroll
          for (int v = WT - 1; v >= 0; --v)
            if (v < u && a[pp][v] == a[pp][u]) f = v;
          s_first[u * LP + j] = (unsigned char)f;
          if (f == u && u < W) {
            own[pp] |= 1u << u;
            ++cnt;
          }
        }
      }
      if (j == pos) {  // the beam's own new tokens: W keys, each visible to its own row
#pragma unroll
        for (int u = 0; u < WT; ++u) s_first[u * LP + j] = (unsigned char)u;
      }
    }
    // block-wide exclusive prefix sum of cnt
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) s_ws[wave] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if (w < wave) base += s_ws[w];
    int k = base + inc - cnt;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int u = 0; u < WT; ++u)
        if (own[pp] & (1u << u)) {
          s_kj[k] = (unsigned short)(tid + 256 * pp);
          s_ks[k] = (unsigned short)a[pp][u];
          s_ku[k] = (unsigned char)u;
          ++k;
        }
    if (tid == 0) {
      const int nkc0 = s_ws[0] + s_ws[1] + s_ws[2] + s_ws[3];
      s_ws[4] = nkc0 + W;
      for (int u = 0; u < W; ++u) {
        s_kj[nkc0 + u] = (unsigned short)pos;
        s_ks[nkc0 + u] = (unsigned short)u;  // (index into the beam's rows of qkv)
        s_ku[nkc0 + u] = (unsigned char)u;
      }
    }
    __syncthreads();
  }
  const int nk = s_ws[4], nkc = nk - W;

  // ---- query fragments (B operand: column = beam row lr, k-slice lg), 1 / sqrt(64) folded in (exact)
  bf16x8 qf[2];
  {
    const int u = lr < W ? lr : W - 1;
    const bf16* qrow = qkv + (size_t)(r0 + u) * 3 * d + h * 64 + lg * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 raw = *(const bf16x8*)(qrow + ks * 32);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[ks][e] = (bf16)((float)raw[e] * 0.125f);
    }
  }
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kc, 0, (int)((size_t)Lmax * n * d * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vc, 0, (int)((size_t)Lmax * n * d * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + (size_t)r0 * 3 * d), 0, W * 3 * d * 2, 0x00020000);
  unsigned char* const tile = tree_lds + wave * 8192;
  f32x4 acc_o[4], acc_l = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < 4; ++f) acc_o[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float row_m = -INFINITY;
  const bf16x8 ones = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};
  constexpr float LOG2E = 1.4426950408889634f;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const int ntile = (nk + 63) >> 6;
  for (int t = wave; t < ntile; t += 4) {
    // ---- gather: key 64 t + 8 i + (lane >> 3), 16-byte chunk lane & 7 of its K and V rows
    u32x4 kr[8], vr[8];
    const int ch = lane & 7;
    const bool has_new = 64 * t + 64 > nkc;  // (uniform) the tile holds keys of the current position: their rows come from qkv
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 64 * t + 8 * i + (lane >> 3);
      const int kk = k < nk ? k : nk - 1;
      const int j = s_kj[kk], sl = s_ks[kk];
      const bool in_cache = k < nkc;
      const unsigned off = in_cache ? (unsigned)((((size_t)j * n + sl) * d + h * 64 + ch * 8) * 2) : 0xffffffffu;
      kr[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, 0);
      vr[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, 0);
      if (has_new) {
        const bool is_new = k >= nkc && k < nk;
        const unsigned qo = is_new ? (unsigned)(((size_t)sl * 3 * d + d + h * 64 + ch * 8) * 2) : 0xffffffffu;
        const unsigned qo2 = is_new ? qo + (unsigned)(d * 2) : 0xffffffffu;
        kr[i] |= __builtin_amdgcn_raw_buffer_load_b128(rq, qo, 0, 0);
        vr[i] |= __builtin_amdgcn_raw_buffer_load_b128(rq, qo2, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int key = 8 * i + (lane >> 3);
      *(u32x4*)(tile + key * 128 + ((ch ^ (key & 7)) << 4)) = kr[i];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- S^T (64 keys x 16 rows)
    f32x4 sc[4];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) sc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        const int row = 16 * nf + lr;
        const bf16x8 kf = *(const bf16x8*)(tile + row * 128 + (((ks * 4 + lg) ^ (row & 7)) << 4));
        sc[nf] = MM::mma(kf, qf[ks], sc[nf]);
      }
    // ---- which of this lane's 16 keys (16 nf + 4 lg + r) its row lr sees
    float tm = -INFINITY;
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      const int k0 = 64 * t + 16 * nf + 4 * lg;
      const uint2 jj = *(const uint2*)(s_kj + k0);
      const unsigned uu = *(const unsigned*)(s_ku + k0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int j = (int)(((r & 2) ? jj.y : jj.x) >> ((r & 1) * 16)) & 0xffff;
        j = j < LP ? j : LP - 1;  // (slots past the list hold whatever: keep the look-up inside the table)
        const int u = (int)(uu >> (8 * r)) & 0xff;
        const bool ok = (k0 + r < nk) & (lr < W) & ((int)s_first[lr * LP + j] == u);
        sc[nf][r] = ok ? sc[nf][r] : -INFINITY;
        tm = fmaxf(tm, sc[nf][r]);
      }
    }
    tm = wave_xor16_max(tm);
    tm = wave_xor32_max(tm);
    const float mn = fmaxf(row_m, tm);
    // (a row beyond the beam, or a tile without a key of this row: everything is -inf; keep the state untouched)
    const bool any = mn > -INFINITY;
    const float alpha = any ? __expf(row_m - mn) : 1.f;
    const float mnl = any ? mn * LOG2E : 0.f;
    row_m = mn;
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
    bf16x8 pb[2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) pb[jp] = __builtin_bit_cast(bf16x8, (u32x4){pbu[jp][0], pbu[jp][1], pbu[jp][2], pbu[jp][3]});
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      acc_o[f][0] *= alpha; acc_o[f][1] *= alpha; acc_o[f][2] *= alpha; acc_o[f][3] *= alpha;
    }
    acc_l[0] *= alpha;
    // ---- V rows over the K rows (this wave's reads of them are done: LDS operations of a wave execute in order)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int key = 8 * i + (lane >> 3);
      *(u32x4*)(tile + key * 128 + ((ch ^ (key & 7)) << 4)) = vr[i];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- O^T += V^T . P^T: the A operand of (dk fragment f, 32-key step jp) is V[key][16 f + lr] for the lane group's key
    // slots 32 jp + 4 lg + e and 32 jp + 16 + 4 lg + e, e < 4 (the order the probabilities sit in pb)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      acc_l = MM::mma(ones, pb[jp], acc_l);
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int dk = 16 * f + lr;
        unsigned w4[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          const int ka = 32 * jp + 16 * (e2 >> 1) + 4 * lg + 2 * (e2 & 1);
          const unsigned lo = *(const unsigned short*)(tile + ka * 128 + ((((dk >> 3) ^ (ka & 7))) << 4) + (dk & 7) * 2);
          const unsigned hi = *(const unsigned short*)(tile + (ka + 1) * 128 + ((((dk >> 3) ^ ((ka + 1) & 7))) << 4) + (dk & 7) * 2);
          w4[e2] = lo | (hi << 16);
        }
        const bf16x8 vf = __builtin_bit_cast(bf16x8, (u32x4){w4[0], w4[1], w4[2], w4[3]});
        acc_o[f] = MM::mma(vf, pb[jp], acc_o[f]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // ---- the waves' partial states meet in wave 0 (lane (lr, lg): row lr, channels 16 f + 4 lg + r)
  __syncthreads();
  float* const mg = (float*)tree_lds;  // [3 waves][18][64 lanes]: the tiles are free by now
  if (wave > 0) {
    float* o = mg + (wave - 1) * 18 * 64 + lane;
    o[0] = row_m;
    o[64] = acc_l[0];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[(2 + f * 4 + r) * 64] = acc_o[f][r];
  }
  __syncthreads();
  if (wave == 0) {
    float M = row_m;
#pragma unroll
    for (int w = 0; w < 3; ++w) M = fmaxf(M, mg[w * 18 * 64 + lane]);
    const float a0 = row_m > -INFINITY ? __expf(row_m - M) : 0.f;
    float l = acc_l[0] * a0;
    float o[16];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[f * 4 + r] = acc_o[f][r] * a0;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      const float* src = mg + w * 18 * 64 + lane;
      const float mw = src[0];
      const float aw = mw > -INFINITY ? __expf(mw - M) : 0.f;
      l += src[64] * aw;
#pragma unroll
      for (int q = 0; q < 16; ++q) o[q] += src[(2 + q) * 64] * aw;
    }
    if (lr < W) {
      const float inv = l > 0.f ? 1.0f / l : 0.f;
      bf16* dst = ctx + (size_t)(r0 + lr) * d + h * 64 + 4 * lg;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const bf16x4 pk = {(bf16)(o[f * 4] * inv), (bf16)(o[f * 4 + 1] * inv), (bf16)(o[f * 4 + 2] * inv), (bf16)(o[f * 4 + 3] * inv)};
        *(bf16x4*)(dst + 16 * f) = pk;
      }
    }
  }
}

template <typename T>
int self_attn_launch(const void* qkv, void* kc, void* vc, const int* anc, const int* anc_odd, int n,
                     int d, int heads, int Lmax, int pos, const int* pos_dev, int group,
                     const int* tok_tab, void* ctx, hipStream_t s) {
  const int dk = d / heads;
  // waves per row: batches of 64 positions run in parallel instead of in sequence once a prefix is longer than that
  const int fsplit = em_sw().sa_split;
  // (only while the launch is small: at 640 rows x 8 heads the second wave per row costs more than the shorter chains
  // give back - 0.704 -> 0.769 ms per label step, profiles/r03af - and the rows themselves fill the chip)
  const int SA_SPLIT = fsplit > 0 ? (fsplit > 8 ? 8 : fsplit) : (heads * n <= 2048 ? 2 : 1);  // (two: 0.369 -> 0.358 ms per label step; four 0.363, eight 0.392 - every wave has its fixed cost; profiles/r03x)
  group = group < 1 ? 1 : (group > 16 / SA_SPLIT ? 16 / SA_SPLIT : group);  // 16 waves per workgroup
  // (round 5: at most four rows - four waves, one per SIMD - per workgroup when every row is one wave: configs[3]'s per-GPU
  // step, 640 rows, ran its half-beam groups of five at 0.715 ms per label step and runs groups of four at 0.651;
  // groups of 1 / 2 / 3: 0.658 / 0.667 / 0.671; profiles/r05l_label_step_dispatch_sweep.txt)
  if (SA_SPLIT == 1 && group > 4) group = 4;
  while (group > 1 && (size_t)(group * Lmax + SA_MERGE_FLOATS) * sizeof(int) > 64 * 1024) --group;  // default LDS limit
  // Rows of a beam on one CU share their ancestors' cache rows in its L1, but three quarters of the chip idle with
  // 160 rows x 4 heads in 64 workgroups: fewer rows per workgroup until the grid covers the CUs (the L2 still serves
  // the shared rows once).  ESPNET_AMD_SA_GROUP: developer A/B.
  const int forced = em_sw().sa_group;
  if (forced > 0) group = forced < group ? forced : group;
  else while (group > 1 && heads * em_cdiv(n, group) < 512) --group;
  if ((size_t)(group * Lmax + SA_MERGE_FLOATS) * sizeof(int) > 64 * 1024) return EM_ERR_UNSUPPORTED;
  dim3 grid(heads, em_cdiv(n, group)), block(64 * group * SA_SPLIT);
  const size_t lds = (size_t)(group * Lmax + SA_MERGE_FLOATS) * sizeof(int);
  if (dk == 64)
    hipLaunchKernelGGL((dec_self_attn_kernel<T, 64>), grid, block, lds, s, (const T*)qkv, (T*)kc,
                       (T*)vc, anc, anc_odd, n, d, Lmax, pos, pos_dev, tok_tab, (T*)ctx, SA_SPLIT);
  else if (dk == 32)
    hipLaunchKernelGGL((dec_self_attn_kernel<T, 32>), grid, block, lds, s, (const T*)qkv, (T*)kc,
                       (T*)vc, anc, anc_odd, n, d, Lmax, pos, pos_dev, tok_tab, (T*)ctx, SA_SPLIT);
  else
    return EM_ERR_UNSUPPORTED;
  EM_CHECK_LAUNCH();
  return EM_OK;
}

template <typename T>
int src_attn_launch(const void* qs, const void* kmem, int ldk, const void* vT, const int* klens,
                    int B, int W, int d, int heads, int Tn, int Tpad, void* ctx, hipStream_t s, const SrcLnQ* lq = nullptr) {
  const int dk = d / heads;
  size_t lds = (size_t)16 * (Tpad + 4) * 4 + 64 + (size_t)16 * (Tpad + 16 / sizeof(T)) * sizeof(T);
  if (lq) {  // fused LayerNorm + query projection (bf16, d_k = 64, d = 256 or 512)
    if constexpr (sizeof(T) == 2) {
      if (dk != 64 || (d != 256 && d != 512)) return EM_ERR_UNSUPPORTED;
      lds += (size_t)16 * (d + 8) * 2 + 16 * (64 + 8) * 2;
      if (lds > 160 * 1024) return EM_ERR_UNSUPPORTED;
      dim3 grid(heads, B, em_cdiv(W, 16));
      static EmLdsCap capq4 = {}, capq8 = {};
      if (d == 256) {
        if (em_raise_lds_cap((const void*)dec_src_attn_kernel<T, 64, 4>, lds, &capq4) != EM_OK) return EM_ERR_LAUNCH;
        hipLaunchKernelGGL((dec_src_attn_kernel<T, 64, 4>), grid, dim3(256), lds, s, (const T*)nullptr, (const T*)kmem, ldk,
                           (const T*)vT, klens, W, d, Tn, Tpad, (T*)ctx, *lq);
      } else {
        if (em_raise_lds_cap((const void*)dec_src_attn_kernel<T, 64, 8>, lds, &capq8) != EM_OK) return EM_ERR_LAUNCH;
        hipLaunchKernelGGL((dec_src_attn_kernel<T, 64, 8>), grid, dim3(256), lds, s, (const T*)nullptr, (const T*)kmem, ldk,
                           (const T*)vT, klens, W, d, Tn, Tpad, (T*)ctx, *lq);
      }
      EM_CHECK_LAUNCH();
      return EM_OK;
    } else {
      return EM_ERR_UNSUPPORTED;
    }
  }
  // the 16 score rows of a workgroup live in LDS: memories beyond ~1 690 frames (bf16; ~1 270 in f32) do
  // not fit the 160 KB of a CU -- say so instead of failing at launch
  if (lds > 160 * 1024) return EM_ERR_UNSUPPORTED;
  dim3 grid(heads, B, em_cdiv(W, 16));
  // (the attribute is raised once per device and size: no runtime API call on later launches,
  //  which also keeps the launch legal inside a stream capture)
  static EmLdsCap cap64 = {}, cap32 = {};
  if (dk == 64) {
    if (em_raise_lds_cap((const void*)dec_src_attn_kernel<T, 64, 0>, lds, &cap64) != EM_OK) return EM_ERR_LAUNCH;
    hipLaunchKernelGGL((dec_src_attn_kernel<T, 64, 0>), grid, dim3(256), lds, s, (const T*)qs,
                       (const T*)kmem, ldk, (const T*)vT, klens, W, d, Tn, Tpad, (T*)ctx, SrcLnQ{});
  } else if (dk == 32) {
    if (em_raise_lds_cap((const void*)dec_src_attn_kernel<T, 32, 0>, lds, &cap32) != EM_OK) return EM_ERR_LAUNCH;
    hipLaunchKernelGGL((dec_src_attn_kernel<T, 32, 0>), grid, dim3(256), lds, s, (const T*)qs,
                       (const T*)kmem, ldk, (const T*)vT, klens, W, d, Tn, Tpad, (T*)ctx, SrcLnQ{});
  } else {
    return EM_ERR_UNSUPPORTED;
  }
  EM_CHECK_LAUNCH();
  return EM_OK;
}

}  // namespace

// the beam-tree form of em_dec_self_attention for the offline search's decoder step (bf16, d_k = 64, rows = B x W with
// W <= 16 consecutive rows per utterance, Lmax <= 512, no token mask); EM_ERR_UNSUPPORTED otherwise (the caller keeps the
// per-row kernel).  ESPNET_AMD_NO_SA_TREE=1: developer A/B switch.
int em_dec_self_attention_tree_bf16(const void* qkv, void* kc, void* vc, const int* anc, const int* anc_odd, int n, int d,
                                    int heads, int Lmax, int pos, const int* pos_dev, int W, void* ctx, void* stream) {
  const bool off = em_sw().no_sa_tree;
  if (off || heads <= 0 || d != 64 * heads || W < 1 || W > 16 || n % W != 0 || n > 65535 || Lmax > TREE_LMAX || pos < 0 ||
      pos >= Lmax)
    return EM_ERR_UNSUPPORTED;
  // one workgroup per (utterance, head) walks the whole prefix: worth it once the per-row kernel's n x heads waves are several
  // rounds of the chip - configs[3] per GPU (640 rows): 0.620 -> 0.593 ms per label step; 480 / 320 / 240 rows: 0.578 -> 0.557,
  // 0.499 -> 0.492, 0.417 -> 0.410; configs[2] (160 rows): 0.352 -> 0.367, the per-row kernel stays (profiles/
  // r05n_tree_self_attention_ab.txt, r05l_label_step_dispatch_sweep.txt).  ESPNET_AMD_SA_TREE_MIN_ROWS: developer switch.
  const int min_rows = em_sw().sa_tree_min_rows;  // (the kernel tests lower it in-process: em_dev_switches_reload)
  if (n < min_rows) return EM_ERR_UNSUPPORTED;
  if ((size_t)Lmax * n * d * 2 >= ((size_t)1 << 31)) return EM_ERR_UNSUPPORTED;  // (buffer offsets are 32-bit)
  const size_t lds = tree_lds_bytes(W, Lmax);
  if (lds > 64 * 1024) return EM_ERR_UNSUPPORTED;  // (the default dynamic-LDS limit: no attribute call, legal inside a stream capture)
  dim3 grid(heads, n / W);
#define EM_TREE_LAUNCH(WT)                                                                                                  \
  hipLaunchKernelGGL(dec_self_attn_tree_kernel<WT>, grid, dim3(256), lds, (hipStream_t)stream, (const bf16*)qkv, (bf16*)kc, \
                     (bf16*)vc, anc, anc_odd, n, d, Lmax, pos, pos_dev, W, (bf16*)ctx)
  if (W <= 5) EM_TREE_LAUNCH(5);
  else if (W <= 10) EM_TREE_LAUNCH(10);
  else EM_TREE_LAUNCH(16);
#undef EM_TREE_LAUNCH
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_dec_embed_f32(const float* embed, const float* pe, const int32_t* tok_row,
                                int32_t n, int32_t V, int32_t d, int32_t pos,
                                const int32_t* pos_dev, int32_t pe_len, float* x, void* stream) {
  if (n <= 0 || d <= 0 || pos < 0) return EM_ERR_BAD_ARG;
  hipLaunchKernelGGL(dec_embed_kernel, dim3(n), dim3(128), 0, (hipStream_t)stream, embed, pe,
                     tok_row, V, d, pos, pos_dev, pe_len, sqrtf((float)d), x);
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_dec_self_attention(int dtype, const void* qkv, void* kc, void* vc,
                                     const int32_t* anc, const int32_t* anc_odd, int32_t n,
                                     int32_t d, int32_t heads, int32_t Lmax, int32_t pos,
                                     const int32_t* pos_dev, int32_t group, const int32_t* tok_tab,
                                     void* ctx, void* stream) {
  if (n <= 0 || heads <= 0 || pos < 0 || pos >= Lmax || Lmax > SA_MAXL) return EM_ERR_BAD_ARG;
  if (dtype == EM_F32)
    return self_attn_launch<float>(qkv, kc, vc, anc, anc_odd, n, d, heads, Lmax, pos, pos_dev, group, tok_tab, ctx, (hipStream_t)stream);
  if (dtype == EM_BF16)
    return self_attn_launch<bf16>(qkv, kc, vc, anc, anc_odd, n, d, heads, Lmax, pos, pos_dev, group, tok_tab, ctx, (hipStream_t)stream);
  return EM_ERR_BAD_ARG;
}

extern "C" int em_dec_self_attention_beam(int dtype, const void* qkv, void* kc, void* vc, const int32_t* anc,
                                          const int32_t* anc_odd, int32_t n, int32_t d, int32_t heads, int32_t Lmax,
                                          int32_t pos, const int32_t* pos_dev, int32_t W, void* ctx, void* stream) {
  if (n <= 0 || heads <= 0 || W <= 0 || n % W != 0 || pos < 0 || pos >= Lmax || Lmax > SA_MAXL) return EM_ERR_BAD_ARG;
  int rc = EM_ERR_UNSUPPORTED;
  if (dtype == EM_BF16 && W > 1)
    rc = em_dec_self_attention_tree_bf16(qkv, kc, vc, anc, anc_odd, n, d, heads, Lmax, pos, pos_dev, W, ctx, stream);
  if (rc == EM_ERR_UNSUPPORTED)
    rc = em_dec_self_attention(dtype, qkv, kc, vc, anc, anc_odd, n, d, heads, Lmax, pos, pos_dev, (W + 1) / 2, nullptr, ctx,
                               stream);
  return rc;
}

extern "C" int em_dec_src_attention(int dtype, const void* qs, const void* kmem, int32_t ldk,
                                    const void* vT, const int32_t* klens, int32_t B, int32_t W,
                                    int32_t d, int32_t heads, int32_t T, int32_t Tpad, void* ctx,
                                    void* stream) {
  if (B <= 0 || W <= 0 || T <= 0 || Tpad < T || Tpad % 32 != 0) return EM_ERR_BAD_ARG;
  if (dtype == EM_F32)
    return src_attn_launch<float>(qs, kmem, ldk, vT, klens, B, W, d, heads, T, Tpad, ctx, (hipStream_t)stream);
  if (dtype == EM_BF16)
    return src_attn_launch<bf16>(qs, kmem, ldk, vT, klens, B, W, d, heads, T, Tpad, ctx, (hipStream_t)stream);
  return EM_ERR_BAD_ARG;
}

extern "C" int em_dec_src_attention_lnq(int dtype, const float* x, const float* g, const float* be, float eps,
                                        const void* wq, const float* bq, const void* kmem, int32_t ldk, const void* vT,
                                        const int32_t* klens, int32_t B, int32_t W, int32_t d, int32_t heads, int32_t T,
                                        int32_t Tpad, void* ctx, void* stream) {
  if (!x || !g || !be || !wq || !bq || !kmem || !vT || !klens || !ctx) return EM_ERR_BAD_ARG;
  if (B <= 0 || W <= 0 || T <= 0 || Tpad < T || Tpad % 32 != 0 || heads <= 0) return EM_ERR_BAD_ARG;
  if (dtype != EM_BF16) return EM_ERR_UNSUPPORTED;  // (the f32 parity mode keeps ln_gemm + em_dec_src_attention)
  const SrcLnQ lq = {x, g, be, bq, wq, eps, 0, nullptr, nullptr};
  return src_attn_launch<bf16>(nullptr, kmem, ldk, vT, klens, B, W, d, heads, T, Tpad, ctx, (hipStream_t)stream, &lq);
}

// ... with wq FRAGMENT-MAJOR (round 6; EmDecoderLayer.src_wq_frag): the same arithmetic in the same order - bit for bit the
// context of em_dec_src_attention_lnq - with the query projection's operand rows read as contiguous KiB.
extern "C" int em_dec_src_attention_lnq_frag(int dtype, const float* x, const float* g, const float* be, float eps,
                                             const void* wq_frag, const float* bq, const void* kmem, int32_t ldk, const void* vT,
                                             const int32_t* klens, int32_t B, int32_t W, int32_t d, int32_t heads, int32_t T,
                                             int32_t Tpad, void* ctx, void* stream) {
  if (!x || !g || !be || !wq_frag || !bq || !kmem || !vT || !klens || !ctx) return EM_ERR_BAD_ARG;
  if (B <= 0 || W <= 0 || T <= 0 || Tpad < T || Tpad % 32 != 0 || heads <= 0) return EM_ERR_BAD_ARG;
  if (dtype != EM_BF16) return EM_ERR_UNSUPPORTED;
  const SrcLnQ lq = {x, g, be, bq, wq_frag, eps, 1, nullptr, nullptr};
  return src_attn_launch<bf16>(nullptr, kmem, ldk, vT, klens, B, W, d, heads, T, Tpad, ctx, (hipStream_t)stream, &lq);
}

// csrc/search.hip (em_common.h): the memory of one decoder layer fragment-major, and the source attention of a label step on
// it (bf16, d_k = 64, d = 256 | 512; wq fragment-major as well).  The same arithmetic in the same order as
// em_dec_src_attention_lnq: bit for bit its context.
int em_dec_pack_memory_frag_bf16(const void* kv, int B, int T, int Tpad, int d, int heads, void* kf, void* vf, void* stream) {
  if (!kv || !kf || !vf || B <= 0 || T <= 0 || Tpad < T || Tpad % 32 != 0 || heads <= 0 || d % heads != 0 || (d / heads) % 32 != 0)
    return EM_ERR_BAD_ARG;
  const size_t nchunk = (size_t)B * d * Tpad / 8;
  hipLaunchKernelGGL(dec_pack_mem_frag_kernel<bf16>, dim3((unsigned)((2 * nchunk + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)kv, B, T, Tpad, d, heads, (bf16*)kf, (bf16*)vf);
  EM_CHECK_LAUNCH();
  return EM_OK;
}
int em_dec_src_attention_lnq_memfrag_bf16(const float* x, const float* g, const float* be, float eps, const void* wq_frag,
                                          const float* bq, const void* kf, const void* vf, const int32_t* klens, int B, int W, int d,
                                          int heads, int T, int Tpad, void* ctx, void* stream) {
  if (!x || !g || !be || !wq_frag || !bq || !kf || !vf || !klens || !ctx) return EM_ERR_BAD_ARG;
  if (B <= 0 || W <= 0 || T <= 0 || Tpad < T || Tpad % 32 != 0 || heads <= 0) return EM_ERR_BAD_ARG;
  const SrcLnQ lq = {x, g, be, bq, wq_frag, eps, 1, kf, vf};
  return src_attn_launch<bf16>(nullptr, nullptr, 0, nullptr, klens, B, W, d, heads, T, Tpad, ctx, (hipStream_t)stream, &lq);
}

extern "C" int em_dec_transpose_v(int dtype, const void* kv, int32_t B, int32_t T, int32_t d,
                                  int32_t Tpad, void* vT, void* stream) {
  if (B <= 0 || T <= 0 || d <= 0 || Tpad < T) return EM_ERR_BAD_ARG;
  dim3 grid(em_cdiv(T, 32), em_cdiv(d, 32), B);
  if (dtype == EM_F32)
    hipLaunchKernelGGL(transpose_v_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const float*)kv, T, d, Tpad, (float*)vT);
  else if (dtype == EM_BF16)
    hipLaunchKernelGGL(transpose_v_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)kv, T, d, Tpad, (bf16*)vT);
  else
    return EM_ERR_BAD_ARG;
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_lm_embed(int dtype, const float* embed, const int32_t* tok_row, int32_t n, int32_t V,
                           int32_t embed_unit, const int32_t* pos_dev, int32_t Lmax, void* e,
                           void* stream) {
  if (!embed || !tok_row || !e || n <= 0 || embed_unit <= 0) return EM_ERR_BAD_ARG;
  if (dtype == EM_F32)
    hipLaunchKernelGGL(lm_embed_kernel<float>, dim3(n), dim3(128), 0, (hipStream_t)stream, embed, tok_row,
                       V, embed_unit, pos_dev, Lmax, (float*)e);
  else if (dtype == EM_BF16)
    hipLaunchKernelGGL(lm_embed_kernel<bf16>, dim3(n), dim3(128), 0, (hipStream_t)stream, embed, tok_row,
                       V, embed_unit, pos_dev, Lmax, (bf16*)e);
  else
    return EM_ERR_BAD_ARG;
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" int em_lm_input_norm_f32(float* x, const float* g, const float* b, const float* pe,
                                    int32_t n, int32_t d, int32_t pos, const int32_t* pos_dev,
                                    int32_t Lmax, void* stream) {
  if (!x || !g || !b || n <= 0 || d <= 0) return EM_ERR_BAD_ARG;
  hipLaunchKernelGGL(lm_input_norm_kernel, dim3(em_cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, x, g,
                     b, pe, n, d, pos, pos_dev, Lmax);
  EM_CHECK_LAUNCH();
  return EM_OK;
}
