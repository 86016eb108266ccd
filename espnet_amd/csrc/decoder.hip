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
#include <string.h>

#include <mutex>
#include <unordered_map>
#include <vector>

#include "em_common.h"

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
// is 32 iterations for QK^T and 32 for PV; partial dots are reduced across the chunk lanes,
// partial contexts across the position lanes.
constexpr int SA_MAXL = 4096;  // Lmax scores + Lmax ancestor slots per wave must fit the 64 KiB default LDS
template <typename T, int DK>
__global__ __launch_bounds__(1024) void dec_self_attn_kernel(const T* __restrict__ qkv,
                                                           T* __restrict__ kc, T* __restrict__ vc,
                                                           const int* __restrict__ anc,
                                                           const int* __restrict__ anc_odd, int n,
                                                           int d, int Lmax, int pos,
                                                           const int* __restrict__ pos_dev,
                                                           const int* __restrict__ tok_tab,
                                                           T* __restrict__ ctx) {
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
  constexpr int NJ = 64 / NCH;  // positions per iteration
  extern __shared__ float sa_lds[];  // per wave: Lmax scores + Lmax ancestor slots
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* p_s = sa_lds + (size_t)wave * 2 * Lmax;
  int* a_s = (int*)(p_s + Lmax);
  const int h = blockIdx.x;
  int r = blockIdx.y * (blockDim.x >> 6) + wave;
  const bool live = r < n;  // the last group may be partial; keep every wave in the barriers
  r = live ? r : n - 1;
  const int ch = lane % NCH, jsub = lane / NCH;
  const T* row = qkv + (size_t)r * 3 * d + h * DK;
  float q[8];
  load8<T>(row + ch * 8, q);
  // append this position's K/V to the cache (read back by later steps only)
  if (lane < NCH && live) {
    const size_t o = ((size_t)pos * n + r) * d + h * DK + lane * 8;
    float t8[8];
    load8<T>(row + d + lane * 8, t8);
    store8<T>(kc + o, t8);
    load8<T>(row + 2 * d + lane * 8, t8);
    store8<T>(vc + o, t8);
  }
  const float scale = rsqrtf((float)DK);
  // ancestor slots of the prefix: one coalesced read into LDS, so the K/V row addresses of the
  // loops below do not hang off a second dependent global load
  for (int j = lane; j < pos; j += 64) a_s[j] = anc[(size_t)r * Lmax + j];
  __syncthreads();
  const int niter = (pos + NJ) / NJ;  // ceil((pos+1)/NJ)
  // The gathers are latency-bound (one 128-byte row per position, scattered over the cache): keep
  // UN row loads per lane in flight before the first use.
  constexpr int UN = sizeof(T) == 2 ? 8 : 4;
  for (int it0 = 0; it0 < niter; it0 += UN) {
    Raw8<T> k8[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = (it0 + u) * NJ + jsub;
      if (j <= pos) {
        const T* kr = (j == pos) ? row + d : kc + ((size_t)j * n + a_s[j]) * d + h * DK;
        k8[u].load(kr + ch * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = (it0 + u) * NJ + jsub;
      float dot = 0.f;
      if (j <= pos) {
#pragma unroll
        for (int e = 0; e < 8; ++e) dot = fmaf(q[e], k8[u].at(e), dot);
      }
#pragma unroll
      for (int o = 1; o < NCH; o <<= 1) dot += __shfl_xor(dot, o, 64);
      if (ch == 0 && j <= pos) {
        const bool masked = tok_tab && tok_tab[(size_t)j * n + (j == pos ? r : a_s[j])] == 0;
        p_s[j] = masked ? -INFINITY : dot * scale;
      }
    }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int j = lane; j <= pos; j += 64) mx = fmaxf(mx, p_s[j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j <= pos; j += 64) {
    const float p = mx > -INFINITY ? expf(p_s[j] - mx) : 0.f;  // all keys masked -> zeros
    p_s[j] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  __syncthreads();
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int it0 = 0; it0 < niter; it0 += UN) {
    Raw8<T> v8[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = (it0 + u) * NJ + jsub;
      if (j <= pos) {
        const T* vr = (j == pos) ? row + 2 * d : vc + ((size_t)j * n + a_s[j]) * d + h * DK;
        v8[u].load(vr + ch * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = (it0 + u) * NJ + jsub;
      if (j <= pos) {
        const float p = p_s[j];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, v8[u].at(e), acc[e]);
      }
    }
  }
#pragma unroll
  for (int o = NCH; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  if (lane < NCH && live) {
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= inv;
    store8<T>(ctx + (size_t)r * d + h * DK + lane * 8, acc);
  }
}

// Source attention of the W hypotheses of one utterance over its encoder memory.
// grid (heads, B, ceil(W/16)), 256 threads.  qs [n][d]; kmem rows at stride ldk (K part of the
// per-layer [B*T][2d] projection); vT [B][d][Tpad] (V transposed, zero padded); klens [B].
template <typename T, int DK>
__global__ __launch_bounds__(256) void dec_src_attn_kernel(const T* __restrict__ qs,
                                                           const T* __restrict__ kmem, int ldk,
                                                           const T* __restrict__ vT,
                                                           const int* __restrict__ klens, int W,
                                                           int d, int Tn, int Tpad,
                                                           T* __restrict__ ctx) {
  using M = Mma<T>;
  constexpr int KS = DK / M::K;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int LDS_S = Tpad + 4;                       // f32 score row stride
  const int LDS_P = Tpad + 16 / (int)sizeof(T);     // probability row stride (elements)
  float* S = (float*)smem_raw;                      // [16][LDS_S]
  float* sums = S + 16 * LDS_S;                     // [16]
  T* P = (T*)(sums + 16);                           // [16][LDS_P]

  const int h = blockIdx.x, b = blockIdx.y, rg = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int row0 = b * W + rg * 16;
  const int nrows = (W - rg * 16) < 16 ? (W - rg * 16) : 16;
  const int klen = klens[b] < Tn ? klens[b] : Tn;
  const float scale = rsqrtf((float)DK);

  typename M::frag qf[KS];
  {
    const int rr = lr < nrows ? lr : nrows - 1;
    const T* qrow = qs + (size_t)(row0 + rr) * d + h * DK;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = M::load(qrow + ks * M::K + lg * M::EPL);
  }
  const T* kb = kmem + (size_t)b * Tn * ldk + h * DK;
  for (int nt = wave; nt < Tpad / 16; nt += 4) {
    int key = nt * 16 + lr;
    const int kc_ = key < Tn ? key : Tn - 1;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      acc = M::mma(qf[ks], M::load(kb + (size_t)kc_ * ldk + ks * M::K + lg * M::EPL), acc);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      S[(lg * 4 + r) * LDS_S + key] = key < klen ? acc[r] * scale : -INFINITY;
  }
  __syncthreads();
  // softmax numerators per row (4 rows per wave); P holds exp(s - max) in the act dtype and
  // sums[] the sum of the ROUNDED values, so the normalisation below is exact for what PV sees
  for (int rr = wave * 4; rr < wave * 4 + 4; ++rr) {
    float mx = -INFINITY;
    for (int j = lane; j < Tpad; j += 64) mx = fmaxf(mx, S[rr * LDS_S + j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < Tpad; j += 64) {
      T pt = from_f32<T>(expf(S[rr * LDS_S + j] - mx));
      P[rr * LDS_P + j] = pt;
      sum += to_f32(pt);
    }
    sum = wave_sum(sum);
    if (lane == 0) sums[rr] = sum;
  }
  __syncthreads();
  if (wave < DK / 16) {
    const T* vb = vT + ((size_t)b * d + h * DK + wave * 16 + lr) * Tpad;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kk = 0; kk < Tpad / M::K; ++kk) {
      const int ko = kk * M::K + lg * M::EPL;
      acc = M::mma(M::load(P + lr * LDS_P + ko), M::load(vb + ko), acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = lg * 4 + r;
      if (rr < nrows)
        ctx[(size_t)(row0 + rr) * d + h * DK + wave * 16 + lr] = from_f32<T>(acc[r] / sums[rr]);
    }
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

template <typename T>
int self_attn_launch(const void* qkv, void* kc, void* vc, const int* anc, const int* anc_odd, int n,
                     int d, int heads, int Lmax, int pos, const int* pos_dev, int group,
                     const int* tok_tab, void* ctx, hipStream_t s) {
  const int dk = d / heads;
  group = group < 1 ? 1 : (group > 16 ? 16 : group);
  while (group > 1 && (size_t)group * 2 * Lmax * sizeof(float) > 64 * 1024) --group;  // default LDS limit
  dim3 grid(heads, em_cdiv(n, group)), block(64 * group);
  const size_t lds = (size_t)group * 2 * Lmax * sizeof(float);
  if (dk == 64)
    hipLaunchKernelGGL((dec_self_attn_kernel<T, 64>), grid, block, lds, s, (const T*)qkv, (T*)kc,
                       (T*)vc, anc, anc_odd, n, d, Lmax, pos, pos_dev, tok_tab, (T*)ctx);
  else if (dk == 32)
    hipLaunchKernelGGL((dec_self_attn_kernel<T, 32>), grid, block, lds, s, (const T*)qkv, (T*)kc,
                       (T*)vc, anc, anc_odd, n, d, Lmax, pos, pos_dev, tok_tab, (T*)ctx);
  else
    return EM_ERR_UNSUPPORTED;
  EM_CHECK_LAUNCH();
  return EM_OK;
}

template <typename T>
int src_attn_launch(const void* qs, const void* kmem, int ldk, const void* vT, const int* klens,
                    int B, int W, int d, int heads, int Tn, int Tpad, void* ctx, hipStream_t s) {
  const int dk = d / heads;
  const size_t lds = (size_t)16 * (Tpad + 4) * 4 + 64 + (size_t)16 * (Tpad + 16 / sizeof(T)) * sizeof(T);
  // the 16 score rows of a workgroup live in LDS: memories beyond ~1 690 frames (bf16; ~1 270 in f32) do
  // not fit the 160 KB of a CU -- say so instead of failing at launch
  if (lds > 160 * 1024) return EM_ERR_UNSUPPORTED;
  dim3 grid(heads, B, em_cdiv(W, 16));
  // (the attribute is raised once per process and size: no runtime API call on later launches,
  //  which also keeps the launch legal inside a stream capture)
  static size_t attr64 = 0, attr32 = 0;
  if (dk == 64) {
    if (lds > attr64) {
      if (hipFuncSetAttribute((const void*)dec_src_attn_kernel<T, 64>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return EM_ERR_LAUNCH;
      attr64 = lds;
    }
    hipLaunchKernelGGL((dec_src_attn_kernel<T, 64>), grid, dim3(256), lds, s, (const T*)qs,
                       (const T*)kmem, ldk, (const T*)vT, klens, W, d, Tn, Tpad, (T*)ctx);
  } else if (dk == 32) {
    if (lds > attr32) {
      if (hipFuncSetAttribute((const void*)dec_src_attn_kernel<T, 32>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return EM_ERR_LAUNCH;
      attr32 = lds;
    }
    hipLaunchKernelGGL((dec_src_attn_kernel<T, 32>), grid, dim3(256), lds, s, (const T*)qs,
                       (const T*)kmem, ldk, (const T*)vT, klens, W, d, Tn, Tpad, (T*)ctx);
  } else {
    return EM_ERR_UNSUPPORTED;
  }
  EM_CHECK_LAUNCH();
  return EM_OK;
}

}  // namespace

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

namespace {

// =================================================================================================
// One launch per decoder step (round 2): TransformerDecoder.forward_one_step for all rows in ONE kernel.
//
// Why.  The per-operator sequence is ~50 dependent launches per label step, each at or near the ~4.6 us a
// dependent launch costs on this part (profiles/r02f_bench_beam_large_b16.md: 0.65 ms per step, 0.5 ms of it
// the decoder).  A decoder step has no dependency BETWEEN hypotheses: a row reads its own ancestors' K/V and its
// utterance's memory.  So the rows are split by utterance over the 8 XCDs (32 CUs and one L2 each), every XCD runs
// the whole step for its rows, and the phases of the step (LayerNorm + projection, attention, ...) are separated
// by XCD-LOCAL barriers: a counter in that XCD's L2 and an L1 invalidate.  No device-scope fence anywhere -- those
// write back the L2 and cost ~20 us per launch when 32+ workgroups issue one (profiles/r02_experiments_not_kept.txt).
// Inside an XCD every projection is split by output columns over the 32 workgroups x 4 waves (a wave owns 16
// columns and all rows: the weights are streamed once per XCD), each workgroup re-normalising the <= 32 rows
// itself (64 KB of L2 reads instead of a barrier).
//
// Relies on workgroup L of a launch running on XCD L % 8 (MI355X_MICROARCH.md; also what csrc/gemm.hip's tile order
// assumes).  All 256 workgroups must be co-resident (one per CU); a spin that exceeds its budget raises an error
// flag instead of hanging.
struct MegaArgs {
  int B, W, T, Tpad, Lmax, d, ff, heads, L, V, U, n;  // U = utterances per XCD, n = B * W
  int pos;
  const int* pos_dev;
  const int *tok, *anc_a, *anc_b, *xlens;
  const float *embed, *pe;
  int pe_len;
  const float *after_g, *after_b;
  const void* out_w;
  const float* out_b;
  const EmDecoderLayer* layers;  // DEVICE copy of the layer table
  float* x;
  void *qkv, *qs, *ctx, *hbuf;
  float* logits;
  void *self_k, *self_v;
  const void *mem_kv, *mem_vT;
  unsigned* bar;  // [8][16] u32, zero between launches: [0] barrier count, [1] exit tickets, [8] error flag
  float eps;
};

constexpr int MG_ROWS = 32;  // rows per XCD (two MFMA row tiles)
constexpr int MG_NV = 8;     // d <= 64 * MG_NV

// LayerNorm of the XCD's rows (local row m -> global row row0 + min(m, nrows-1)) into LDS in the operand dtype:
// wave w rows 8w .. 8w+7, 16 lanes per row (csrc/ln_gemm.hip's prologue).
template <typename T>
__device__ __forceinline__ void mg_layernorm(const float* __restrict__ x, const float* __restrict__ g,
                                             const float* __restrict__ be, float eps, int K, int row0, int nrows,
                                             T* sA, int LDA, int wave, int lane) {
  const int grp = lane >> 4, li = lane & 15, nv = K >> 6;
  float4 g4[MG_NV], b4[MG_NV];
#pragma unroll
  for (int j = 0; j < MG_NV; ++j) {
    g4[j] = j < nv ? *(const float4*)(g + (j * 16 + li) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    b4[j] = j < nv ? *(const float4*)(be + (j * 16 + li) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int ml = wave * 8 + ps * 4 + grp;
    const int m = row0 + (ml < nrows ? ml : nrows - 1);
    const float4* xr = (const float4*)(x + (size_t)m * K);
    float4 v[MG_NV];
#pragma unroll
    for (int j = 0; j < MG_NV; ++j) v[j] = j < nv ? xr[j * 16 + li] : make_float4(0.f, 0.f, 0.f, 0.f);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MG_NV; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
    const float mean = s / (float)K;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MG_NV; ++j)
      if (j < nv) {
        const float a = v[j].x - mean, b2 = v[j].y - mean, c = v[j].z - mean, d2 = v[j].w - mean;
        q += (a * a + b2 * b2) + (c * c + d2 * d2);
      }
    q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64); q += __shfl_xor(q, 8, 64);
    const float rstd = 1.0f / sqrtf(q / (float)K + eps);
    T* dst = sA + (size_t)ml * LDA;
#pragma unroll
    for (int j = 0; j < MG_NV; ++j)
      if (j < nv) {
        const int c0 = (j * 16 + li) * 4;
        __attribute__((aligned(16))) T o[4];
        o[0] = from_f32<T>((v[j].x - mean) * rstd * g4[j].x + b4[j].x);
        o[1] = from_f32<T>((v[j].y - mean) * rstd * g4[j].y + b4[j].y);
        o[2] = from_f32<T>((v[j].z - mean) * rstd * g4[j].z + b4[j].z);
        o[3] = from_f32<T>((v[j].w - mean) * rstd * g4[j].w + b4[j].w);
        if (sizeof(T) == 2) *(uint2*)(dst + c0) = *(const uint2*)o;
        else *(uint4*)(dst + c0) = *(const uint4*)o;
      }
  }
}

constexpr int MG_STORE = 0, MG_RELU = 1, MG_STORE_F32 = 2, MG_RESID = 3;

// C/D fragment (col = lr, row = lg*4 + r of row tile i) -> memory
template <typename T, int EPI>
__device__ __forceinline__ void mg_epilogue(const f32x4 acc, int i, int ncol, int N, const float* __restrict__ bias,
                                            void* __restrict__ Cv, int ldc, int row0, int nrows, int lg) {
  if (ncol >= N) return;
  const float bv = bias ? bias[ncol] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ml = i * 16 + lg * 4 + r;
    if (ml >= nrows) continue;
    const size_t o = (size_t)(row0 + ml) * ldc + ncol;
    float y = acc[r] + bv;
    if (EPI == MG_RELU) y = fmaxf(y, 0.f);
    if (EPI == MG_STORE_F32) ((float*)Cv)[o] = y;
    else if (EPI == MG_RESID) ((float*)Cv)[o] += y;
    else ((T*)Cv)[o] = from_f32<T>(y);
  }
}

// Projection of the XCD's rows, split by 16-column tiles over the XCD's waves (tile = slot*4 + wave, stride
// 4*NS); a wave walks the whole K.  A rows from LDS (sA != NULL: the LayerNorm output) or from global memory
// (Ag, row stride lda).
template <typename T, int EPI, bool FROM_LDS>
__device__ __forceinline__ void mg_project(const T* sA, int LDA, const T* __restrict__ Ag, int lda,
                                           const T* __restrict__ W, const float* __restrict__ bias,
                                           void* __restrict__ Cv, int ldc, int N, int K, int row0, int nrows,
                                           int slot, int NS, int wave, int lane) {
  using MM = Mma<T>;
  constexpr int U = 16;
  const int lr = lane & 15, lg = lane >> 4;
  const int nsteps = K / MM::K, ntiles = (N + 15) >> 4;
  // (two separate pointer pairs: one pointer that may be LDS or global crashes hipcc's address-space inference)
  const T* const l0 = sA + (size_t)lr * LDA + lg * MM::EPL;
  const T* const l1 = l0 + (size_t)16 * LDA;
  const int m0 = lr < nrows ? lr : nrows - 1, m1 = 16 + lr < nrows ? 16 + lr : nrows - 1;
  const T* const g0 = Ag + (size_t)(row0 + m0) * lda + lg * MM::EPL;
  const T* const g1 = Ag + (size_t)(row0 + m1) * lda + lg * MM::EPL;
  for (int tile = slot * 4 + wave; tile < ntiles; tile += 4 * NS) {
    int n = tile * 16 + lr;
    n = n < N ? n : N - 1;
    const T* wrow = W + (size_t)n * K + lg * MM::EPL;
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < nsteps; s0 += U) {
      typename MM::frag fw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) fw[u] = MM::load(wrow + (size_t)(s0 + u < nsteps ? s0 + u : nsteps - 1) * MM::K);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (s0 + u < nsteps) {
          const size_t ko = (size_t)(s0 + u) * MM::K;
          acc0 = MM::mma(FROM_LDS ? MM::load(l0 + ko) : MM::load(g0 + ko), fw[u], acc0);
          if (nrows > 16) acc1 = MM::mma(FROM_LDS ? MM::load(l1 + ko) : MM::load(g1 + ko), fw[u], acc1);
        }
    }
    mg_epilogue<T, EPI>(acc0, 0, tile * 16 + lr, N, bias, Cv, ldc, row0, nrows, lg);
    if (nrows > 16) mg_epilogue<T, EPI>(acc1, 1, tile * 16 + lr, N, bias, Cv, ldc, row0, nrows, lg);
  }
}

// The long-K residual projection (FFN w_2): one 16-column tile per workgroup, K split over its 4 waves, partial
// tiles added in wave order through LDS (red: [4][2][256] f32).
template <typename T>
__device__ __forceinline__ void mg_project_splitk(const T* __restrict__ Ag, int lda, const T* __restrict__ W,
                                                  const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                  int N, int K, int row0, int nrows, int slot, int NS, int wave,
                                                  int lane, float* red) {
  using MM = Mma<T>;
  constexpr int U = 16;
  const int lr = lane & 15, lg = lane >> 4;
  const int nsteps = K / MM::K, per = nsteps >> 2, ntiles = (N + 15) >> 4;
  const int m0 = lr < nrows ? lr : nrows - 1, m1 = 16 + lr < nrows ? 16 + lr : nrows - 1;
  const T* a0 = Ag + (size_t)(row0 + m0) * lda + lg * MM::EPL;
  const T* a1 = Ag + (size_t)(row0 + m1) * lda + lg * MM::EPL;
  for (int tile = slot; tile < ntiles; tile += NS) {  // workgroup-uniform trip count: barriers inside
    int n = tile * 16 + lr;
    n = n < N ? n : N - 1;
    const T* wrow = W + (size_t)n * K + lg * MM::EPL;
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s0 = wave * per; s0 < (wave + 1) * per; s0 += U) {
      const int lim = (wave + 1) * per;
      typename MM::frag fw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) fw[u] = MM::load(wrow + (size_t)(s0 + u < lim ? s0 + u : lim - 1) * MM::K);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (s0 + u < lim) {
          acc0 = MM::mma(MM::load(a0 + (size_t)(s0 + u) * MM::K), fw[u], acc0);
          if (nrows > 16) acc1 = MM::mma(MM::load(a1 + (size_t)(s0 + u) * MM::K), fw[u], acc1);
        }
    }
    *(f32x4*)(red + ((wave * 2 + 0) * 64 + lane) * 4) = acc0;
    *(f32x4*)(red + ((wave * 2 + 1) * 64 + lane) * 4) = acc1;
    __syncthreads();
    if (wave < 2 && (wave == 0 || nrows > 16)) {
      f32x4 s = *(const f32x4*)(red + ((0 * 2 + wave) * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const f32x4 v = *(const f32x4*)(red + ((w * 2 + wave) * 64 + lane) * 4);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
      }
      mg_epilogue<T, MG_RESID>(s, wave, tile * 16 + lr, N, bias, C, ldc, row0, nrows, lg);
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dec_mega_kernel(const MegaArgs a) {
  using MM = Mma<T>;
  constexpr int DK = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char mg_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, NS = gridDim.x >> 3;
  int pos = a.pos;
  if (a.pos_dev) {
    pos = *a.pos_dev;
    if (pos >= a.Lmax || pos >= a.pe_len) return;  // a replayed graph may run past the last useful step
  }
  const int* anc = (pos & 1) ? a.anc_b : a.anc_a;
  const int u0 = xcd * a.U;
  const int nu = a.B - u0 < a.U ? a.B - u0 : a.U;
  if (nu <= 0) return;  // this XCD has no utterance: nothing of it takes part in any barrier
  const int row0 = u0 * a.W, nrows = nu * a.W, n = a.n, d = a.d;
  unsigned* const bar = a.bar + xcd * 16;
  unsigned epoch = 0;
  // XCD-local barrier: every wave's stores have been acknowledged by the L2, one arrival per workgroup on a
  // counter in that L2, then this CU's L1 is dropped so that the other CUs' rows are read from the L2
  auto xbar = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    ++epoch;
    if (tid == 0) {
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = epoch * (unsigned)NS;
      int spins = 0;
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) {  // ~ a second: a workgroup of this XCD is not resident
          bar[8] = 1u;
          break;
        }
      }
    }
    __syncthreads();
    asm volatile("buffer_inv sc1" ::: "memory");
  };
  const int LDA = d + 16 / (int)sizeof(T);
  T* const sA = (T*)mg_smem;
  float* const fbuf = (float*)mg_smem;
  const size_t es = sizeof(T);
  const float xscale = sqrtf((float)d);

  // ---- x = embed[tok] * sqrt(d) + pe[pos]   (embedding.py:93)
  {
    const int* tok_row = a.tok + (size_t)pos * n;
    for (int ml = slot * 4 + wave; ml < nrows; ml += 4 * NS) {
      const int r = row0 + ml;
      int t = tok_row[r];
      t = t < 0 ? 0 : (t >= a.V ? a.V - 1 : t);
      const float* e = a.embed + (size_t)t * d;
      const float* p = a.pe + (size_t)pos * d;
      for (int c = lane; c < d; c += 64) a.x[(size_t)r * d + c] = e[c] * xscale + p[c];
    }
  }
  xbar();
  for (int l = 0; l < a.L; ++l) {
    const EmDecoderLayer q = a.layers[l];
    T* const kc = (T*)a.self_k + (size_t)l * a.Lmax * n * d;
    T* const vc = (T*)a.self_v + (size_t)l * a.Lmax * n * d;
    // ---- q | k | v = LN1(x) Wqkv^T + b   (decoder_layer.py:96-113)
    mg_layernorm<T>(a.x, q.norm1_g, q.norm1_b, a.eps, d, row0, nrows, sA, LDA, wave, lane);
    __syncthreads();
    mg_project<T, MG_STORE, true>(sA, LDA, (const T*)a.ctx, d, (const T*)q.self_wqkv, q.self_bqkv, a.qkv, 3 * d, 3 * d, d, row0,
                            nrows, slot, NS, wave, lane);
    xbar();
    // ---- self-attention over the token-tree K/V cache: one wave per (row, head) (dec_self_attn_kernel)
    {
      constexpr int NCH = DK / 8, NJ = 64 / NCH;
      float* p_s = fbuf + (size_t)wave * 2 * a.Lmax;
      int* a_s = (int*)(p_s + a.Lmax);
      const int nitems = nrows * a.heads;
      const int iters = (nitems + 4 * NS - 1) / (4 * NS);  // the same for every wave: barriers inside
      const int ch = lane % NCH, jsub = lane / NCH;
      const float scale = rsqrtf((float)DK);
      for (int it = 0; it < iters; ++it) {
        const int item = (it * NS + slot) * 4 + wave;
        const bool live = item < nitems;
        const int itc = live ? item : nitems - 1;
        const int r = row0 + itc / a.heads, h = itc % a.heads;
        const T* row = (const T*)a.qkv + (size_t)r * 3 * d + h * DK;
        float qv[8];
        load8<T>(row + ch * 8, qv);
        if (lane < NCH && live) {
          const size_t o = ((size_t)pos * n + r) * d + h * DK + lane * 8;
          float t8[8];
          load8<T>(row + d + lane * 8, t8);
          store8<T>(kc + o, t8);
          load8<T>(row + 2 * d + lane * 8, t8);
          store8<T>(vc + o, t8);
        }
        for (int j = lane; j < pos; j += 64) a_s[j] = anc[(size_t)r * a.Lmax + j];
        __syncthreads();
        const int niter = (pos + NJ) / NJ;
        constexpr int UN = sizeof(T) == 2 ? 8 : 4;
        for (int it0 = 0; it0 < niter; it0 += UN) {
          Raw8<T> k8[UN];
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            const int j = (it0 + u) * NJ + jsub;
            if (j <= pos) {
              const T* kr = (j == pos) ? row + d : kc + ((size_t)j * n + a_s[j]) * d + h * DK;
              k8[u].load(kr + ch * 8);
            }
          }
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            const int j = (it0 + u) * NJ + jsub;
            float dot = 0.f;
            if (j <= pos) {
#pragma unroll
              for (int e = 0; e < 8; ++e) dot = fmaf(qv[e], k8[u].at(e), dot);
            }
#pragma unroll
            for (int o = 1; o < NCH; o <<= 1) dot += __shfl_xor(dot, o, 64);
            if (ch == 0 && j <= pos) p_s[j] = dot * scale;
          }
        }
        __syncthreads();
        float mx = -INFINITY;
        for (int j = lane; j <= pos; j += 64) mx = fmaxf(mx, p_s[j]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j <= pos; j += 64) {
          const float p = expf(p_s[j] - mx);
          p_s[j] = p;
          sum += p;
        }
        sum = wave_sum(sum);
        __syncthreads();
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int it0 = 0; it0 < niter; it0 += UN) {
          Raw8<T> v8[UN];
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            const int j = (it0 + u) * NJ + jsub;
            if (j <= pos) {
              const T* vr = (j == pos) ? row + 2 * d : vc + ((size_t)j * n + a_s[j]) * d + h * DK;
              v8[u].load(vr + ch * 8);
            }
          }
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            const int j = (it0 + u) * NJ + jsub;
            if (j <= pos) {
              const float p = p_s[j];
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, v8[u].at(e), acc[e]);
            }
          }
        }
#pragma unroll
        for (int o = NCH; o < 64; o <<= 1)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
        if (lane < NCH && live) {
          const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] *= inv;
          store8<T>((T*)a.ctx + (size_t)r * d + h * DK + lane * 8, acc);
        }
        __syncthreads();
      }
    }
    xbar();
    // ---- x += ctx Wo^T + b
    mg_project<T, MG_RESID, false>(sA, LDA, (const T*)a.ctx, d, (const T*)q.self_wout, q.self_bout, a.x, d, d, d, row0,
                            nrows, slot, NS, wave, lane);
    xbar();
    // ---- qs = LN2(x) Wq^T + b   (decoder_layer.py:119-135)
    mg_layernorm<T>(a.x, q.norm2_g, q.norm2_b, a.eps, d, row0, nrows, sA, LDA, wave, lane);
    __syncthreads();
    mg_project<T, MG_STORE, true>(sA, LDA, (const T*)a.ctx, d, (const T*)q.src_wq, q.src_bq, a.qs, d, d, d, row0, nrows, slot, NS,
                            wave, lane);
    xbar();
    // ---- source attention: one workgroup per (utterance, head) (dec_src_attn_kernel, W <= 16 rows)
    {
      constexpr int KS = DK / MM::K;
      const int Tpad = a.Tpad, Tn = a.T;
      const int LDS_S = Tpad + 4, LDS_P = Tpad + 16 / (int)sizeof(T);
      float* S = fbuf;
      float* sums = S + 16 * LDS_S;
      T* P = (T*)(sums + 16);
      const T* kmem = (const T*)a.mem_kv + (size_t)l * a.B * Tn * 2 * d;
      const T* vT = (const T*)a.mem_vT + (size_t)l * a.B * d * Tpad;
      const int ldk = 2 * d;
      const float scale = rsqrtf((float)DK);
      for (int item = slot; item < nu * a.heads; item += NS) {  // workgroup-uniform
        const int b = u0 + item / a.heads, h = item % a.heads;
        const int rw0 = b * a.W, nr = a.W;
        const int klen = a.xlens[b] < Tn ? a.xlens[b] : Tn;
        typename MM::frag qf[KS];
        {
          const int rr = lr < nr ? lr : nr - 1;
          const T* qrow = (const T*)a.qs + (size_t)(rw0 + rr) * d + h * DK;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) qf[ks] = MM::load(qrow + ks * MM::K + lg * MM::EPL);
        }
        const T* kb = kmem + (size_t)b * Tn * ldk + h * DK;
        for (int nt = wave; nt < Tpad / 16; nt += 4) {
          const int key = nt * 16 + lr;
          const int kc_ = key < Tn ? key : Tn - 1;
          f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            acc = MM::mma(qf[ks], MM::load(kb + (size_t)kc_ * ldk + ks * MM::K + lg * MM::EPL), acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) S[(lg * 4 + r) * LDS_S + key] = key < klen ? acc[r] * scale : -INFINITY;
        }
        __syncthreads();
        for (int rr = wave * 4; rr < wave * 4 + 4; ++rr) {
          float mx = -INFINITY;
          for (int j = lane; j < Tpad; j += 64) mx = fmaxf(mx, S[rr * LDS_S + j]);
          mx = wave_max(mx);
          float sum = 0.f;
          for (int j = lane; j < Tpad; j += 64) {
            T pt = from_f32<T>(expf(S[rr * LDS_S + j] - mx));
            P[rr * LDS_P + j] = pt;
            sum += to_f32(pt);
          }
          sum = wave_sum(sum);
          if (lane == 0) sums[rr] = sum;
        }
        __syncthreads();
        if (wave < DK / 16) {
          const T* vb = vT + ((size_t)b * d + h * DK + wave * 16 + lr) * Tpad;
          f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
          for (int kk = 0; kk < Tpad / MM::K; ++kk) {
            const int ko = kk * MM::K + lg * MM::EPL;
            acc = MM::mma(MM::load(P + lr * LDS_P + ko), MM::load(vb + ko), acc);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rr = lg * 4 + r;
            if (rr < nr) ((T*)a.ctx)[(size_t)(rw0 + rr) * d + h * DK + wave * 16 + lr] = from_f32<T>(acc[r] / sums[rr]);
          }
        }
        __syncthreads();
      }
    }
    xbar();
    // ---- x += ctx Wo^T + b
    mg_project<T, MG_RESID, false>(sA, LDA, (const T*)a.ctx, d, (const T*)q.src_wout, q.src_bout, a.x, d, d, d, row0,
                            nrows, slot, NS, wave, lane);
    xbar();
    // ---- h = relu(LN3(x) W1^T + b1)   (decoder_layer.py:150-152, positionwise_feed_forward.py:30-32)
    mg_layernorm<T>(a.x, q.norm3_g, q.norm3_b, a.eps, d, row0, nrows, sA, LDA, wave, lane);
    __syncthreads();
    mg_project<T, MG_RELU, true>(sA, LDA, (const T*)a.ctx, d, (const T*)q.w1, q.b1, a.hbuf, a.ff, a.ff, d, row0, nrows, slot, NS,
                           wave, lane);
    xbar();
    // ---- x += h W2^T + b2
    mg_project_splitk<T>((const T*)a.hbuf, a.ff, (const T*)q.w2, q.b2, a.x, d, d, a.ff, row0, nrows, slot, NS, wave,
                         lane, fbuf);
    xbar();
  }
  // ---- logits = after_norm(x) Wout^T + b   (transformer_decoder.py:226-233; log-softmax in the pre-beam kernel)
  mg_layernorm<T>(a.x, a.after_g, a.after_b, a.eps, d, row0, nrows, sA, LDA, wave, lane);
  __syncthreads();
  mg_project<T, MG_STORE_F32, true>(sA, LDA, (const T*)a.ctx, d, (const T*)a.out_w, a.out_b, a.logits, a.V, a.V, d, row0, nrows,
                              slot, NS, wave, lane);
  // ---- leave: the last workgroup of this XCD to get here re-arms the counters for the next launch (every other
  // one has left its last barrier by then)
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (unsigned)NS - 1) {
      __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(bar + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

namespace {
// The kernel indexes the layer table at run time, so it needs it in device memory (EmDecoderWeights.layers is a
// host array).  One small device copy per weight set, refreshed when the host table changes; never allocated or
// copied while a stream capture is in progress.
struct LayerCopy {
  std::vector<EmDecoderLayer> host;
  EmDecoderLayer* dev = nullptr;
};
std::mutex g_layers_mu;
std::unordered_map<const void*, LayerCopy> g_layers;

const EmDecoderLayer* device_layers(const EmDecoderWeights* dw) {
  std::lock_guard<std::mutex> lk(g_layers_mu);
  const int L = dw->num_blocks;
  LayerCopy& c = g_layers[(const void*)dw->layers];
  if (c.dev && (int)c.host.size() == L && memcmp(c.host.data(), dw->layers, L * sizeof(EmDecoderLayer)) == 0)
    return c.dev;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  // a capture on ANY stream of this thread forbids synchronous copies: be conservative
  if (hipThreadExchangeStreamCaptureMode == nullptr) return nullptr;
  if (!c.dev && hipMalloc((void**)&c.dev, 8 * sizeof(EmDecoderLayer)) != hipSuccess) return nullptr;
  (void)st;
  if (hipMemcpy(c.dev, dw->layers, L * sizeof(EmDecoderLayer), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  c.host.assign(dw->layers, dw->layers + L);
  return c.dev;
}
}  // namespace

// Internal (csrc/search.hip): the whole decoder step in one launch, when the shape fits (see em_common.h).
int em_decoder_mega_step(int dtype, const EmDecoderWeights* dw, int B, int W, int T, int Tpad, int Lmax, int pos,
                         const int32_t* pos_dev, const int32_t* tok, const int32_t* anc_a, const int32_t* anc_b,
                         const int32_t* xlens, void* self_k, void* self_v, const void* mem_kv, const void* mem_vT,
                         float* x, void* qkv, void* qs, void* ctx, void* hbuf, float* logits, void* bar,
                         void* stream) {
  const int d = dw->d, U = (B + 7) / 8;
  if (!bar || dw->num_blocks > 8 || dw->heads * 64 != d || d % 64 != 0 || d > 64 * MG_NV) return EM_ERR_UNSUPPORTED;
  if (W > 16 || U * W > MG_ROWS || dw->ff % 128 != 0 || Tpad % 32 != 0) return EM_ERR_UNSUPPORTED;
  const size_t es = dtype == EM_BF16 ? 2 : 4;
  const size_t lds_a = (size_t)MG_ROWS * (d + 16 / es) * es;
  const size_t lds_sa = (size_t)4 * 2 * Lmax * sizeof(float);
  const size_t lds_src = (size_t)16 * (Tpad + 4) * 4 + 64 + (size_t)16 * (Tpad + 16 / es) * es;
  size_t lds = lds_a > lds_sa ? lds_a : lds_sa;
  lds = lds > lds_src ? lds : lds_src;
  lds = lds > 8192 ? lds : 8192;
  if (lds > 150 * 1024) return EM_ERR_UNSUPPORTED;
  MegaArgs a = {};
  a.B = B; a.W = W; a.T = T; a.Tpad = Tpad; a.Lmax = Lmax; a.d = d; a.ff = dw->ff; a.heads = dw->heads;
  a.L = dw->num_blocks; a.V = dw->vocab; a.U = U; a.n = B * W;
  a.pos = pos; a.pos_dev = pos_dev; a.tok = tok; a.anc_a = anc_a; a.anc_b = anc_b; a.xlens = xlens;
  a.embed = dw->embed; a.pe = dw->pe; a.pe_len = dw->pe_len;
  a.after_g = dw->after_norm_g; a.after_b = dw->after_norm_b; a.out_w = dw->out_w; a.out_b = dw->out_b;
  a.layers = device_layers(dw);
  if (!a.layers) return EM_ERR_UNSUPPORTED;  // first use inside a stream capture: the per-operator sequence runs
  a.x = x; a.qkv = qkv; a.qs = qs; a.ctx = ctx; a.hbuf = hbuf; a.logits = logits;
  a.self_k = self_k; a.self_v = self_v; a.mem_kv = mem_kv; a.mem_vT = mem_vT;
  a.bar = (unsigned*)bar; a.eps = 1e-12f;
  hipStream_t s = (hipStream_t)stream;
  static size_t attr_bf = 0, attr_f = 0;
  if (dtype == EM_BF16) {
    if (lds > attr_bf) {
      if (hipFuncSetAttribute((const void*)dec_mega_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return EM_ERR_LAUNCH;
      attr_bf = lds;
    }
    hipLaunchKernelGGL(dec_mega_kernel<bf16>, dim3(256), dim3(256), lds, s, a);
  } else if (dtype == EM_F32) {
    if (lds > attr_f) {
      if (hipFuncSetAttribute((const void*)dec_mega_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return EM_ERR_LAUNCH;
      attr_f = lds;
    }
    hipLaunchKernelGGL(dec_mega_kernel<float>, dim3(256), dim3(256), lds, s, a);
  } else {
    return EM_ERR_BAD_ARG;
  }
  EM_CHECK_LAUNCH();
  return EM_OK;
}
