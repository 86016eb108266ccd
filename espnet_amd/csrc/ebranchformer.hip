// Host-side launch sequence of the E-Branchformer / Branchformer encoder: one C call enqueues every kernel
// of a forward pass on the caller's stream.  The only device code here is the Branchformer learned_ave
// merge (two small kernels at the end of the anonymous namespace).
//
// Reference call graph being reproduced:
//   EBranchformerEncoder.forward       espnet2/asr/encoder/e_branchformer_encoder.py:418-520
//   EBranchformerEncoderLayer.forward  espnet2/asr/encoder/e_branchformer_encoder.py:110-183
//   ConvolutionalGatingMLP / CSGU      espnet2/asr/layers/cgmlp.py:14-145
//   BranchformerEncoderLayer.forward   espnet2/asr/encoder/branchformer_encoder.py:138-290 (merges :207-278)
//   Conv2dSubsampling, RelPositionalEncoding, RelPositionMultiHeadedAttention, PositionwiseFeedForward:
//   the Conformer ones (see encoder.hip)
//
// Per layer: macaron FFN -> two branches on the same x (rel-pos attention | cgMLP) whose outputs are
// written side by side into one [M][2d] matrix by their last GEMMs (ldc = 2d) -> depthwise conv over
// the 2d channels with the self residual fused -> merge_proj GEMM into the f32 residual stream ->
// FFN -> norm_final (fused with the next layer's first LayerNorm).
#include <math.h>
#include <stdlib.h>

#include "em_common.h"
#include "switches.h"
#include "subsample.h"

namespace {

constexpr float LN_EPS = 1e-12f;

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Ws {
  size_t c1, c2, c3, x, xn, xn2, big, gn, gated, cat, tmp, ctx, pall, mw, qh, kh, vt, total;
  int Tpad;
};
inline Ws layout(int dtype, const EmEBranchformerWeights* w, int B, int T_f) {
  const size_t es = dtype == EM_BF16 ? 2 : 4;
  em_sub::Geo g;
  em_sub::geo(w->subsample, T_f, w->n_mels, &g);
  const size_t M = (size_t)B * g.T_out, d = w->d;
  size_t mb[3];
  em_sub::map_bytes(g, B, w->d, es, mb);
  size_t wide = w->ff > 3 * w->d ? w->ff : 3 * w->d;
  if ((size_t)w->cg > wide) wide = w->cg;
  Ws s;
  size_t o = 0;
  s.c1 = o; o += align_up(mb[0]);
  s.c2 = o; o += align_up(mb[1]);
  s.c3 = o; o += align_up(mb[2]);
  s.x = o; o += align_up(M * d * 4);
  s.xn = o; o += align_up(M * d * es);
  s.xn2 = o; o += align_up(M * d * es);
  s.big = o; o += align_up(M * wide * es);
  s.gn = o; o += align_up(M * (size_t)(w->cg / 2) * es);
  s.gated = o; o += align_up(M * (size_t)(w->cg / 2) * es);
  s.cat = o; o += align_up(M * 2 * d * es);
  s.tmp = o; o += align_up(M * 2 * d * es);
  s.ctx = o; o += align_up(M * d * es);
  s.pall = o; o += align_up((size_t)(2 * g.T_out - 1) * w->num_blocks * d * es);
  s.mw = o; o += align_up((size_t)B * 2 * sizeof(float));
  // per-head operands of the LDS-resident attention (csrc/attention2.hip; bf16, d_k = 64): as in encoder.hip
  s.Tpad = (g.T_out + 255) / 256 * 256;
  const size_t per_head = (size_t)B * d * s.Tpad * es;
  s.qh = o; o += align_up(per_head);
  s.kh = o; o += align_up(per_head);
  s.vt = o; o += align_up(per_head);
  s.total = o;
  return s;
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

// ---- Branchformer learned_ave (branchformer_encoder.py:212-270).  Branch k of utterance b is attention-pooled
// over its valid frames with scores (x_t . pool_w + pool_b) / sqrt(d), and the pooled vector is projected to
// one scalar: weight = (sum_t softmax_t x_t) . wproj_w + wproj_b = sum_t softmax_t (x_t . wproj_w) + wproj_b,
// so one pass over the rows with two dot products per row and an online softmax does it.
// grid (B, 2), 4 waves; wave w takes rows w, w+4, ...
template <typename T>
__global__ __launch_bounds__(256) void branch_pool_kernel(const T* __restrict__ cat, const int32_t* __restrict__ lens,
                                                          int Tn, int d, const float* __restrict__ pool_w,
                                                          const float* __restrict__ pool_b,
                                                          const float* __restrict__ wproj_w,
                                                          const float* __restrict__ wproj_b, float* __restrict__ mw) {
  const int b = blockIdx.x, k = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int n = lens[b];
  n = n < Tn ? n : Tn;
  const float* pw = pool_w + (size_t)k * d;
  const float* ww = wproj_w + (size_t)k * d;
  const float inv = 1.f / sqrtf((float)d), pb = pool_b[k];
  float m = -INFINITY, l = 0.f, acc = 0.f;
  for (int t = wave; t < n; t += 4) {
    const T* row = cat + ((size_t)b * Tn + t) * 2 * d + (size_t)k * d;
    float s = 0.f, u = 0.f;
    for (int c = lane; c < d; c += 64) {
      const float v = to_f32(row[c]);
      s += v * pw[c];
      u += v * ww[c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o, 64);
      u += __shfl_xor(u, o, 64);
    }
    s = (s + pb) * inv;
    const float mn = fmaxf(m, s), a = expf(m - mn), e = expf(s - mn);
    l = l * a + e;
    acc = acc * a + e * u;
    m = mn;
  }
  __shared__ float sm[4], sl[4], sa[4];
  if (lane == 0) {
    sm[wave] = m;
    sl[wave] = l;
    sa[wave] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])), L = 0.f, A = 0.f;
    for (int w = 0; w < 4; ++w)
      if (sl[w] > 0.f) {
        const float a = expf(sm[w] - M);
        L += sl[w] * a;
        A += sa[w] * a;
      }
    mw[b * 2 + k] = A / L + wproj_b[k];
  }
}

// out[b,t,:] = w1 x1 + w2 x2 with (w1, w2) = softmax(mw[b]) (:260-269)
template <typename T>
__global__ __launch_bounds__(256) void branch_mix_kernel(const T* __restrict__ cat, const float* __restrict__ mw,
                                                         int Tn, int d, T* __restrict__ out) {
  const int row = blockIdx.x, b = row / Tn;
  const float a = mw[b * 2], c2 = mw[b * 2 + 1], mx = fmaxf(a, c2);
  const float e1 = expf(a - mx), e2 = expf(c2 - mx), w1 = e1 / (e1 + e2), w2 = e2 / (e1 + e2);
  const T* x1 = cat + (size_t)row * 2 * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x)
    out[(size_t)row * d + c] = from_f32<T>(w1 * to_f32(x1[c]) + w2 * to_f32(x1[d + c]));
}

}  // namespace

extern "C" int em_branch_learned_ave(int dtype, const void* cat, const int32_t* lens, int32_t B, int32_t T,
                                     int32_t d, const float* pool_w, const float* pool_b, const float* wproj_w,
                                     const float* wproj_b, float* mw_ws, void* out, void* stream) {
  if (!cat || !lens || !pool_w || !pool_b || !wproj_w || !wproj_b || !mw_ws || !out) return EM_ERR_BAD_ARG;
  if (B <= 0 || T <= 0 || d <= 0) return EM_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == EM_BF16) {
    hipLaunchKernelGGL(branch_pool_kernel<bf16>, dim3(B, 2), dim3(256), 0, s, (const bf16*)cat, lens, T, d, pool_w,
                       pool_b, wproj_w, wproj_b, mw_ws);
    hipLaunchKernelGGL(branch_mix_kernel<bf16>, dim3(B * T), dim3(256), 0, s, (const bf16*)cat, mw_ws, T, d,
                       (bf16*)out);
  } else if (dtype == EM_F32) {
    hipLaunchKernelGGL(branch_pool_kernel<float>, dim3(B, 2), dim3(256), 0, s, (const float*)cat, lens, T, d, pool_w,
                       pool_b, wproj_w, wproj_b, mw_ws);
    hipLaunchKernelGGL(branch_mix_kernel<float>, dim3(B * T), dim3(256), 0, s, (const float*)cat, mw_ws, T, d,
                       (float*)out);
  } else {
    return EM_ERR_BAD_ARG;
  }
  EM_CHECK_LAUNCH();
  return EM_OK;
}

extern "C" size_t em_ebranchformer_workspace_bytes(int dtype, const EmEBranchformerWeights* w, int32_t B,
                                                   int32_t T_f) {
  if (!w || B <= 0 || T_f < em_sub::min_frames(w->subsample)) return 0;
  return layout(dtype, w, B, T_f).total;
}

extern "C" int em_ebranchformer_encode(int dtype, const EmEBranchformerWeights* w, const float* feats,
                                       const float* mvn_partial, const int32_t* flens,
                                       const int32_t* olens, int32_t B, int32_t T_f, const void* pos_emb,
                                       void* workspace, size_t workspace_bytes, float* enc_out,
                                       void* enc_act, int32_t flags, void* stream) {
  const int32_t* conv_lens = (flags & EM_ENC_ISOLATE_UTTS) ? olens : nullptr;
  if (!w || !feats || !flens || !olens || !pos_emb || !workspace || !enc_out || !enc_act) return EM_ERR_BAD_ARG;
  if (dtype != EM_F32 && dtype != EM_BF16) return EM_ERR_BAD_ARG;
  if (B <= 0) return EM_ERR_BAD_ARG;
  if (w->merge_method != EM_MERGE_CONCAT && (w->merge_method != EM_MERGE_LEARNED_AVE || w->merge_conv))
    return EM_ERR_UNSUPPORTED;
  if (T_f < em_sub::min_frames(w->subsample)) return EM_ERR_TOO_SHORT;
  const int d = w->d, h = w->heads, ff = w->ff, cg = w->cg, L = w->num_blocks, ch = cg / 2;
  const int kalign = dtype == EM_BF16 ? 64 : 32;
  if (d % 64 != 0 || h <= 0 || d / h != 64 || (w->use_ffn && ff % 64 != 0) || cg % 128 != 0 || ch % kalign != 0)
    return EM_ERR_UNSUPPORTED;
  const Ws s = layout(dtype, w, B, T_f);
  if (workspace_bytes < s.total) return EM_ERR_WORKSPACE;
  em_sub::Geo g;
  if (!em_sub::geo(w->subsample, T_f, w->n_mels, &g)) return EM_ERR_UNSUPPORTED;
  const int T = g.T_out, M = B * T;
  unsigned char* ws = (unsigned char*)workspace;
  void* c1 = ws + s.c1;
  void* c2 = ws + s.c2;
  void* c3 = ws + s.c3;
  float* x = (float*)(ws + s.x);
  void* xn = ws + s.xn;
  void* xn2 = ws + s.xn2;
  void* big = ws + s.big;
  void* gn = ws + s.gn;
  void* gated = ws + s.gated;
  unsigned char* cat = ws + s.cat;
  void* tmp = ws + s.tmp;
  void* ctx = ws + s.ctx;
  void* pall = ws + s.pall;
  const size_t es = dtype == EM_BF16 ? 2 : 4;

  // ---- Conv2dSubsampling{,6,8} (+MVN) -> Linear, * sqrt(d) (subsample.h); linear_pos of every block in one GEMM
  EM_TRY(em_sub::run(dtype, w, g, feats, mvn_partial, flens, B, c1, c2, c3, x, stream, w->conv1_wf, w->conv2_wf));
  EM_TRY(gemm(dtype, EM_EPI_STORE, pos_emb, w->wpos_all, pall, nullptr, w->legacy_relpos ? T : 2 * T - 1, L * d, d, d, L * d, 1.f,
              stream));
  const EmEBranchformerLayer* ly = w->layers;
  const bool ffn = w->use_ffn != 0;
  // round 4: bf16 with d_k = 64 -> the LDS-resident attention, its operands written per head by the projection GEMMs
  // (see encoder.hip; ESPNET_AMD_NO_ATTN2_LARGE=1: developer A/B switch)
  const bool no_attn2 = em_sw().no_attn2_large;
  const bool attn2 = dtype == EM_BF16 && !w->legacy_relpos && !no_attn2 && !(flags & EM_ENC_NO_FUSED) && d == 64 * h &&
                     (size_t)B * d * s.Tpad * 4 < ((size_t)1 << 32) - 64;
  void* qh = ws + s.qh;
  void* vt = ws + s.vt;
  if (attn2 && hipMemsetAsync(qh, 0, (s.vt - s.qh) + (size_t)B * d * s.Tpad * es, (hipStream_t)stream) != hipSuccess)
    return EM_ERR_LAUNCH;
  // Round 6: the two feed-forward modules as row-block launches of csrc/ffn_rows.hip (bf16, d = 512, the host packed their
  // operand streams): [macaron FFN + residual + norm_mha] and [FFN + residual + norm_final + the next LayerNorm] - three
  // launches each until now (two tiled GEMMs through the [M][ff] hidden activation + a LayerNorm) - when a round of
  // 64-row workgroups fills its share of the chip (em_rows_fill_ok: B = 32 x 10 s is 125 workgroups - taken with two batches
  // in flight, EM_ENC_IN_FLIGHT).  ESPNET_AMD_NO_FFN_ROWS=1: developer switch.
  bool ffn_rows = ffn && dtype == EM_BF16 && d == 512 && ff % 128 == 0 && ff >= 256 && !(flags & EM_ENC_NO_FUSED) &&
                  !em_sw().no_ffn_rows && em_rows_fill_ok(M, flags);
  for (int l = 0; ffn_rows && l < L; ++l) ffn_rows = ly[l].ffm_w1p && ly[l].ffm_w2p && ly[l].ff_w1p && ly[l].ff_w2p;
  const EmEBranchformerLayer* ln_in = nullptr;  // set for the launch that computes norm_ff itself (reset after it)
  // ... and q | k | v walked behind the macaron launch (EmFfnRowsArgs.post_q; ESPNET_AMD_NO_ROWS_QKV=1: developer switch)
  bool rows_qkv = ffn_rows && attn2 && h == 8 && !em_sw().no_rows_qkv;
  for (int l = 0; rows_qkv && l < L; ++l) rows_qkv = ly[l].wqkvp != nullptr;
  auto ffn_fused = [&](const void* w1p, const void* w2p, const float* b1, const float* b2, int ln_mode, const float* g1,
                       const float* be1, const float* g2, const float* be2, void* xn_out, float* out_f32,
                       const EmEBranchformerLayer* qkv_of = nullptr) {
    EmFfnRowsArgs fa = {};
    if (qkv_of) {
      fa.post_w = qkv_of->wqkvp; fa.post_b = qkv_of->bqkv; fa.post_chunks = 12;
      fa.post_q = qh; fa.post_k = ws + s.kh; fa.post_vt = vt; fa.post_T = T; fa.post_Tpad = s.Tpad;
    }
    if (ln_in) {  // the module's input = LayerNorm(x; norm_ff) computed in the launch
      fa.pre_g = ln_in->norm_ff_g; fa.pre_be = ln_in->norm_ff_b;
    }
    fa.xn_in = ln_in ? nullptr : xn; fa.x = x; fa.w1p = w1p; fa.w2p = w2p; fa.b1 = b1; fa.b2 = b2;
    fa.g1 = g1; fa.be1 = be1; fa.g2 = g2; fa.be2 = be2; fa.xn_out = xn_out; fa.out_f32 = out_f32;
    fa.M = M; fa.d = d; fa.ff = ff; fa.ln_mode = ln_mode; fa.scale = 0.5f; fa.eps = LN_EPS;
    return em_ffn_rows_fused(&fa, stream);
  };
  if (ffn)
    EM_TRY(em_layernorm(dtype, x, ly[0].norm_ff_mac_g, ly[0].norm_ff_mac_b, M, d, LN_EPS, xn, nullptr, stream));
  for (int l = 0; l < L; ++l) {
    const EmEBranchformerLayer& q = ly[l];
    if (ffn_rows) {
      // macaron FFN + residual + norm_mha (the attention branch's LayerNorm) in one launch
      // (with the q | k | v walk norm_mha's result never leaves the launch, and norm_mlp - the cgMLP branch's LayerNorm of the
      // same rows - leaves in its place: one set of statistics, two affine maps, no LayerNorm launch)
      if (rows_qkv)
        EM_TRY(ffn_fused(q.ffm_w1p, q.ffm_w2p, q.ffm_b1, q.ffm_b2, 1, q.norm_mha_g, q.norm_mha_b, q.norm_mlp_g, q.norm_mlp_b, xn2,
                         nullptr, &q));
      else
        EM_TRY(ffn_fused(q.ffm_w1p, q.ffm_w2p, q.ffm_b1, q.ffm_b2, 1, q.norm_mha_g, q.norm_mha_b, nullptr, nullptr, xn, nullptr));
    } else {
      if (ffn) {
        // macaron FFN (:132-135): x += 0.5 * w2(swish(w1 LN(x)))
        EM_TRY(gemm(dtype, EM_EPI_SWISH, xn, q.ffm_w1, big, q.ffm_b1, M, ff, d, d, ff, 1.f, stream));
        EM_TRY(gemm(dtype, EM_EPI_RESID_F32, big, q.ffm_w2, x, q.ffm_b2, M, d, ff, ff, d, 0.5f, stream));
      }
      // the two branches read the same x (:138-139)
      EM_TRY(em_layernorm(dtype, x, q.norm_mha_g, q.norm_mha_b, M, d, LN_EPS, xn, nullptr, stream));
    }
    if (!(ffn_rows && rows_qkv)) EM_TRY(em_layernorm(dtype, x, q.norm_mlp_g, q.norm_mlp_b, M, d, LN_EPS, xn2, nullptr, stream));
    // branch 1 (:141-152): rel-pos self-attention; linear_out lands in cat[:, :d]
    if (attn2) {
      if (!rows_qkv) {
        EmGemmArgs a = {};
        a.A = xn; a.W = q.wqkv; a.C = qh; a.bias = q.bqkv;
        a.M = M; a.N = 2 * d; a.K = d; a.lda = d; a.ldc = 64; a.scale = 1.f;
        a.T1 = T; a.T2 = s.Tpad; a.F1 = h; a.d = d;
        EM_TRY(em_gemm(dtype, EM_EPI_QK_HEADS, EM_A_PLAIN, &a, stream));
        a.A = (const unsigned char*)q.wqkv + (size_t)2 * d * d * es; a.W = xn; a.C = vt; a.bias = q.bqkv + 2 * d;
        a.M = d; a.N = M; a.ldc = s.Tpad;
        EM_TRY(em_gemm(dtype, EM_EPI_VT_HEADS, EM_A_PLAIN, &a, stream));
      }
      EM_TRY(em_relpos_attention2_bf16(qh, ws + s.kh, vt, (const unsigned char*)pall + (size_t)l * d * es, L * d, q.pos_u,
                                       q.pos_v, olens, B, T, s.Tpad, h, ctx, stream));
    } else {
      EM_TRY(gemm(dtype, EM_EPI_STORE, xn, q.wqkv, big, q.bqkv, M, 3 * d, d, d, 3 * d, 1.f, stream));
      EM_TRY((w->legacy_relpos ? em_legacy_relpos_attention : em_relpos_attention)(dtype, big, (const unsigned char*)pall + (size_t)l * d * es, L * d, q.pos_u,
                                 q.pos_v, olens, B, T, h, 64, ctx, stream));
    }
    EM_TRY(gemm(dtype, EM_EPI_STORE, ctx, q.wout, cat, q.bout, M, d, d, d, 2 * d, 1.f, stream));
    // branch 2 (:154-163): cgMLP = Linear + GELU -> [r | g]; g <- LN(g); r * (dwconv(g) + b) -> Linear,
    // which lands in cat[:, d:]
    EM_TRY(gemm(dtype, EM_EPI_GELU, xn2, q.proj1_w, big, q.proj1_b, M, cg, d, d, cg, 1.f, stream));
    // (the LayerNorm of the gate half is applied by the conv's input stage from per-row statistics:
    // the normalised half is never written)
    EM_TRY(em_row_stats(dtype, (const unsigned char*)big + (size_t)ch * es, cg, M, ch, LN_EPS, (float*)gn, stream));
    EM_TRY(em_dwconv_ln_gate(dtype, (const unsigned char*)big + (size_t)ch * es, cg, (const float*)gn,
                             q.csgu_norm_g, q.csgu_norm_b, q.csgu_conv_w, q.csgu_conv_b, conv_lens, B, T, ch,
                             w->cg_kernel, big, cg, gated, ch, stream));
    EM_TRY(gemm(dtype, EM_EPI_STORE, gated, q.proj2_w, cat + (size_t)d * es, q.proj2_b, M, d, ch, ch, 2 * d,
                1.f, stream));
    // merge: E-Branchformer (:165-170) x += merge_proj(cat + dwconv(cat)); Branchformer (branchformer_encoder.py:
    // 207-278) concat: x += merge_proj(cat); fixed_ave: the same launch, the host packs merge_proj as
    // [(1-w) W | w W]; learned_ave: x += merge_proj(w1 x1 + w2 x2) with per-utterance pooled weights
    if (w->merge_method == EM_MERGE_LEARNED_AVE) {
      if (!q.pool_w || !q.pool_b || !q.wproj_w || !q.wproj_b) return EM_ERR_BAD_ARG;
      EM_TRY(em_branch_learned_ave(dtype, cat, olens, B, T, d, q.pool_w, q.pool_b, q.wproj_w, q.wproj_b,
                                   (float*)(ws + s.mw), tmp, stream));
      EM_TRY(gemm(dtype, EM_EPI_RESID_F32, tmp, q.merge_w, x, q.merge_b, M, d, d, d, d, 1.f, stream));
    } else {
      const void* merged = cat;
      if (w->merge_conv) {
        EM_TRY(em_dwconv(dtype, EM_DW_SELFRES, cat, 2 * d, q.merge_conv_w, q.merge_conv_b, conv_lens, B, T, 2 * d,
                         w->merge_kernel, nullptr, 0, tmp, 2 * d, stream));
        merged = tmp;
      }
      EM_TRY(gemm(dtype, EM_EPI_RESID_F32, merged, q.merge_w, x, q.merge_b, M, d, 2 * d, 2 * d, d, 1.f, stream));
    }
    if (ffn_rows) {
      // norm_ff + FFN (:172-176) + residual + norm_final (:178) + the next consumer's LayerNorm in one launch (norm_ff in its
      // prologue, EmFfnRowsArgs.pre_g / pre_be without a projection; ESPNET_AMD_NO_ROWS_QKV=1 also keeps the LayerNorm launch)
      ln_in = !em_sw().no_rows_qkv ? &q : nullptr;
      if (!ln_in) EM_TRY(em_layernorm(dtype, x, q.norm_ff_g, q.norm_ff_b, M, d, LN_EPS, xn, nullptr, stream));
      if (l + 1 < L)
        EM_TRY(ffn_fused(q.ff_w1p, q.ff_w2p, q.ff_b1, q.ff_b2, 2, q.norm_final_g, q.norm_final_b, ly[l + 1].norm_ff_mac_g,
                         ly[l + 1].norm_ff_mac_b, xn, nullptr));
      else
        EM_TRY(ffn_fused(q.ff_w1p, q.ff_w2p, q.ff_b1, q.ff_b2, 2, q.norm_final_g, q.norm_final_b, w->after_norm_g, w->after_norm_b,
                         enc_act, enc_out));
      ln_in = nullptr;
      continue;
    }
    if (ffn) {
      // FFN (:172-176)
      EM_TRY(em_layernorm(dtype, x, q.norm_ff_g, q.norm_ff_b, M, d, LN_EPS, xn, nullptr, stream));
      EM_TRY(gemm(dtype, EM_EPI_SWISH, xn, q.ff_w1, big, q.ff_b1, M, ff, d, d, ff, 1.f, stream));
      EM_TRY(gemm(dtype, EM_EPI_RESID_F32, big, q.ff_w2, x, q.ff_b2, M, d, ff, ff, d, 0.5f, stream));
    }
    // norm_final (:178), fused with the next consumer's LayerNorm where there is exactly one
    if (l + 1 == L)
      EM_TRY(em_layernorm2(dtype, x, q.norm_final_g, q.norm_final_b, w->after_norm_g, w->after_norm_b, M, d,
                           LN_EPS, enc_act, enc_out, stream));
    else if (ffn)
      EM_TRY(em_layernorm2(dtype, x, q.norm_final_g, q.norm_final_b, ly[l + 1].norm_ff_mac_g,
                           ly[l + 1].norm_ff_mac_b, M, d, LN_EPS, xn, nullptr, stream));
    else
      EM_TRY(em_layernorm_inplace_f32(x, q.norm_final_g, q.norm_final_b, M, d, LN_EPS, stream));
  }
  return EM_OK;
}
