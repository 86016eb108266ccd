// Host-side launch sequence of the E-Branchformer encoder (no device code here): one C call
// enqueues every kernel of a forward pass on the caller's stream.
//
// Reference call graph being reproduced:
//   EBranchformerEncoder.forward       espnet2/asr/encoder/e_branchformer_encoder.py:418-520
//   EBranchformerEncoderLayer.forward  espnet2/asr/encoder/e_branchformer_encoder.py:110-183
//   ConvolutionalGatingMLP / CSGU      espnet2/asr/layers/cgmlp.py:14-145
//   Conv2dSubsampling, RelPositionalEncoding, RelPositionMultiHeadedAttention, PositionwiseFeedForward:
//   the Conformer ones (see encoder.hip)
//
// Per layer: macaron FFN -> two branches on the same x (rel-pos attention | cgMLP) whose outputs are
// written side by side into one [M][2d] matrix by their last GEMMs (ldc = 2d) -> depthwise conv over
// the 2d channels with the self residual fused -> merge_proj GEMM into the f32 residual stream ->
// FFN -> norm_final (fused with the next layer's first LayerNorm).
#include <math.h>

#include "em_common.h"
#include "subsample.h"

namespace {

constexpr float LN_EPS = 1e-12f;

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Ws {
  size_t c1, c2, c3, x, xn, xn2, big, gn, gated, cat, tmp, ctx, pall, total;
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

}  // namespace

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
  EM_TRY(em_sub::run(dtype, w, g, feats, mvn_partial, flens, B, c1, c2, c3, x, stream));
  EM_TRY(gemm(dtype, EM_EPI_STORE, pos_emb, w->wpos_all, pall, nullptr, w->legacy_relpos ? T : 2 * T - 1, L * d, d, d, L * d, 1.f,
              stream));
  const EmEBranchformerLayer* ly = w->layers;
  const bool ffn = w->use_ffn != 0;
  if (ffn)
    EM_TRY(em_layernorm(dtype, x, ly[0].norm_ff_mac_g, ly[0].norm_ff_mac_b, M, d, LN_EPS, xn, nullptr, stream));
  for (int l = 0; l < L; ++l) {
    const EmEBranchformerLayer& q = ly[l];
    if (ffn) {
      // macaron FFN (:132-135): x += 0.5 * w2(swish(w1 LN(x)))
      EM_TRY(gemm(dtype, EM_EPI_SWISH, xn, q.ffm_w1, big, q.ffm_b1, M, ff, d, d, ff, 1.f, stream));
      EM_TRY(gemm(dtype, EM_EPI_RESID_F32, big, q.ffm_w2, x, q.ffm_b2, M, d, ff, ff, d, 0.5f, stream));
    }
    // the two branches read the same x (:138-139)
    EM_TRY(em_layernorm(dtype, x, q.norm_mha_g, q.norm_mha_b, M, d, LN_EPS, xn, nullptr, stream));
    EM_TRY(em_layernorm(dtype, x, q.norm_mlp_g, q.norm_mlp_b, M, d, LN_EPS, xn2, nullptr, stream));
    // branch 1 (:141-152): rel-pos self-attention; linear_out lands in cat[:, :d]
    EM_TRY(gemm(dtype, EM_EPI_STORE, xn, q.wqkv, big, q.bqkv, M, 3 * d, d, d, 3 * d, 1.f, stream));
    EM_TRY((w->legacy_relpos ? em_legacy_relpos_attention : em_relpos_attention)(dtype, big, (const unsigned char*)pall + (size_t)l * d * es, L * d, q.pos_u,
                               q.pos_v, olens, B, T, h, 64, ctx, stream));
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
    // merge: E-Branchformer (:165-170) x += merge_proj(cat + dwconv(cat)); Branchformer (concat,
    // branchformer_encoder.py:207-211) x += merge_proj(cat)
    const void* merged = cat;
    if (w->merge_conv) {
      EM_TRY(em_dwconv(dtype, EM_DW_SELFRES, cat, 2 * d, q.merge_conv_w, q.merge_conv_b, conv_lens, B, T, 2 * d,
                       w->merge_kernel, nullptr, 0, tmp, 2 * d, stream));
      merged = tmp;
    }
    EM_TRY(gemm(dtype, EM_EPI_RESID_F32, merged, q.merge_w, x, q.merge_b, M, d, 2 * d, 2 * d, d, 1.f, stream));
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
