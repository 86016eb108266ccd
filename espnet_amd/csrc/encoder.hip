// Host-side launch sequence of the Conformer encoder and the greedy CTC head (no device code
// here): one C call enqueues every kernel of a forward pass on the caller's stream, so the Python
// layer pays one FFI crossing per batch.
//
// Reference call graph being reproduced:
//   ConformerEncoder.forward          espnet2/asr/encoder/conformer_encoder.py:327-429
//   Conv2dSubsampling.forward         .../transformer/subsampling.py:432-447
//   RelPositionalEncoding.forward     .../transformer/embedding.py:318-334  (x * sqrt(d))
//   EncoderLayer.forward (x12)        .../conformer/encoder_layer.py:79-179
//   CTC.argmax + G1 collapse          espnet2/asr/ctc.py:207-215, bin/asr_inference.py:574-575
#include <math.h>
#include <stdlib.h>

#include "em_common.h"
#include "switches.h"
#include "subsample.h"

namespace {

constexpr float LN_EPS = 1e-12f;  // transformer/layer_norm.py:23

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Ws {
  size_t c1, c2, c3, x, xn, big, g, g2, ctx, pall, qh, kh, vt, ppk, total;
  int Tpad;
};
inline int fused_tpad(int T) { return (T + 255) / 256 * 256; }
inline Ws layout(int dtype, const EmConformerWeights* w, int B, int T_f) {
  const size_t es = dtype == EM_BF16 ? 2 : 4;
  em_sub::Geo g;
  em_sub::geo(w->subsample, T_f, w->n_mels, &g);
  const size_t M = (size_t)B * g.T_out, d = w->d;
  size_t mb[3];
  em_sub::map_bytes(g, B, w->d, es, mb);
  size_t wide = w->ff > 3 * w->d ? w->ff : 3 * w->d;
  Ws s;
  size_t o = 0;
  s.c1 = o; o += align_up(mb[0]);
  s.c2 = o; o += align_up(mb[1]);
  s.c3 = o; o += align_up(mb[2]);
  s.x = o; o += align_up(M * d * 4);
  s.xn = o; o += align_up(M * d * es);
  s.big = o; o += align_up(M * wide * es);
  s.g = o; o += align_up(M * d * es);
  s.g2 = o; o += align_up(M * d * es);
  s.ctx = o; o += align_up(M * d * es);
  s.pall = o; o += align_up((size_t)(2 * g.T_out - 1) * w->num_blocks * d * es);
  // fused path (csrc/block.hip, csrc/attention2.hip): Q, K [B][H][Tpad][64], V^T [B][H][64][Tpad]
  s.Tpad = fused_tpad(g.T_out);
  const size_t per_head = (size_t)B * d * s.Tpad * es;
  s.qh = o; o += align_up(per_head);
  s.kh = o; o += align_up(per_head);
  s.vt = o; o += align_up(per_head);
  // block<ATT|C> (round 6): the position rows of every block, fragment-major (em_relpos_pack_pos_bf16)
  s.ppk = o; o += align_up((size_t)w->num_blocks * 4 * em_relpos_pos_fragments(g.T_out) * 2048);
  s.total = o;
  return s;
}

inline int gemm(int dtype, int epi, const void* A, const void* W, void* C, const float* bias, int M,
                int N, int K, int lda, int ldc, float scale, void* stream) {
  EmGemmArgs a = {};
  a.A = A; a.W = W; a.C = C; a.bias = bias;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.scale = scale;
  a.T1 = a.F1 = a.T2 = a.F2 = a.d = 0;
  return em_gemm(dtype, epi, EM_A_PLAIN, &a, stream);
}


}  // namespace

// A row-block launch takes ~80 us per ROUND of 64-row workgroups whatever the number of rows (one workgroup streams all
// 4 MiB of the module's weights through its CU), the launches it replaces scale with M (90 us at M = 15 936, ~31 us at
// the beam search's M = 3 984, where only 63 CUs would work: profiles/r04r_search_kernel_stats.csv): taken only when its
// rounds fill at least 72 % of the chip - or 72 % / n when the caller keeps n batches in flight (EM_ENC_IN_FLIGHT): a launch
// that fills half the chip runs side by side with the other stream's (round 6, profiles/r06ae_rows_fill_ab.txt: Conformer-
// large at B = 32, 125 workgroups: 3.84 against 3.72 ms per batch alone, 2.63 against 3.28 with two in flight; B = 16 with
// three in flight 1.52 against 1.71).  ESPNET_AMD_FFN_ROWS_MIN_FILL: developer switch (percent).
// (per DEVICE, not per process: a process may decode on MI355X partitions with different CU counts)
bool em_rows_fill_ok(long M, int flags) {
  static int n_cu_of[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (n_cu_of[dev] == 0) {
    hipDeviceProp_t prop;
    n_cu_of[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  const int n_cu = n_cu_of[dev];
  int n = (flags >> 8) & 15;
  n = n < 1 ? 1 : n;
  const int forced = em_sw().ffn_rows_min_fill;
  const long fill = forced > 0 ? forced : 72 / n;
  const long wgs = (M + 63) / 64, rounds = (wgs + n_cu - 1) / n_cu;
  return M > 0 && 100 * wgs >= fill * rounds * n_cu;
}

namespace {

// The 512-wide model's row-block launches (csrc/ffn_rows.hip).  Round 4: at d = 512 (bf16) the block's row-local operators
// run as three launches of 64-row workgroups when the host packed their operands: [macaron FFN + residual + norm_mha],
// [linear_out + residual + norm_conv + pointwise_conv1 + GLU], [pointwise_conv2 + residual + norm_ff + FFN + residual +
// norm_final + the next LayerNorm] (+ the CTC head's arg-max behind the last one).  ESPNET_AMD_NO_FFN_ROWS=1: developer switch.
struct RowsPlan {
  bool ffn, pre_pw2, glu, ctc;
};
inline RowsPlan rows_plan(int dtype, const EmConformerWeights* w, int flags, long M) {
  RowsPlan r = {false, false, false, false};
  const bool no_ffn_rows = em_sw().no_ffn_rows;
  const int d = w->d, ff = w->ff, L = w->num_blocks;
  const EmConformerLayer* ly = w->layers;
  if (!(dtype == EM_BF16 && d == 512 && ff % 128 == 0 && ff >= 256 && !(flags & EM_ENC_NO_FUSED) && !no_ffn_rows && ly && L > 0 &&
        M > 0))
    return r;
  r.ffn = em_rows_fill_ok(M, flags);
  for (int l = 0; r.ffn && l < L; ++l) r.ffn = ly[l].ffm_w1p && ly[l].ffm_w2p && ly[l].ff_w1p && ly[l].ff_w2p;
  // ... pointwise_conv2 + residual + norm_ff ride in the second module's launch when the host packed pw2 as well
  r.pre_pw2 = r.ffn;
  for (int l = 0; r.pre_pw2 && l < L; ++l) r.pre_pw2 = ly[l].pw2p != nullptr;
  // ... linear_out + residual + norm_conv + pointwise_conv1 + GLU are one launch (EM_ROWS_GLU) with woutp / pw1f / fp_c
  // (= pointwise_conv1's bias in the chunk order) packed
  r.glu = r.ffn;
  for (int l = 0; r.glu && l < L; ++l) r.glu = ly[l].woutp && ly[l].pw1f && ly[l].fp_c;
  // ... and the CTC head's arg-max is a walk behind the last launch (ctc_w: ctc_lo.weight in 128-row chunks, w1p layout)
  const bool no_rows_ctc = em_sw().no_rows_ctc;  // developer A/B switch
  r.ctc = r.ffn && !no_rows_ctc && w->ctc_ids && w->ctc_w && w->ctc_b && w->ctc_units > 0;
  return r;
}

// Which launch sequence em_conformer_encode takes: the ONE place that decides (the host layer asks through
// em_conformer_encode_plan instead of re-deriving the shape conditions).
inline int encode_plan(int dtype, const EmConformerWeights* w, int flags, long M = 0) {
  if (rows_plan(dtype, w, flags, M).ctc) return EM_ENC_PLAN_CTC_IDS;
  const int d = w->d, h = w->heads, ff = w->ff, L = w->num_blocks;
  const EmConformerLayer* ly = w->layers;
  bool fused = dtype == EM_BF16 && d == 256 && h == 4 && ff <= 1024 && w->kernel == 31 && !w->legacy_relpos &&
               !(flags & EM_ENC_NO_FUSED) && ly && L > 0 && ly[0].fp_a != nullptr;
  for (int l = 0; fused && l < L; ++l)
    fused = ly[l].pw1f && ly[l].fp_c && ly[l].fp_da && ly[l].ffm_w2p && ly[l].ff_w2p && ly[l].woutp && ly[l].pw2p &&
            ly[l].ff_w1p && ly[l].ffm_w1p && ly[l].wqkvp;
  if (!fused) return 0;
  const bool ctc = w->ctc_ids && w->ctc_w && w->ctc_b && w->ctc_units > 0;
  return EM_ENC_PLAN_FUSED | (ctc ? EM_ENC_PLAN_CTC_IDS : 0);
}

#define EM_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != EM_OK) return rc__; \
  } while (0)

}  // namespace

extern "C" size_t em_conformer_workspace_bytes(int dtype, const EmConformerWeights* w, int32_t B,
                                               int32_t T_f) {
  if (!w || B <= 0 || T_f < em_sub::min_frames(w->subsample)) return 0;
  return layout(dtype, w, B, T_f).total;
}

extern "C" int em_conformer_encode_plan(int dtype, const EmConformerWeights* w, int32_t flags) {
  if (!w || (dtype != EM_F32 && dtype != EM_BF16)) return EM_ERR_BAD_ARG;
  return encode_plan(dtype, w, flags);
}

extern "C" int em_conformer_encode_plan_for(int dtype, const EmConformerWeights* w, int32_t flags, int32_t B, int32_t T_f) {
  if (!w || (dtype != EM_F32 && dtype != EM_BF16) || B <= 0) return EM_ERR_BAD_ARG;
  em_sub::Geo g;
  if (T_f < em_sub::min_frames(w->subsample) || !em_sub::geo(w->subsample, T_f, w->n_mels, &g)) return EM_ERR_BAD_ARG;
  return encode_plan(dtype, w, flags, (long)B * g.T_out);
}

extern "C" int em_conformer_encode(int dtype, const EmConformerWeights* w, const float* feats,
                                   const float* mvn_partial, const int32_t* flens,
                                   const int32_t* olens, int32_t B, int32_t T_f,
                                   const void* pos_emb, void* workspace, size_t workspace_bytes,
                                   float* enc_out, void* enc_act, int32_t flags, void* stream) {
  // EM_ENC_ISOLATE_UTTS: the depthwise conv treats frames >= olens[b] as zero (see conv.hip)
  const int32_t* conv_lens = (flags & EM_ENC_ISOLATE_UTTS) ? olens : nullptr;
  if (!w || !feats || !flens || !olens || !pos_emb || !workspace || !enc_out || !enc_act)
    return EM_ERR_BAD_ARG;
  if (dtype != EM_F32 && dtype != EM_BF16) return EM_ERR_BAD_ARG;
  if (B <= 0) return EM_ERR_BAD_ARG;
  if (T_f < em_sub::min_frames(w->subsample)) return EM_ERR_TOO_SHORT;  // check_short_utt, subsampling.py:31-49
  const int d = w->d, h = w->heads, ff = w->ff, L = w->num_blocks;
  if (d % 64 != 0 || h <= 0 || d / h != 64 || ff % 64 != 0) return EM_ERR_UNSUPPORTED;
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
  void* big = ws + s.big;
  void* gl = ws + s.g;
  void* g2 = ws + s.g2;
  void* ctx = ws + s.ctx;
  void* pall = ws + s.pall;
  const size_t es = dtype == EM_BF16 ? 2 : 4;

  // ---- Conv2dSubsampling{,6,8}: conv1 (+MVN) -> implicit-GEMM conv(s) -> Linear, * sqrt(d)  (subsample.h)
  EM_TRY(em_sub::run(dtype, w, g, feats, mvn_partial, flens, B, c1, c2, c3, x, stream, w->conv1_wf, w->conv2_wf));
  const EmConformerLayer* ly = w->layers;
  // ---- linear_pos of every block in one GEMM: pall[2T-1][L*d]
  // (EM_ENC_POS_PROJECTED: the caller did it once for this length - the product depends on T and the weights only)
  if (flags & EM_ENC_POS_PROJECTED)
    pall = const_cast<void*>(pos_emb);
  else
    EM_TRY(gemm(dtype, EM_EPI_STORE, pos_emb, w->wpos_all, pall, nullptr, w->legacy_relpos ? T : 2 * T - 1, L * d, d, d,
                L * d, 1.f, stream));
  // ---- fused per Conformer block (csrc/block.hip): three launches per block instead of nineteen
  const int plan = encode_plan(dtype, w, flags);
  const bool fused = (plan & EM_ENC_PLAN_FUSED) != 0;
  if (fused) {
    void* qh = ws + s.qh;
    void* kh = ws + s.kh;
    void* vt = ws + s.vt;
    // key columns >= 32 * ceil(T / 32) of V^T are never written and meet probability 0 in P.V: they must be finite.
    // (When the 32-frame row blocks cover Tpad - T = 249: 8 x 32 = 256 - every column is written by the block kernels,
    // frames past T as recomputed copies of frame T - 1: no memset, 5 us of a 1.07 ms step.)
    if (32 * em_cdiv(T, 32) < s.Tpad &&
        hipMemsetAsync(vt, 0, (size_t)B * d * s.Tpad * es, (hipStream_t)stream) != hipSuccess)
      return EM_ERR_LAUNCH;
    EmBlockArgs ba = {};
    ba.B = B; ba.T = T; ba.Tpad = s.Tpad; ba.d = d; ba.ff = ff; ba.kernel = w->kernel; ba.eps = LN_EPS;
    ba.x = x; ba.ctx = ctx; ba.glu = gl; ba.qh = qh; ba.kh = kh; ba.vt = vt;
    ba.enc_out = enc_out; ba.enc_act = enc_act; ba.tlens = conv_lens;
    auto set_a = [&](const EmConformerLayer& q) { ba.ffm_w1 = q.ffm_w1p; ba.ffm_w2 = q.ffm_w2p; ba.wqkv = q.wqkvp; };
    // Round 6: attention + the C part of a block as ONE launch, block<ATT|C> (EM_ENC_SPLIT_ATT / EM_ENC_FOLD_C: the forms of
    // rounds 2-5).  Its K / V^T / position operands are fragment-major: packed position rows here, kv_frag for the A parts.
    const bool att_c = !(flags & EM_ENC_FOLD_C) && !(flags & EM_ENC_SPLIT_ATT);
    void* ppk = ws + s.ppk;
    const int npg = em_relpos_pos_fragments(T);
    if (att_c) {
      if ((flags & EM_ENC_POS_PROJECTED) && (flags & EM_ENC_POS_PACKED))  // the caller packed them once for this length
        ppk = (unsigned char*)const_cast<void*>(pos_emb) + align_up((size_t)(2 * T - 1) * L * d * es);
      else
        EM_TRY(em_relpos_pack_pos_bf16(pall, L * d, T, L, ppk, stream));
      ba.kv_frag = 1;  // the A parts write K and V^T in the order the attention's MFMAs take them
    }
    set_a(ly[0]);
    ba.params = ly[0].fp_a;
    EM_TRY(em_conformer_block_fused(EM_BLOCK_A, &ba, stream));
    // Round 4: the C part runs inside the launch that consumes it (block<C|D|...>: two launches per Conformer block).
    // The residual stream then ping-pongs between `x` and the idle `big` slot of the workspace, because a workgroup
    // reads its neighbours' rows of x as the depthwise conv's halo.
    // Measured (profiles/r04c_*, r04d_*: one box, in-call A/B): correct (53 kernel tests, every bf16 golden) and NOT
    // faster - block<C|D|A> 54.1 us against 40.2 + 12.4 for the two launches, step 1.086 vs 1.066 ms.  The halo doubles
    // the C part's prologue (ctx + x of 64 frames: 15 K cycles to issue 300 KiB per CU against a memory system that
    // serves every CU's cold burst at ~13 B/clk), which costs what the saved launch boundary gains.  So it is OPT-IN:
    // flag EM_ENC_FOLD_C (the host layer sets it from ESPNET_AMD_FOLD=1 / `encoder.fold_c`).
    const bool no_fold = !(flags & EM_ENC_FOLD_C);
    const int cbit = no_fold ? 0 : EM_BLOCK_C;
    float* xa = x;
    float* xb = (float*)big;
    for (int l = 0; l < L; ++l) {
      const EmConformerLayer& q = ly[l];
      ba.wout = q.woutp; ba.pw1f = q.pw1f;
      if (att_c) {
        // Round 6: attention + the C part as ONE launch per (utterance, 32 queries) - block<ATT|C>, csrc/block.hip; the
        // context never exists in memory.  EM_ENC_SPLIT_ATT: the two launches of rounds 2-5 (developer A/B switch).
        ba.pos = (const unsigned char*)ppk + (size_t)l * 4 * npg * 2048; ba.ldp = npg; ba.pos_u = q.pos_u; ba.pos_v = q.pos_v;
        ba.klens = olens; ba.params = q.fp_c;
        EM_TRY(em_conformer_block_fused(EM_BLOCK_ATT | EM_BLOCK_C, &ba, stream));
      } else {
        EM_TRY(em_relpos_attention2_bf16(qh, kh, vt, (const unsigned char*)pall + (size_t)l * d * es, L * d, q.pos_u,
                                         q.pos_v, olens, B, T, s.Tpad, h, ctx, stream));
      }
      if (att_c) {
        // (done above)
      } else if (no_fold) {
        ba.params = q.fp_c;
        EM_TRY(em_conformer_block_fused(EM_BLOCK_C, &ba, stream));
      } else {
        ba.params_c = q.fp_c; ba.x = xa; ba.x_out = xb;
      }
      ba.pw2 = q.pw2p; ba.ff_w1 = q.ff_w1p; ba.ff_w2 = q.ff_w2p; ba.dw_w = q.dw_w; ba.dw_b = q.dw_b;
      ba.params = q.fp_da;
      if (l + 1 < L) {
        set_a(ly[l + 1]);
        EM_TRY(em_conformer_block_fused(cbit | EM_BLOCK_D | EM_BLOCK_A, &ba, stream));
        if (!no_fold) { float* t = xa; xa = xb; xb = t; }
      } else if (plan & EM_ENC_PLAN_CTC_IDS) {
        ba.ctc_w = w->ctc_w; ba.ctc_b = w->ctc_b; ba.ctc_ids = w->ctc_ids; ba.ctc_units = w->ctc_units;
        EM_TRY(em_conformer_block_fused(cbit | EM_BLOCK_D | EM_BLOCK_FINAL | EM_BLOCK_CTC, &ba, stream));
      } else {
        EM_TRY(em_conformer_block_fused(cbit | EM_BLOCK_D | EM_BLOCK_FINAL, &ba, stream));
      }
    }
    return EM_OK;
  }

  // ---- one operator per launch (f32 parity mode, the 512-wide model, legacy rel-pos, EM_ENC_NO_FUSED).  Two
  // earlier fusions of this sequence -- LayerNorm in a whole-row GEMM epilogue and a row-block fused FFN --
  // measured slower than what follows and live on only as records (tools/experiments/gemm_ln_epilogue.hip.txt,
  // ffn_fused_rowblock.hip.txt).
  // Round 4: wherever d_k = 64 and the dtype is bf16 (the large model, 8 heads) the attention runs in the LDS-resident
  // kernel of the fused path (csrc/attention2.hip: 0.11 of the MFMA peak against 0.05 for relpos_attn_kernel, which takes
  // 90 us per block at B = 64): its per-head operands are written by the projection GEMMs themselves - q | k through
  // EM_EPI_QK_HEADS, V^T as the swapped product W_v . xn^T through EM_EPI_VT_HEADS - so there is no repacking pass.
  // ESPNET_AMD_NO_ATTN2_LARGE=1: developer A/B switch.
  const bool no_attn2 = em_sw().no_attn2_large;
  const bool attn2 = dtype == EM_BF16 && !w->legacy_relpos && !(flags & EM_ENC_NO_FUSED) && !no_attn2 && d == 64 * h &&
                     (size_t)B * d * s.Tpad * 4 < ((size_t)1 << 32) - 64;
  void* qh = ws + s.qh;
  void* vt = ws + s.vt;
  if (attn2) {
    // frames >= T of every (b, head) slab must be finite: keys / probabilities there are masked, not skipped
    if (hipMemsetAsync(qh, 0, (s.vt - s.qh) + (size_t)B * d * s.Tpad * es, (hipStream_t)stream) != hipSuccess)
      return EM_ERR_LAUNCH;
  }
  const RowsPlan rp = rows_plan(dtype, w, flags, M);
  const bool ffn_rows = rp.ffn, pre_pw2 = rp.pre_pw2, rows_glu = rp.glu;
  const EmConformerLayer* pre_layer = nullptr;  // set for the call that carries the projection
  const void* const conv_out = g2;              // the depthwise conv's output: the projection's input
  // round 6: q | k | v walked behind the macaron launch (EmFfnRowsArgs.post_q): no projection GEMMs, LN(x) never stored.
  // ESPNET_AMD_NO_ROWS_QKV=1: developer A/B switch.
  bool rows_qkv = ffn_rows && attn2 && d == 512 && h == 8 && !em_sw().no_rows_qkv;
  for (int l = 0; rows_qkv && l < L; ++l) rows_qkv = ly[l].wqkvp != nullptr;
  auto ffn_fused = [&](const void* w1p, const void* w2p, const float* b1, const float* b2, int ln_mode, const float* g1,
                       const float* be1, const float* g2, const float* be2, void* xn_out, float* out_f32,
                       const EmConformerLayer* qkv_of = nullptr) {
    EmFfnRowsArgs fa = {};
    if (qkv_of) {
      fa.post_w = qkv_of->wqkvp; fa.post_b = qkv_of->bqkv; fa.post_chunks = 12;
      fa.post_q = qh; fa.post_k = ws + s.kh; fa.post_vt = vt; fa.post_T = T; fa.post_Tpad = s.Tpad;
    }
    if (pre_layer) {
      fa.pre_in = conv_out; fa.pre_w = pre_layer->pw2p; fa.pre_b = pre_layer->pw2_b;
      fa.pre_g = pre_layer->norm_ff_g; fa.pre_be = pre_layer->norm_ff_b;
    }
    fa.xn_in = xn; fa.x = x; fa.w1p = w1p; fa.w2p = w2p; fa.b1 = b1; fa.b2 = b2;
    fa.g1 = g1; fa.be1 = be1; fa.g2 = g2; fa.be2 = be2; fa.xn_out = xn_out; fa.out_f32 = out_f32;
    fa.M = M; fa.d = d; fa.ff = ff; fa.ln_mode = ln_mode; fa.scale = 0.5f; fa.eps = LN_EPS;
    if (out_f32 && rp.ctc) {  // the last launch of the stack: the CTC head's arg-max behind after_norm
      fa.post_w = w->ctc_w; fa.post_b = w->ctc_b; fa.post_ids = w->ctc_ids; fa.post_chunks = w->ctc_units;
      fa.post_vocab = w->ctc_units * 128;
    }
    return em_ffn_rows_fused(&fa, stream);
  };
  EM_TRY(em_layernorm(dtype, x, ly[0].norm_ff_mac_g, ly[0].norm_ff_mac_b, M, d, LN_EPS, xn, nullptr,
                      stream));
  for (int l = 0; l < L; ++l) {
    const EmConformerLayer& q = ly[l];
    pre_layer = nullptr;
    if (ffn_rows) {
      // macaron FFN + residual + norm_mha
      EM_TRY(ffn_fused(q.ffm_w1p, q.ffm_w2p, q.ffm_b1, q.ffm_b2, 1, q.norm_mha_g, q.norm_mha_b, nullptr, nullptr,
                       rows_qkv ? nullptr : xn, nullptr, rows_qkv ? &q : nullptr));  // (with the walk LN(x) is not stored)
    } else {
      // macaron FFN: x += 0.5 * w2(swish(w1 LN(x)))
      EM_TRY(gemm(dtype, EM_EPI_SWISH, xn, q.ffm_w1, big, q.ffm_b1, M, ff, d, d, ff, 1.f, stream));
      EM_TRY(gemm(dtype, EM_EPI_RESID_F32, big, q.ffm_w2, x, q.ffm_b2, M, d, ff, ff, d, 0.5f, stream));
      // self-attention
      EM_TRY(em_layernorm(dtype, x, q.norm_mha_g, q.norm_mha_b, M, d, LN_EPS, xn, nullptr, stream));
    }
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
      EM_TRY((w->legacy_relpos ? em_legacy_relpos_attention : em_relpos_attention)(dtype, big, (const unsigned char*)pall + (size_t)l * d * es, L * d,
                                 q.pos_u, q.pos_v, olens, B, T, h, 64, ctx, stream));
    }
    if (rows_glu) {  // linear_out + residual + norm_conv + pointwise_conv1 + GLU: one row-block launch
      EmFfnRowsArgs fa = {};
      fa.x = x; fa.w1p = q.pw1f; fa.b1 = q.fp_c; fa.xn_out = gl;
      fa.M = M; fa.d = d; fa.ff = 2 * d; fa.ln_mode = 1; fa.scale = 1.f; fa.eps = LN_EPS;
      fa.pre_in = ctx; fa.pre_w = q.woutp; fa.pre_b = q.bout; fa.pre_g = q.norm_conv_g; fa.pre_be = q.norm_conv_b;
      fa.main = EM_ROWS_GLU;
      EM_TRY(em_ffn_rows_fused(&fa, stream));
    } else {
      EM_TRY(gemm(dtype, EM_EPI_RESID_F32, ctx, q.wout, x, q.bout, M, d, d, d, d, 1.f, stream));
      // convolution module
      EM_TRY(em_layernorm(dtype, x, q.norm_conv_g, q.norm_conv_b, M, d, LN_EPS, xn, nullptr, stream));
      EM_TRY(gemm(dtype, EM_EPI_GLU, xn, q.pw1, gl, q.pw1_b, M, 2 * d, d, d, d, 1.f, stream));
    }
    EM_TRY(em_dwconv_bn_swish(dtype, gl, q.dw_w, q.dw_b, conv_lens, B, T, d, w->kernel, g2, stream));
    if (!pre_pw2) {
      EM_TRY(gemm(dtype, EM_EPI_RESID_F32, g2, q.pw2, x, q.pw2_b, M, d, d, d, d, 1.f, stream));
      // FFN
      EM_TRY(em_layernorm(dtype, x, q.norm_ff_g, q.norm_ff_b, M, d, LN_EPS, xn, nullptr, stream));
    }
    pre_layer = pre_pw2 ? &q : nullptr;
    if (ffn_rows) {  // FFN + residual + norm_final + the next consumer's LayerNorm
      if (l + 1 < L)
        EM_TRY(ffn_fused(q.ff_w1p, q.ff_w2p, q.ff_b1, q.ff_b2, 2, q.norm_final_g, q.norm_final_b, ly[l + 1].norm_ff_mac_g,
                         ly[l + 1].norm_ff_mac_b, xn, nullptr));
      else
        EM_TRY(ffn_fused(q.ff_w1p, q.ff_w2p, q.ff_b1, q.ff_b2, 2, q.norm_final_g, q.norm_final_b, w->after_norm_g,
                         w->after_norm_b, enc_act, enc_out));
      continue;
    }
    EM_TRY(gemm(dtype, EM_EPI_SWISH, xn, q.ff_w1, big, q.ff_b1, M, ff, d, d, ff, 1.f, stream));
    EM_TRY(gemm(dtype, EM_EPI_RESID_F32, big, q.ff_w2, x, q.ff_b2, M, d, ff, ff, d, 0.5f, stream));
    // norm_final, fused with the next consumer's LayerNorm
    if (l + 1 < L) {
      EM_TRY(em_layernorm2(dtype, x, q.norm_final_g, q.norm_final_b, ly[l + 1].norm_ff_mac_g,
                           ly[l + 1].norm_ff_mac_b, M, d, LN_EPS, xn, nullptr, stream));
    } else {
      EM_TRY(em_layernorm2(dtype, x, q.norm_final_g, q.norm_final_b, w->after_norm_g,
                           w->after_norm_b, M, d, LN_EPS, enc_act, enc_out, stream));
    }
  }
  return EM_OK;
}

extern "C" int em_ctc_greedy(int dtype, const void* enc_act, const void* w_ctc, const float* b_ctc,
                             int32_t B, int32_t T, int32_t d, int32_t V, const int32_t* olens,
                             int32_t blank, int32_t sos_eos, float* logits_ws, int32_t* ids,
                             int32_t* tokens, int32_t* out_lens, void* stream) {
  if (!enc_act || !w_ctc || !logits_ws || !ids || !tokens || !out_lens) return EM_ERR_BAD_ARG;
  const int M = B * T;
  // arg-max fused into the ctc_lo GEMM epilogue: the (M, V) f32 logits (160 MB at B = 32) are
  // never written; only 2*ceil(V/128) (value, column) partials per frame are
  const int G = 2 * em_cdiv(V, 128);
  EM_TRY(gemm(dtype, EM_EPI_ARGMAX_PART, enc_act, w_ctc, logits_ws, b_ctc, M, V, d, d, G, 1.f, stream));
  EM_TRY(em_argmax_partials(logits_ws, M, G, ids, stream));
  EM_TRY(em_ctc_collapse(ids, olens, B, T, blank, sos_eos, tokens, out_lens, stream));
  return EM_OK;
}
