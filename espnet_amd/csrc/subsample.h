// Conv2dSubsampling front of the encoders (host-side launch helpers, shared by encoder.hip and
// ebranchformer.hip).  Reference: espnet2/legacy/nets/pytorch_backend/transformer/subsampling.py
//   Conv2dSubsampling  (:386-470)  Conv2d(1,d,3,2)+ReLU, Conv2d(d,d,3,2)+ReLU, Linear(d*F2, d)      time / 4
//   Conv2dSubsampling6 (:692-775)  Conv2d(1,d,3,2)+ReLU, Conv2d(d,d,5,3)+ReLU, Linear               time / 6
//   Conv2dSubsampling8 (:785-870)  three Conv2d(.,d,3,2)+ReLU, Linear                               time / 8
// The first conv has one input channel (csrc/frontend.hip, utterance MVN folded in); every further conv
// is the implicit GEMM of gemm.hip (EM_A_CONV2 with kernel width / stride in the arguments) over the
// channel-last map of the previous one; the Linear is a GEMM with the x * sqrt(d) of the positional
// encoding (embedding.py:328) in its epilogue.
#pragma once
#include "switches.h"
#include <math.h>
#include <stdlib.h>

#include "em_common.h"

namespace em_sub {

struct Geo {
  int n;        // convs after the first one (1 or 2)
  int T[4], F[4];  // T[0], F[0] = feature frames / mel bins; level i+1 = output of conv i
  int k[3], s[3];
  int T_out, F_out;
};

inline int mode_of(int subsample) { return subsample == 0 ? 4 : subsample; }

inline int min_frames(int subsample) {  // check_short_utt, subsampling.py:31-49
  switch (mode_of(subsample)) {
    case 6: return 11;
    case 8: return 15;
    default: return 7;
  }
}

inline bool geo(int subsample, int T_f, int n_mels, Geo* g) {
  const int m = mode_of(subsample);
  if (m != 4 && m != 6 && m != 8) return false;
  g->k[0] = 3; g->s[0] = 2;
  g->k[1] = m == 6 ? 5 : 3; g->s[1] = m == 6 ? 3 : 2;
  g->k[2] = 3; g->s[2] = 2;
  const int convs = m == 8 ? 3 : 2;
  g->n = convs - 1;
  g->T[0] = T_f; g->F[0] = n_mels;
  for (int i = 0; i < convs; ++i) {
    g->T[i + 1] = (g->T[i] - g->k[i]) / g->s[i] + 1;
    g->F[i + 1] = (g->F[i] - g->k[i]) / g->s[i] + 1;
    if (g->T[i] < g->k[i] || g->F[i] < g->k[i]) return false;
  }
  g->T_out = g->T[convs];
  g->F_out = g->F[convs];
  return g->T_out >= 1 && g->F_out >= 1;
}

// bytes of the channel-last maps c1 (after conv 0), c2, c3 (0 when absent)
inline void map_bytes(const Geo& g, int B, int d, size_t es, size_t out[3]) {
  for (int i = 0; i < 3; ++i) out[i] = i <= g.n ? (size_t)B * g.T[i + 1] * g.F[i + 1] * d * es : 0;
}

// feats -> x [B*T_out][d] f32 = (Linear(conv stack) + b) * sqrt(d).  W: EmConformerWeights-like.
// conv1_wf / conv2_wf: the operands of the fused conv1 + conv2 kernel (csrc/subsample2.hip), or NULL
template <typename W>
inline int run(int dtype, const W* w, const Geo& g, const float* feats, const float* mvn_partial,
               const int32_t* flens, int B, void* c1, void* c2, void* c3, float* x, void* stream,
               const void* conv1_wf = nullptr, const void* conv2_wf = nullptr) {
  const int d = w->d;
  int rc = EM_ERR_UNSUPPORTED;
  const bool no_fused = em_sw().no_sub12;  // developer A/B switch
  const bool try_fused = dtype == EM_BF16 && conv1_wf && conv2_wf && mode_of(w->subsample) == 4 && !no_fused;
  if (try_fused)
    rc = em_conv2d_sub12_bf16(feats, mvn_partial, flens, B, g.T[0], g.F[0], conv1_wf, conv2_wf, w->conv2_b, d, c2, stream);
  if (rc != EM_OK && rc != EM_ERR_UNSUPPORTED) return rc;
  const bool fused = rc == EM_OK;
  if (!fused) {
    rc = em_conv2d_sub1(dtype, feats, mvn_partial, flens, B, g.T[0], g.F[0], w->conv1_w, w->conv1_b, d, c1, stream);
    if (rc != EM_OK) return rc;
  }
  const void* in = fused ? c2 : c1;
  void* outs[2] = {c2, c3};
  const void* ws[2] = {w->conv2_w, w->conv3_w};
  const float* bs[2] = {w->conv2_b, w->conv3_b};
  for (int i = fused ? 1 : 0; i < g.n; ++i) {
    EmGemmArgs a = {};
    a.A = in; a.W = ws[i]; a.C = outs[i]; a.bias = bs[i];
    a.M = B * g.T[i + 2] * g.F[i + 2]; a.N = d; a.K = g.k[i + 1] * g.k[i + 1] * d; a.lda = 0; a.ldc = d; a.scale = 1.f;
    a.T1 = g.T[i + 1]; a.F1 = g.F[i + 1]; a.T2 = g.T[i + 2]; a.F2 = g.F[i + 2]; a.d = d;
    a.conv_k = g.k[i + 1]; a.conv_s = g.s[i + 1];
    if (!a.W || !a.C) return EM_ERR_BAD_ARG;
    rc = em_gemm(dtype, EM_EPI_RELU, EM_A_CONV2, &a, stream);
    if (rc != EM_OK) return rc;
    in = outs[i];
  }
  EmGemmArgs e = {};
  e.A = in; e.W = w->embed_w; e.C = x; e.bias = w->embed_b;
  e.M = B * g.T_out; e.N = d; e.K = g.F_out * d; e.lda = g.F_out * d; e.ldc = d; e.scale = sqrtf((float)d);
  return em_gemm(dtype, EM_EPI_SCALE_F32, EM_A_PLAIN, &e, stream);
}

}  // namespace em_sub
