/* espnet_amd.h — C ABI of libespnet_amd.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for ESPnet's ASR-inference hot path.  The reference has no FFI
 * for this path — its boundary is the Python plugin API — so each entry point below names the
 * reference Python interface it replaces (paths relative to espnet/espnet).  The Python host layer
 * in espnet_amd/ mirrors those interfaces and binds this ABI with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns every buffer (PyTorch allocations, `tensor.data_ptr()`); the library
 *     allocates nothing and keeps no mutable global state;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default);
 *   - return value: EM_OK or a negative EM_ERR_* code; nothing throws across the ABI;
 *   - dtype: EM_F32 (exact-f32 MFMA, parity mode) or EM_BF16 (bf16 MFMA, f32 accumulate).
 *     "act" buffers hold elements of that dtype; residual stream / features / logits are f32.
 */
#ifndef ESPNET_AMD_H_
#define ESPNET_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EM_OK 0
#define EM_ERR_UNSUPPORTED (-1) /* shape/config outside the fast path */
#define EM_ERR_BAD_ARG (-2)
#define EM_ERR_TOO_SHORT (-3) /* < 7 feature frames: TooShortUttError, subsampling.py:31-49 */
#define EM_ERR_LAUNCH (-4)    /* hipGetLastError() != hipSuccess after a launch */
#define EM_ERR_WORKSPACE (-5) /* workspace too small */
#define EM_ERR_IO (-6)        /* host file I/O failed (em_wav_*) */

#define EM_F32 0
#define EM_BF16 1

/* GEMM epilogues (C = A[M,K] * W[N,K]^T, f32 accumulate) */
#define EM_EPI_STORE 0     /* C[act]  = acc + bias                                  */
#define EM_EPI_SWISH 1     /* C[act]  = swish(acc + bias)       (FFN w_1, swish.py) */
#define EM_EPI_RELU 2      /* C[act]  = relu(acc + bias)        (conv2d / decoder FFN) */
#define EM_EPI_RESID_F32 3 /* C[f32] += scale * (acc + bias)    (residual stream)   */
#define EM_EPI_SCALE_F32 4 /* C[f32]  = scale * (acc + bias)    (embed out * sqrt(d)) */
#define EM_EPI_GLU 5       /* C[act][n/2] = (acc_v+b_v) * sigmoid(acc_g+b_g); W rows interleaved
                              in 16-row granules [v0..15,g0..15,v16..31,...] (convolution.py:68-69) */
#define EM_EPI_STORE_F32 6 /* C[f32]  = acc + bias              (CTC / decoder logits) */
/* GEMM + per-row partial arg-max (CTC.argmax without materialising the (M, V) logits, asr/ctc.py:207-215):
 *   C[m][g] = (max, argmax) of (acc + bias) over the 64 columns [64 g, 64 g + 64) as an (f32, i32) pair;
 *   ldc = number of 64-column groups = 2 * ceil(N / 128); finish with em_argmax_partials.        */
#define EM_EPI_ARGMAX_PART 9
#define EM_EPI_GELU 10     /* C[act]  = gelu_erf(acc + bias)    (cgMLP channel_proj1, cgmlp.py:106) */
/* Operands of the LDS-resident attention (em_relpos_attention2_bf16) written by the projection GEMM itself (bf16 only;
 * T1 = T frames per utterance, T2 = Tpad, F1 = heads, d = heads * 64; M resp. N a multiple of T):
 *   EM_EPI_QK_HEADS  A = activations [B*T][K], W = the q | k rows [2d][K], bias [2d]:
 *                    C = q then k, each [B][heads][Tpad][64]   (column n = which * d + 64 * head + c, row m = (b, t))
 *   EM_EPI_VT_HEADS  the SWAPPED product A = W_v [d][K], W = activations [B*T][K], bias PER ROW [d]:
 *                    C = V^T [B][heads][64][Tpad]              (row m = channel, column n = (b, t))
 * Frames t >= T of a (b, head) slab are not written.                                                              */
#define EM_EPI_QK_HEADS 11
#define EM_EPI_VT_HEADS 12

/* A-operand addressing */
#define EM_A_PLAIN 0 /* row m at A + m*lda */
#define EM_A_CONV2 1 /* implicit GEMM of Conv2d(d,d,k,stride s) over a channel-last (B,T1,F1,d) map */

typedef struct EmGemmArgs {
  const void* A;     /* act dtype */
  const void* W;     /* act dtype, [N][K] row-major (torch Linear layout) */
  void* C;           /* act dtype or f32 depending on the epilogue */
  const float* bias; /* [N] f32 or NULL */
  int32_t M, N, K;   /* K % 64 == 0 */
  int32_t lda, ldc;  /* in elements */
  float scale;
  int32_t T1, F1, T2, F2, d; /* EM_A_CONV2 only: M = B*T2*F2, K = 9*d */
  int32_t conv_k, conv_s; /* EM_A_CONV2: square kernel width / stride; 0 = 3 / 2.  K = conv_k^2 * d
                             (5, 3: the second conv of Conv2dSubsampling6, subsampling.py:706-711) */
} EmGemmArgs;

/* ---- library ------------------------------------------------------------------------------- */
int em_version(void);
const char* em_error_string(int code);
/* Developer switches (ESPNET_AMD_* / EM_*_STAMPS environment variables, DESIGN.md: A/B timing and diagnostics between DEVICE code
 * paths, never a CPU path) are read from the environment ONCE, at the first launch that consults one, into a struct the
 * launchers share (csrc/switches.h): no getenv() on a call path.  A process that changes such a variable afterwards - the
 * tests that compare two kernels in one process do - calls this to have them read again. */
void em_dev_switches_reload(void);

/* ---- A1+A2: Stft.forward (espnet2/layers/stft.py:48-120) + power (asr/frontend/default.py:110)
 *      + LogMel.forward (espnet2/layers/log_mel.py:57-84), fused.  n_fft is fixed at 512.
 *   wav        [B][N] f32 (zero padded to the batch maximum N, as espnet_model.py:454)
 *   window     [512] f32: hann(win_length) zero padded/centred to n_fft (torch.stft semantics)
 *   mel_packed [mel_maxlen][n_mels] f32: mel_packed[s][m] = melmat[mel_lo[m]+s][m] (0 past the band)
 *   mel_lo     [n_mels] i32 first non-zero rFFT bin of each filter
 *   flens      [B] i32 valid frames per utterance (stft.py:108-115); frames >= flens are zeroed
 *   wlens      [B] i32 valid samples per utterance, or NULL.  NULL: the reflect padding of
 *              center=True sits at the padded length N, which is what torch.stft does for a padded
 *              batch.  Given: it sits at each utterance's own end, i.e. what the utterance gets when
 *              Speech2Text decodes it alone (asr_inference.py:514, one utterance per call); every
 *              wlens[b] must exceed n_fft/2 like torch.stft's own reflect-pad check
 *   feats      [B][T_f][n_mels] f32 out, T_f = 1 + N/hop                                         */
int em_frontend_logmel_f32(const float* wav, int32_t B, int32_t N, int32_t hop, const float* window,
                           const float* mel_packed, const int32_t* mel_lo, int32_t mel_maxlen,
                           int32_t n_mels, const int32_t* flens, const int32_t* wlens, int32_t T_f,
                           float* feats, void* stream);

/* ---- A3: utterance_mvn (espnet2/layers/utterance_mvn.py:45-88), first half: per-utterance
 *      partial column sums over the valid frames.  partial [B][8][n_mels] f32 out.  The mean
 *      subtraction itself is fused into em_conv2d_sub1 (padded frames become -mean, :73).       */
int em_utt_mvn_partial_f32(const float* feats, const int32_t* flens, int32_t B, int32_t T_f,
                           int32_t n_mels, float* partial, void* stream);

/*   utterance_mvn applied stand-alone (streaming frontend): feats <- (valid ? feats : 0) - mean,
 *   in place; partial from em_utt_mvn_partial_f32.                                              */
int em_utt_mvn_apply_f32(float* feats, const float* partial, const int32_t* flens, int32_t B,
                         int32_t T_f, int32_t n_mels, void* stream);

/* ---- A3 (alternative): GlobalMVN.forward (espnet2/layers/global_mvn.py:71-100), in place:
 *      feats <- (feats - mean) / stdv on frames < flens[b], 0 on padded frames.  mean / stdv [n_mels]
 *      f32 or NULL (norm_means / norm_vars off); flens NULL = all frames valid.                   */
int em_global_mvn_f32(float* feats, const int32_t* flens, const float* mean, const float* stdv,
                      int32_t B, int32_t T_f, int32_t n_mels, void* stream);

/* ---- A4 (first conv): Conv2dSubsampling.conv[0..1] = Conv2d(1,d,3,2)+ReLU
 *      (transformer/subsampling.py:400-403), with the MVN mean subtraction fused on the input.
 *   partial may be NULL (no mean subtraction).  w1 [d][9] f32, b1 [d] f32.
 *   out [B][T1][F1][d] act dtype, channel-last.                                                 */
/* ---- A4 (both convs in one kernel, bf16, d = 256): Conv2dSubsampling.conv = Conv2d(1,d,3,2)+ReLU+Conv2d(d,d,3,2)+ReLU
 *      (transformer/subsampling.py:400-403, 432-447), MVN mean subtraction fused on the input; the [B][T1][F1][d]
 *      map between the two never exists (csrc/subsample2.hip).  Operands packed by the host as MFMA fragments
 *      (lane l = 16 lg + lr of a fragment holds 8 bf16 at bytes [16 l, 16 l + 16)):
 *   conv1_wf [8 chunks][2][64][8] bf16: fragment (cc, f), lane (lr, lg), e: channel c = 32 cc + 16 f + lr, k = 8 lg + e:
 *            k 0-8 hi(w1[c][k]) | 9-17 hi(w1[c][k-9]) | 18-26 lo(w1[c][k-18]) | 27 hi(b1[c]) | 28 lo(b1[c]) | 29-31 0
 *            with hi(x) = bf16(x), lo(x) = bf16(x - hi(x)): conv1 runs on the matrix cores at f32-class accuracy
 *   conv2_wf [8 chunks][9 taps][4 waves][4][64][8] bf16: W2[n = 64 w + 16 (lr / 4) + 4 j + lr % 4][(kt*3+kf)*d + 32 cc + 8 lg + e]
 *   c2       [B][T2][F2][d] bf16 out (channel-last), T2 = ((T_f-1)/2-1)/2, F2 = ((n_mels-1)/2-1)/2
 *   EM_ERR_UNSUPPORTED for d != 256 or n_mels > 82 (the caller then uses em_conv2d_sub1 + em_gemm EM_A_CONV2).   */
int em_conv2d_sub12_bf16(const float* feats, const float* mvn_partial, const int32_t* flens, int32_t B, int32_t T_f,
                         int32_t n_mels, const void* conv1_wf, const void* conv2_wf, const float* conv2_b, int32_t d,
                         void* c2, void* stream);

int em_conv2d_sub1(int dtype, const float* feats, const float* partial, const int32_t* flens,
                   int32_t B, int32_t T_f, int32_t n_mels, const float* w1, const float* b1,
                   int32_t d, void* out, void* stream);

/* ---- dense contraction (A4 conv2 / embed out, A7 projections, A8 pointwise convs, A9 FFN,
 *      A11 ctc_lo, A14 decoder linears): torch.nn.Linear / Conv1d(k=1) / Conv2d semantics.     */
int em_gemm(int dtype, int epilogue, int a_mode, const EmGemmArgs* args, void* stream);

/* ---- A10: LayerNorm (transformer/layer_norm.py:12-42, eps 1e-12) over the f32 residual stream.
 *   em_layernorm : out[act] = LN(x; g, b)               (out_f32 optional f32 copy, may be NULL)
 *   em_layernorm2: x <- LN(x; g1, b1) (in place, f32), out[act] = LN(x; g2, b2)
 *                  (EncoderLayer.norm_final followed by the next block's first norm, or by
 *                  after_norm; encoder_layer.py:170-171, conformer_encoder.py:423-424);
 *                  out_f32 (optional) receives the second LN in f32                              */
int em_layernorm(int dtype, const float* x, const float* g, const float* b, int32_t M, int32_t d,
                 float eps, void* out, float* out_f32, void* stream);
int em_layernorm2(int dtype, float* x, const float* g1, const float* b1, const float* g2,
                  const float* b2, int32_t M, int32_t d, float eps, void* out, float* out_f32,
                  void* stream);

/*   x <- LayerNorm(x; g, b) in place (EncoderLayer.norm_final, encoder_layer.py:170-171)         */
int em_layernorm_inplace_f32(float* x, const float* g, const float* b, int32_t M, int32_t d,
                             float eps, void* stream);

/*   LayerNorm fused into the projection that consumes it, for small-M launches (the decoder / LM step
 *   of the beam search: every pre-norm LayerNorm of decoder_layer.py:96-152 and
 *   transformer_decoder.py:226-227 rides in the prologue of the next Linear):
 *     C[M][N] = epilogue( LN(x[M][K]; ln_g, ln_b, eps) W[N][K]^T + bias ),  x f32, W act,
 *   epilogue in {EM_EPI_STORE, EM_EPI_RELU, EM_EPI_STORE_F32}; K % 64 == 0, K <= 1024.               */
int em_ln_gemm(int dtype, int epilogue, const float* x, const float* ln_g, const float* ln_b, float eps,
               const void* W, const float* bias, void* C, int32_t M, int32_t N, int32_t K, int32_t ldc,
               void* stream);

/* ---- A7: RelPositionMultiHeadedAttention core (transformer/attention.py:416-459 after the
 *      projections): AC = (q+u)k^T, BD[i][j] = (q+v).p[T-1-i+j] (rel_shift :391-408 as index
 *      arithmetic), softmax((AC+BD)/sqrt(dk)) over keys j < klens[b] (masked probs = 0), times V.
 *   qkv  [B*T][3d] act (q | k | v per row);  p [2T-1][ldp] act (ldp = row stride, elements)
 *   pos_u,pos_v [h][dk] f32;  ctx [B*T][d] act out.  dk must be 64.                              */
int em_relpos_attention(int dtype, const void* qkv, const void* p, int32_t ldp, const float* pos_u,
                        const float* pos_v, const int32_t* klens, int32_t B, int32_t T, int32_t h,
                        int32_t dk, void* ctx, void* stream);

/*   LegacyRelPositionMultiHeadedAttention (attention.py:268-360; `rel_pos_type: legacy`, the class default
 *   that older checkpoints were trained with).  p [T][ldp] act = linear_pos of LegacyRelPositionalEncoding
 *   rows (embedding.py:223-262: row k = sinusoid of position max_len-1-k); the square pad-and-reshape
 *   rel_shift (:296-316) gives BD[i][j] = (q_i+v).p[T-1-i+j] for j <= i, 0 for j == i+1 and
 *   (q_{i+1}+v).p[j-i-2] above (the wrapped rows), reproduced as index arithmetic.  zero_triu unsupported. */
int em_legacy_relpos_attention(int dtype, const void* qkv, const void* p, int32_t ldp, const float* pos_u,
                               const float* pos_v, const int32_t* klens, int32_t B, int32_t T, int32_t h,
                               int32_t dk, void* ctx, void* stream);

/* ---- A8 (middle): depthwise Conv1d(k, pad (k-1)/2) + eval BatchNorm1d (folded into w,b by the
 *      caller) + Swish (conformer/convolution.py:72-75).  x,y [B][T][d] act; w [k][d] f32 (tap-major).
 *      tlens [B] i32 or NULL: with lengths, input frames t >= tlens[b] read as zero (the utterance
 *      decoded alone); NULL reproduces the reference's unmasked conv over a padded batch.          */
int em_dwconv_bn_swish(int dtype, const void* x, const float* w, const float* b,
                       const int32_t* tlens, int32_t B, int32_t T, int32_t d, int32_t k, void* y,
                       void* stream);

/*   The same kernel with the output stage selectable and per-operand row strides (E-Branchformer:
 *   espnet2/asr/layers/cgmlp.py:61-79 and e_branchformer_encoder.py:166-170):
 *     EM_DW_SWISH   y = swish(conv(x) + b)      EM_DW_LINEAR  y = conv(x) + b
 *     EM_DW_GATE    y = gate * (conv(x) + b)    EM_DW_SELFRES y = x + conv(x) + b
 *   x [B][T] rows of stride ldx, gate rows of stride ldg (EM_DW_GATE only), y rows of stride ldy.   */
#define EM_DW_SWISH 0
#define EM_DW_LINEAR 1
#define EM_DW_GATE 2
#define EM_DW_SELFRES 3
int em_dwconv(int dtype, int mode, const void* x, int32_t ldx, const float* w, const float* b,
              const int32_t* tlens, int32_t B, int32_t T, int32_t d, int32_t k, const void* gate,
              int32_t ldg, void* y, int32_t ldy, void* stream);
/*   LayerNorm (layer_norm.py:12-42) of an act-dtype matrix with row strides: the gate half of the cgMLP
 *   hidden activation (cgmlp.py:63).  x rows of stride ldx, n channels, out rows of stride ldo.        */
int em_layernorm_act(int dtype, const void* x, int32_t ldx, const float* g, const float* b, int32_t M,
                     int32_t n, float eps, void* out, int32_t ldo, void* stream);

/*   The cgMLP gate path with the LayerNorm folded into the conv's input stage (cgmlp.py:61-79):
 *   stats [B*T][2] = (mean, rstd) of every input row from em_row_stats; y = gate * (conv(LN(x)) + b).
 *   Bit-identical to em_layernorm_act followed by em_dwconv(EM_DW_GATE); saves one write + read of x. */
int em_row_stats(int dtype, const void* x, int32_t ldx, int32_t M, int32_t n, float eps, float* stats,
                 void* stream);
int em_dwconv_ln_gate(int dtype, const void* x, int32_t ldx, const float* stats, const float* ln_g,
                      const float* ln_b, const float* w, const float* b, const int32_t* tlens, int32_t B,
                      int32_t T, int32_t d, int32_t k, const void* gate, int32_t ldg, void* y, int32_t ldy,
                      void* stream);

/* ---- A11 / G1: CTC head.  argmax over vocab (asr/ctc.py:207-215), then groupby + drop
 *      blank/sos/eos (bin/asr_inference.py:574-575).  logits [M][V] f32.                        */
int em_argmax_rows_f32(const float* logits, int32_t M, int32_t V, int32_t* ids, void* stream);
/*   ids[m] = column of the best of the G (value, column) pairs of row m (ties -> lowest column) */
int em_argmax_partials(const float* part, int32_t M, int32_t G, int32_t* ids, void* stream);
int em_log_softmax_rows_f32(float* logits, int32_t M, int32_t V, void* stream);
int em_ctc_collapse(const int32_t* ids, const int32_t* olens, int32_t B, int32_t T, int32_t blank,
                    int32_t sos_eos, int32_t* tokens, int32_t* out_lens, void* stream);

/* ---- A4..A10 assembled: ConformerEncoder.forward (espnet2/asr/encoder/conformer_encoder.py:327-429)
 *      for input_layer=conv2d, rel_pos / rel_selfattn (latest), macaron, cnn module,
 *      normalize_before; EncoderLayer.forward (conformer/encoder_layer.py:79-179) per block.
 *      Weights are packed once at load time by the host layer (espnet_amd/asr/encoder/
 *      conformer_encoder.py) into caller-owned device buffers; this struct only carries pointers
 *      ("act" = dtype of the call, everything else f32).  The struct itself lives in HOST memory. */
typedef struct EmConformerLayer {
  const float *norm_ff_mac_g, *norm_ff_mac_b, *norm_mha_g, *norm_mha_b, *norm_conv_g, *norm_conv_b,
      *norm_ff_g, *norm_ff_b, *norm_final_g, *norm_final_b;
  const void* ffm_w1; /* [ff][d] act */
  const float* ffm_b1;
  const void* ffm_w2; /* [d][ff] act */
  const float* ffm_b2;
  const void* wqkv; /* [3d][d] act: linear_q | linear_k | linear_v rows */
  const float* bqkv;
  const float *pos_u, *pos_v; /* [h][dk] */
  const void* wout;           /* [d][d] act */
  const float* bout;
  const void* pw1; /* [2d][d] act, GLU-interleaved rows (EM_EPI_GLU) */
  const float* pw1_b;
  const float *dw_w, *dw_b; /* [k][d] (tap-major), [d] with eval BatchNorm folded in */
  const void* pw2;          /* [d][d] act */
  const float* pw2_b;
  const void* ff_w1;
  const float* ff_b1;
  const void* ff_w2;
  const float* ff_b2;
  /* row-block fused path (em_conformer_block_fused; bf16, d = 256): NULL when the host did not pack it */
  /* every matrix below is bf16 in FRAGMENT-MAJOR 32 KiB units (see EmBlockArgs): a wave's MFMA weight operand is
   * one contiguous 1 KiB line, lane l reading bytes [16 l, 16 l + 16) */
  const void* pw1f;      /* [2d][d]: pointwise_conv1 rows in 64-row granules [v0..63, g0..63, v64..127, ...], K units */
  const void *ffm_w2p, *ff_w2p; /* w_2 of the two FFNs, packed per pair of hidden chunks (EmBlockArgs) */
  const void *woutp, *pw2p, *ff_w1p, *ffm_w1p, *wqkvp; /* K units of wout, pw2, ff_w1, ffm_w1, wqkv */
  /* d = 512 (bf16): wqkvp = [linear_q | linear_k | linear_v] in the w1p layout (round 6: walked behind the macaron launch,
   * EmFfnRowsArgs.post_q; NULL: two projection GEMMs); ffm_w1p / ffm_w2p / ff_w1p / ff_w2p hold the operand streams of em_ffn_rows_fused instead (EmFfnRowsArgs
   * w1p / w2p), pw2p the projection in front of the second one (EmFfnRowsArgs.pre_w), woutp / pw1f / fp_c the operands of the
   * EM_ROWS_GLU launch (linear_out as pre_w, pointwise_conv1 in value / gate chunk pairs, its bias in that order), and
   * em_conformer_encode runs the block in 10 launches (when its rows fill the chip); the other fields stay NULL */
  const float* fp_c;     /* parameter groups of block<C> for this layer (EM_BLOCK_PARAM_GROUP floats each) */
  const float* fp_da;    /* groups of block<D|A> (D part of this layer, A part of the next) or block<D|FINAL> */
  const float* fp_a;     /* groups of block<A> (layer 0 only) */
} EmConformerLayer;

typedef struct EmConformerWeights {
  int32_t d, heads, ff, num_blocks, kernel, n_mels;
  const float *conv1_w, *conv1_b; /* [d][9], [d] */
  const void* conv2_w;            /* [d][9d] act, k = (kt*3+kf)*d + c_in */
  const float* conv2_b;
  const void* embed_w; /* [d][F2*d] act, columns permuted to f*d + c (channel-last) */
  const float* embed_b;
  const void* wpos_all; /* [num_blocks*d][d] act: linear_pos of every block stacked */
  const float *after_norm_g, *after_norm_b;
  const EmConformerLayer* layers; /* [num_blocks], host array */
  /* input layer (transformer/subsampling.py): 0 or 4 = Conv2dSubsampling (:386), 6 = Conv2dSubsampling6
   * (:692: second conv 5x5 stride 3, conv2_w [d][25d]), 8 = Conv2dSubsampling8 (:785: a third 3x3
   * stride-2 conv, conv3_w [d][9d] act, conv3_b).  embed_w is [d][F_out*d] with F_out of that stack. */
  int32_t subsample;
  const void* conv3_w;
  const float* conv3_b;
  int32_t legacy_relpos; /* 1: rel_pos_type legacy (pos_emb has T rows, em_legacy_relpos_attention) */
  /* optional CTC head fused into the last block kernel (EmBlockArgs.ctc_*; bf16 fused path only): with ctc_ids
   * non-NULL em_conformer_encode also writes the per-frame CTC arg-max ids [B][T] i32 there.               */
  const void* ctc_w;
  const float* ctc_b;
  int32_t* ctc_ids;
  int32_t ctc_units;
  /* optional operands of the fused conv1 + conv2 kernel (em_conv2d_sub12_bf16; bf16, d = 256, input layer conv2d):
   * with both non-NULL em_conformer_encode never materialises the conv1 map.  NULL: conv1 and conv2 as two launches. */
  const void* conv1_wf;
  const void* conv2_wf;
} EmConformerWeights;

/* em_conformer_encode flags */
#define EM_ENC_ISOLATE_UTTS 1 /* every utterance of a ragged batch encodes as if it were alone */
#define EM_ENC_NO_FUSED 2     /* keep the one-operator-per-launch sequence even where the fused block kernels apply */
#define EM_ENC_FOLD_C 8       /* fused path: the C part of a block inside the launch that consumes it (block<C|D|...>: two launches
                               * per Conformer block instead of three; measured no faster at B = 32, DESIGN.md: opt-in) */
#define EM_ENC_SPLIT_ATT 16    /* fused path: attention and the C part of a block as TWO launches (em_relpos_attention2_bf16 + block<C>,
                               * rounds 2-5) instead of the one launch block<ATT|C> of round 6: developer A/B switch */
#define EM_ENC_POS_PACKED 32   /* with EM_ENC_POS_PROJECTED (bf16, d = 256): behind the projected table, at the next multiple of 256
                               * bytes, pos_emb ALSO holds em_relpos_pack_pos_bf16's output for (T, the num_blocks blocks) - what
                               * block<ATT|C> reads; without the flag the encoder packs it per call (~5 us at T = 249) */
#define EM_ENC_POS_PROJECTED 4 /* pos_emb is ALREADY linear_pos of every block: [2T-1 (legacy: T)][L*d] act, i.e. pos_emb x wpos_all^T
                               * (it depends on T and the weights only: a caller decoding many batches of one length projects once) */
#define EM_ENC_IN_FLIGHT(n) (((n) & 15) << 8) /* the caller keeps n batches in flight on n HIP streams (0 / 1: this call has the
                               * chip to itself).  The 512-wide models' row-block launches (64-row workgroups, csrc/ffn_rows.hip) are
                               * taken when a round of them fills 72 % / n of the CUs: a launch that fills half the chip loses to the
                               * per-operator sequence alone and wins when another stream's launch runs beside it (Conformer-large,
                               * B = 32: 3.72 vs 3.84 ms alone, 3.28 vs 2.63 ms with two in flight; profiles/r06ae_rows_fill_ab.txt) */

/* Which launch sequence em_conformer_encode will take for these weights and flags (>= 0; negative = status):
 * the single source of that decision, so that the host layer never re-derives the shape conditions.           */
#define EM_ENC_PLAN_FUSED 1   /* fused per-block kernels (csrc/block.hip) */
#define EM_ENC_PLAN_CTC_IDS 2 /* ... and w->ctc_ids [B][T] WILL be written by the last block kernel */
int em_conformer_encode_plan(int dtype, const EmConformerWeights* w, int32_t flags);
/*   ... for a given batch (B utterances of T_f feature frames): the 512-wide model's row-block launches - and with them the
 *   CTC arg-max behind the last one (ctc_w then = ctc_lo.weight zero-padded to ctc_units * 128 rows in the EmFfnRowsArgs.w1p
 *   layout, ctc_b padded with -3e38) - are taken only when the batch's rows fill the chip, so EM_ENC_PLAN_CTC_IDS depends on
 *   the shape there.  Equal to em_conformer_encode_plan for every other model.                                          */
int em_conformer_encode_plan_for(int dtype, const EmConformerWeights* w, int32_t flags, int32_t B, int32_t T_f);

/* bytes of scratch em_conformer_encode needs for (B, T_f) */
size_t em_conformer_workspace_bytes(int dtype, const EmConformerWeights* w, int32_t B, int32_t T_f);

/*   feats      [B][T_f][n_mels] f32 (log-mel, not yet mean-normalised)
 *   mvn_partial[B][8][n_mels] f32 from em_utt_mvn_partial_f32, or NULL for no mean subtraction
 *   flens      [B] i32 valid feature frames;  olens [B] i32 valid encoder frames (host computes
 *              both from the lengths exactly as the reference's masks do)
 *   pos_emb    [2T-1][d] act: RelPositionalEncoding table rows (embedding.py:286-332); with
 *              w->legacy_relpos [T][d]: LegacyRelPositionalEncoding rows (embedding.py:223-262)
 *   enc_out    [B][T][d] f32 out (after_norm applied); enc_act same in act dtype (input of the
 *              CTC / decoder projections), T = ((T_f-1)/2-1)/2
 *   flags      0: padded-batch semantics of ConformerEncoder.forward (padded frames leak through the
 *              unmasked depthwise conv, convolution.py:72);  EM_ENC_ISOLATE_UTTS: batching is
 *              transparent, row b equals the encoding of utterance b alone                        */
int em_conformer_encode(int dtype, const EmConformerWeights* w, const float* feats,
                        const float* mvn_partial, const int32_t* flens, const int32_t* olens,
                        int32_t B, int32_t T_f, const void* pos_emb, void* workspace,
                        size_t workspace_bytes, float* enc_out, void* enc_act, int32_t flags,
                        void* stream);

/* ---- A6-A10 fused per Conformer block (bf16, d = 256, ff <= 1024, cnn_module_kernel 31): a workgroup
 *      carries 32 frames of one utterance through every row-local operator of EncoderLayer.forward
 *      (conformer/encoder_layer.py:79-179); weights stream from global memory into MFMA registers (csrc/block.hip).
 *   mode  EM_BLOCK_C                : ctx -> linear_out + residual -> norm_conv -> pointwise_conv1 + GLU -> glu
 *         EM_BLOCK_D | EM_BLOCK_A   : glu -> depthwise conv + BN + Swish -> pointwise_conv2 + residual -> norm_ff
 *                                     -> FFN + residual -> norm_final; next block: norm_ff_macaron -> macaron FFN
 *                                     + residual -> norm_mha -> q / k / v projections
 *         EM_BLOCK_C | EM_BLOCK_D | EM_BLOCK_A (round 4): both of the above in ONE launch (see x_out / params_c below)
 *         EM_BLOCK_A                : the second half alone (first block, after the embedding)
 *         EM_BLOCK_D | EM_BLOCK_FINAL: the first half, then after_norm -> enc_out (f32) / enc_act (bf16)
 *         EM_BLOCK_ATT | EM_BLOCK_C (round 6): relative-position attention + the C part in one launch (see EM_BLOCK_ATT)
 *   x     [B*T][256] f32 residual stream, updated in place (block<C> and the A part write it back)
 *   q / k [B][H][Tpad][64] bf16, vt [B][H][64][Tpad] bf16 (V transposed): inputs of em_relpos_attention2;
 *         Tpad % 64 == 0, Tpad >= 32 * ceil(T / 32); frames >= T of the last block hold finite padding
 *   params: f32 groups of EM_BLOCK_PARAM_GROUP floats, in the order the kernel consumes them:
 *     C      : [bout 256 | norm_conv g 256 | b 256 | pointwise_conv1 bias in pw1f row order 512]
 *     D part : [pw2 bias 256 | norm_ff g | b] , [ff b1 (1024 slots) | ff b2 256 | norm_final g | b]
 *     A part : [norm_ff_macaron g | b | ffm b1 (1024 slots) | ffm b2] , [norm_mha g | b | bq bk bv 768]
 *     FINAL  : [after_norm g | b]     (D part + A part / FINAL concatenated for the combined modes)   */
#define EM_BLOCK_C 1
#define EM_BLOCK_D 2
#define EM_BLOCK_A 4
#define EM_BLOCK_FINAL 8
#define EM_BLOCK_CTC 16 /* with D | FINAL: + the CTC head's per-frame arg-max */
#define EM_BLOCK_RELU 32 /* the contextual-block STREAMING layer (contextual_block_encoder_layer.py:197-310): ReLU feed-forward,
                          * depthwise conv width 15, ff up to 4096 (ffm_b1g / ff_b1g).  With EM_BLOCK_A (macaron FFN, norm_mha,
                          * q / k / v per head) or EM_BLOCK_D alone (conv module back, FFN, norm_final -> x: the context
                          * hand-over between layers sits behind it); EM_BLOCK_C is the same kernel as without. */
#define EM_BLOCK_ATT 64  /* round 6, with EM_BLOCK_C only: the relative-position self-attention of the workgroup's 32 queries
                          * (RelPositionMultiHeadedAttention.forward, attention.py:391-459: all four heads, a wave per head)
                          * runs IN FRONT of the C part, inside the same launch: one launch per (utterance, 32 queries) replaces
                          * em_relpos_attention2_bf16 + block<C>, the context `ctx` never exists in memory.  Inputs qh / kh / vt
                          * as A wrote them, plus pos / ldp / pos_u / pos_v / klens below (em_relpos_attention2_bf16's
                          * arguments of the same names).  Any T; d = 256, 4 heads of 64. */
#define EM_BLOCK_PARAM_GROUP 1792
typedef struct EmBlockArgs {
  int32_t B, T, Tpad, d, ff, kernel;
  float eps;                        /* LayerNorm eps (1e-12, layer_norm.py:23) */
  float* x;
  const void* ctx;                  /* C in:  [B*T][256] bf16 */
  void* glu;                        /* C out / D in: [B*T][256] bf16 */
  void *qh, *kh, *vt;               /* A out */
  float* enc_out;                   /* FINAL out [B*T][256] f32 */
  void* enc_act;                    /* FINAL out bf16 */
  const int32_t* tlens;             /* D: valid frames per utterance for the depthwise conv, or NULL (= T) */
  /* Weight matrices (bf16) are streamed in 32 KiB units packed FRAGMENT-MAJOR, so that the eight MFMA operands a
   * wave needs from a unit are eight contiguous 1 KiB lines (lane l = 16 lg + lr reads bytes [16 l, 16 l + 16)):
   *   K unit  u of a [N][256] matrix W (rows 64 u .. 64 u + 63):   unit[nf][ks][lg][lr][e] = W[64 u + 16 nf + lr][32 ks + 8 lg + e]
   * with nf < 4, ks < 8, lg < 4, lr < 16, e < 8.  wout, pw1f, pw2, ff_w1, ffm_w1, wqkv and ctc_w are K units.
   * The second matrix of an FFN (ff_w2 / ffm_w2, [256][ff]) is packed per PAIR of 64-wide hidden chunks, 64 KiB per
   * pair p, in the order in which the wave that computed a slice of the hidden activation holds it in registers
   * (csrc/block.hip, ffn()): wave w contracts over its own 16 hidden columns of both chunks,
   *   pair[p][w][f][lg][lr][e] = W[16 f + lr][64 (2 p + (e >> 2)) + 16 w + 4 lg + (e & 3)],   w < 4, f < 16,
   * i.e. for every wave 16 output-fragment lines of 1 KiB.  ff that is an odd multiple of 64 is padded with one zero
   * chunk (zero rows of w_1, zero bias, zero columns of w_2) by the host packers
   * (espnet_amd.asr.encoder.conformer_encoder.pack_k_units / pack_w1 / pack_w2); the kernel takes the true ff. */
  const void *wout, *pw1f;          /* C */
  const void *pw2, *ff_w1, *ff_w2;  /* D */
  const float *dw_w, *dw_b;         /* D: [31][256] tap-major, [256] (BatchNorm folded) */
  const void *ffm_w1, *ffm_w2, *wqkv; /* A */
  const float* params;
  /* EM_BLOCK_CTC (with D | FINAL): the CTC head's arg-max (CTC.argmax, espnet2/asr/ctc.py:207-215) straight from
   * the after_norm rows in registers: ctc_w [64 * ctc_units][256] act = ctc_lo.weight zero-padded to a multiple
   * of 64 rows, ctc_b [64 * ctc_units] f32 = ctc_lo.bias padded with -3e38, ctc_ids [B][T] i32 out (lowest
   * index on ties, as torch.argmax).  The logits never exist in memory.                                    */
  const void* ctc_w;
  const float* ctc_b;
  int32_t* ctc_ids;
  int32_t ctc_units;
  /* EM_BLOCK_C | EM_BLOCK_D | ...: the C part FOLDED into the launch that consumes it (round 4).  The workgroup computes
   * linear_out / norm_conv / pointwise_conv1 + GLU for its 32 frames AND the 16 either side (the depthwise conv's halo)
   * and keeps the GLU rows in LDS: `glu` is not used, `ctx`, `wout`, `pw1f` are, `params_c` is the C part's parameter
   * group, and the residual stream is read from `x` and written to `x_out` (a different buffer: another workgroup of the
   * same launch reads this one's rows of `x` as its halo).  With EM_BLOCK_FINAL nothing is written to x_out.        */
  float* x_out;
  const float* params_c;
  /* ff > 1024: the first bias of the FFNs in global memory, [ff rounded up to 128] f32 (the 7 KiB parameter groups hold
   * 1024 of it; their b1 slot is then unused).  NULL otherwise. */
  const float* ffm_b1g;
  const float* ff_b1g;
  /* Contextual-block streaming layers with ONE block per stream and call (round 5): the context hand-over between layers
   * (contextual_block_encoder_layer.py:292-304) without a launch of its own.  EM_BLOCK_A alone: slot 0 of block b is
   * read from row0_src + b * row_stride instead of x (the previous call's context vector of the layer in front);
   * EM_BLOCK_D alone: slot T - 1 of block b is ALSO written to last_dst + b * row_stride (this call's context vector of
   * the layer).  [256] f32 rows, row_stride in floats.  NULL: off. */
  const float* row0_src;
  float* last_dst;
  int32_t row_stride;
  /* EM_BLOCK_ATT | EM_BLOCK_C (round 6): the attention's operands besides qh / kh / vt (inputs here, as the A part of
   * the launch in front wrote them; rows >= T of qh / kh may hold anything, columns >= T of vt must be finite):
   * pos / ldp: the position rows p[T - 1 - i + j] fragment-major (see kv_frag below), pos_u / pos_v [4][64] f32, klens [B]
   * valid keys per utterance (clamped to T). */
  int32_t ldp;
  const void* pos;
  const float *pos_u, *pos_v;
  const int32_t* klens;
  /* kv_frag != 0: K and V^T FRAGMENT-MAJOR per 64-key tile - what EM_BLOCK_A writes and what EM_BLOCK_ATT reads (0: the
   * row-major layouts above, em_relpos_attention2_bf16's).  Same buffer sizes; per (utterance, head) and tile jt, 8 KiB:
   *   K   tile[n][ks][lane = 16 lg + lr][e] = k[key 64 jt + 32 (n >> 1) + 8 (lr >> 2) + 4 (n & 1) + (lr & 3)][32 ks + 8 lg + e]
   *   V^T tile[f][jp][lane][e]              = v[key 64 jt + 32 jp + 8 lg + e][16 f + lr]        n, f < 4; ks, jp < 2; e < 8
   * (a lane's eight contraction slots of P . V are eight consecutive keys; every MFMA operand is one contiguous KiB).
   * With EM_BLOCK_ATT, `pos` is em_relpos_pack_pos_bf16's table of THIS block ([4 heads][ldp fragments][2][64][8] bf16)
   * and ldp = em_relpos_pos_fragments(T). */
  int32_t kv_frag;
  /* EM_BLOCK_RELU (streaming layers), ffn_split = S > 1 (round 6): the FFN's hidden dimension is dealt to S workgroups per
   * 32-row block; each leaves its partial sum in ffn_part and takes a ticket from ffn_ticket, the last to arrive adds the
   * partial sums in split order and carries the rows through the rest of the launch (nobody waits).  ff / 64 rounded up to
   * an even number must be a multiple of 2 S.  ffn_part: [B * ceil(T / 32)][S][8192] f32 scratch; ffn_ticket: [B * ceil(T / 32)]
   * i32, ZERO before the first launch (every launch leaves it zero).  S <= 1: off. */
  int32_t ffn_split;
  float* ffn_part;
  int32_t* ffn_ticket;
  /* EM_BLOCK_ATT | EM_BLOCK_C | EM_BLOCK_RELU (streaming layers, round 6): PLAIN multi-head attention over the block's T <= 64
   * slots in front of the C part (q / k [B][4][Tpad][64], V^T [B][4][64][Tpad] row-major as EM_BLOCK_A | EM_BLOCK_RELU
   * writes them; no pos / klens) - bit for bit em_cb_encode_blocks' attention launch + EM_BLOCK_C.  att_mask != 0: the
   * contextual mask (contextual_block_conformer_encoder.py:352-360): the last slot is no key; slot 0 attends to nothing and
   * its context is zero. */
  int32_t att_mask;
} EmBlockArgs;
int em_conformer_block_fused(int mode, const EmBlockArgs* args, void* stream);
/* Position rows for EM_BLOCK_ATT: pall [2T-1][ldp] bf16 holds linear_pos of L blocks side by side (block l at columns
 * 256 l ..); out [L][4][em_relpos_pos_fragments(T)][2][64][8] bf16: fragment g of a head = rows 16 g + T - 32 ceil(T/32) - 32 + lr
 * (clamped to the table), lane 16 lg + lr holding columns 32 ks + 8 lg .. + 7 of its row.  Depends on T and the weights only. */
int em_relpos_pos_fragments(int32_t T);
int em_relpos_pack_pos_bf16(const void* pall, int32_t ldp, int32_t T, int32_t L, void* out, void* stream);

/* ---- A9 + A10, row-block fused feed-forward module for the 512-wide Conformer (bf16; csrc/ffn_rows.hip):
 *      PositionwiseFeedForward.forward (transformer/positionwise_feed_forward.py:30-32) with its residual
 *      (conformer/encoder_layer.py:111-121, :160-168) and the LayerNorm(s) that follow it (:124 norm_mha; :170-171
 *      norm_final + the next block's :112 norm_ff_macaron, or ConformerEncoder's after_norm), 64 rows per workgroup:
 *        ln_mode 1:  x += scale * (W2 . swish(W1 . xn_in + b1) + b2);      xn_out = LN(x; g1, be1)
 *        ln_mode 2:  x  = LN(x + scale * (...); g1, be1);                  xn_out = LN(x; g2, be2)  (+ out_f32)
 *      The [M][ff] hidden activation is never written; x is read and written once.  xn_out may alias xn_in.
 *   w1p: W1 [ff][512] in 128-row chunks of 16 x 8 MFMA operand fragments of 1 KiB (the eight waves' fragments adjacent):
 *        [c][ks 16][w 8][lane = 16 lg + lr][e 8] = W1[128 c + 16 w + lr][32 ks + 8 lg + e]
 *   w2p: W2 [512][ff], the contraction index ordered as the waves hold the hidden activation:
 *        [c][s 4][cf 4][w 8][lane][e 8] = W2[64 w + 16 cf + lr][128 c + 32 s + 16 (e >> 2) + 4 lg + (e & 3)]
 *        (host: pack_ffn_rows_w1 / pack_ffn_rows_w2, espnet_amd/asr/encoder/conformer_encoder.py)            */
typedef struct EmFfnRowsArgs {
  const void* xn_in; /* [M][512] bf16: the module's input, LN(x) */
  float* x;          /* [M][512] f32 residual stream, updated in place */
  const void *w1p, *w2p;
  const float *b1, *b2; /* [ff], [512] */
  const float *g1, *be1, *g2, *be2; /* LayerNorm weights / biases [512] (g2 / be2: ln_mode 2 only) */
  void* xn_out;   /* [M][512] bf16 */
  float* out_f32; /* optional (ln_mode 2): the second LayerNorm's result in f32 as well */
  int32_t M, d, ff, ln_mode;
  float scale, eps;
  /* optional projection in front of the module (pre_in != NULL; xn_in is then unused):
   *   x += W_pre . pre_in + pre_b;   the module's input = LN(x; pre_g, pre_be)
   * i.e. ConvolutionModule's pointwise_conv2 (conformer/convolution.py:78-79) with the residual of
   * encoder_layer.py:158 and norm_ff (:161) inside the launch of the second feed-forward module.
   *   pre_in [M][512] bf16;  pre_w: W_pre [512][512] as 64 x 8 operand fragments of 1 KiB,
   *   [ks 16][cf 4][w 8][lane][e 8] = W_pre[64 w + 16 cf + lr][32 ks + 8 lg + e]  (host: pack_rows_proj)          */
  /* round 6: with pre_in, pre_w AND xn_in all NULL but pre_g / pre_be set (ln_mode 2, no walk) there is no projection and the
   * module's input is LayerNorm(x; pre_g, pre_be) of the residual rows, computed in the launch (E-Branchformer's norm_ff in
   * front of its second feed-forward module, e_branchformer_encoder.py:172-174): no LayerNorm launch in front.                 */
  const void* pre_in;
  const void* pre_w;
  const float *pre_b, *pre_g, *pre_be;
  /* what follows the projection: EM_ROWS_FFN (0) the feed-forward module above;  EM_ROWS_GLU (needs pre_in): pointwise_conv1 +
   * GLU of the conv module (conformer/convolution.py:62-66) on LN(x; pre_g, pre_be) - with linear_out as the projection the
   * launch is attention output -> x += linear_out . ctx + b -> norm_conv -> glu (encoder_layer.py:147-151).  Then
   *   w1p: pointwise_conv1 [2 * 512][512] in the w1p layout with chunk 2 j = the value rows of output columns 128 j ..
   *        128 j + 127 and chunk 2 j + 1 their gate rows (host: pack_rows_glu), b1 its bias in that order, ff = 1024;
   *   xn_out [M][512] bf16 = value * sigmoid(gate);  x is updated;  w2p, b2, g1 .. be2, out_f32, ln_mode, scale unused.   */
  int32_t main;
  /* optional walk behind a ln_mode 2 launch (post_w != NULL): the CTC head's per-row arg-max over the second LayerNorm's
   * result (asr/ctc.py:207-215), post_ids [M] i32 out.  post_w: ctc_lo.weight zero-padded to post_chunks * 128 rows, in the
   * w1p layout; post_b its bias padded with -3e38; post_vocab the vocabulary rounded up to whole 128-label chunks, i.e. what the walk computes (flop accounting only).            */
  const void* post_w;
  const float* post_b;
  int32_t* post_ids;
  int32_t post_chunks, post_vocab;
  /* round 6, optional walk behind a ln_mode 1 launch WITHOUT a projection in front (post_q != NULL): the q | k | v projections
   * of the attention that follows norm_mha (transformer/attention.py:77-98), written per head as csrc/attention2.hip reads
   * them - post_q, post_k [B][8][post_Tpad][64] bf16, post_vt [B][512][post_Tpad] bf16 (V transposed; frames >= post_T are the
   * caller's padding) - with M = B * post_T rows.  post_w: [linear_q | linear_k | linear_v] (1 536 x 512) in the w1p layout
   * (post_chunks = 12), post_b their biases [1 536].  LN(x; g1, be1) is not stored; xn_out, when not NULL, receives a SIBLING
   * LayerNorm of the same rows, LN(x; g2, be2) (E-Branchformer's norm_mlp beside norm_mha, e_branchformer_encoder.py:138-139).   */
  void *post_q, *post_k, *post_vt;
  int32_t post_T, post_Tpad;
} EmFfnRowsArgs;
#define EM_ROWS_FFN 0
#define EM_ROWS_GLU 1
int em_ffn_rows_fused(const EmFfnRowsArgs* args, void* stream);

/* ---- A7, LDS-resident form (bf16, d_k = 64): RelPositionMultiHeadedAttention.forward core
 *      (transformer/attention.py:416-459, rel_shift :391-408) over per-head operands.  One workgroup
 *      per (utterance, head, 128 queries); K, V^T and the position window of 256 keys sit in LDS.
 *   qh / kh [B][H][Tpad][64], vt [B][H][64][Tpad] bf16 (Tpad % 256 == 0), p [2T-1][ldp] bf16 (this block's
 *   linear_pos rows), ctx [B*T][H*64] bf16 out.                                                      */
int em_relpos_attention2_bf16(const void* qh, const void* kh, const void* vt, const void* p, int32_t ldp,
                              const float* pos_u, const float* pos_v, const int32_t* klens, int32_t B,
                              int32_t T, int32_t Tpad, int32_t h, void* ctx, void* stream);

/* ---- §8(f) rank 4: E-Branchformer encoder (espnet2/asr/encoder/e_branchformer_encoder.py:55-520) for
 *      input_layer=conv2d, rel_pos / rel_selfattn (latest), use_ffn + macaron_ffn, swish FFN,
 *      cgMLP with identity gate and no linear after the conv (espnet2/asr/layers/cgmlp.py).
 *      Conv2dSubsampling, rel-pos attention, FFN and LayerNorm are the Conformer kernels.          */
typedef struct EmEBranchformerLayer {
  const float *norm_ff_mac_g, *norm_ff_mac_b, *norm_mha_g, *norm_mha_b, *norm_mlp_g, *norm_mlp_b;
  const float *norm_ff_g, *norm_ff_b, *norm_final_g, *norm_final_b;
  const void *ffm_w1, *ffm_w2, *ff_w1, *ff_w2; /* [ff][d], [d][ff] act */
  const float *ffm_b1, *ffm_b2, *ff_b1, *ff_b2;
  const void* wqkv;  /* [3d][d] act (q | k | v) */
  const float* bqkv;
  const float *pos_u, *pos_v; /* [h][dk] */
  const void* wout;           /* [d][d] act */
  const float* bout;
  const void* proj1_w; /* cgmlp.channel_proj1.0 [cg][d] act */
  const float* proj1_b;
  const float *csgu_norm_g, *csgu_norm_b; /* [cg/2] */
  const float *csgu_conv_w, *csgu_conv_b; /* [k][cg/2] tap-major, [cg/2] */
  const void* proj2_w;                    /* cgmlp.channel_proj2 [d][cg/2] act */
  const float* proj2_b;
  const float *merge_conv_w, *merge_conv_b; /* depthwise_conv_fusion [km][2d] tap-major, [2d] */
  const void* merge_w;                      /* merge_proj [d][2d] act ([d][d] for EM_MERGE_LEARNED_AVE) */
  const float* merge_b;
  /* Branchformer merge_method="learned_ave" (branchformer_encoder.py:102-112); NULL otherwise */
  const float *pool_w, *pool_b;   /* pooling_proj1 | pooling_proj2: [2][d], [2] */
  const float *wproj_w, *wproj_b; /* weight_proj1 | weight_proj2:   [2][d], [2] */
  /* round 6 (bf16, d = 512, ff % 128 == 0): the operand streams of em_ffn_rows_fused for the two feed-forward modules
   * (EmFfnRowsArgs w1p / w2p; host: pack_ffn_rows_w1 / _w2) - with all four set, each module (+ residual + the LayerNorm
   * that follows: norm_mha after the macaron one; norm_final + the next block's norm_ff_macaron / after_norm after the
   * other) is ONE row-block launch when its rounds fill the chip (EM_ENC_IN_FLIGHT); NULL: two GEMMs + LayerNorm */
  const void *ffm_w1p, *ffm_w2p, *ff_w1p, *ff_w2p;
  const void* wqkvp; /* [linear_q | linear_k | linear_v] in the w1p layout: walked behind the macaron launch (EmFfnRowsArgs.post_q), or NULL */
} EmEBranchformerLayer;

#define EM_MERGE_CONCAT 0      /* x += merge_proj([x1 | x2]); also fixed_ave, whose host packing is
                                  merge_w = [(1 - cgmlp_weight) W | cgmlp_weight W] (:271-276) */
#define EM_MERGE_LEARNED_AVE 1 /* x += merge_proj(w1 x1 + w2 x2), (w1, w2) per utterance (:212-270) */

typedef struct EmEBranchformerWeights {
  int32_t d, heads, ff, cg, num_blocks, cg_kernel, merge_kernel, n_mels;
  const float *conv1_w, *conv1_b;
  const void* conv2_w;
  const float* conv2_b;
  const void* embed_w;
  const float* embed_b;
  const void* wpos_all; /* [num_blocks*d][d] act */
  const float *after_norm_g, *after_norm_b;
  const EmEBranchformerLayer* layers;
  /* Branchformer (espnet2/asr/encoder/branchformer_encoder.py:49-290, merge_method="concat") is the same
   * layer without the feed-forward modules and without the depthwise conv in front of merge_proj:   */
  int32_t use_ffn;    /* 1: macaron + final FFN (E-Branchformer); 0: none (ffm_ / ff_ / norm_ff fields unused) */
  int32_t merge_conv; /* 1: x += merge_proj(cat + dwconv(cat)); 0: x += merge_proj(cat) */
  int32_t subsample;  /* as EmConformerWeights.subsample */
  const void* conv3_w;
  const float* conv3_b;
  int32_t legacy_relpos; /* as EmConformerWeights.legacy_relpos */
  int32_t merge_method;  /* EM_MERGE_* (Branchformer; requires merge_conv = 0 unless EM_MERGE_CONCAT) */
  const void* conv1_wf;  /* as EmConformerWeights: operands of the fused conv1 + conv2 kernel (bf16, d = 256 / 512), or NULL */
  const void* conv2_wf;
} EmEBranchformerWeights;

/* ---- Branchformer learned_ave merge input (BranchformerEncoderLayer.forward, branchformer_encoder.py:212-270).
 *   cat [B*T][2d] act = [x1 | x2] (attention | cgMLP branch outputs); lens [B] valid frames (the pad mask).
 *   Per utterance and branch k: score_t = softmax over the valid frames of (x_k[t] . pool_w[k] + pool_b[k]) /
 *   sqrt(d); weight_k = (sum_t score_t x_k[t]) . wproj_w[k] + wproj_b[k]; (w1, w2) = softmax(weight_1, weight_2).
 *   mw_ws [B][2] f32: receives (weight_1, weight_2); out [B*T][d] act = w1 x1 + w2 x2.                      */
int em_branch_learned_ave(int dtype, const void* cat, const int32_t* lens, int32_t B, int32_t T, int32_t d,
                          const float* pool_w, const float* pool_b, const float* wproj_w,
                          const float* wproj_b, float* mw_ws, void* out, void* stream);

size_t em_ebranchformer_workspace_bytes(int dtype, const EmEBranchformerWeights* w, int32_t B, int32_t T_f);
/*   Arguments as em_conformer_encode.  flags: EM_ENC_ISOLATE_UTTS masks both depthwise convs (cgMLP and
 *   merge) by olens; 0 reproduces the reference's unmasked convs over a padded batch.              */
int em_ebranchformer_encode(int dtype, const EmEBranchformerWeights* w, const float* feats,
                            const float* mvn_partial, const int32_t* flens, const int32_t* olens,
                            int32_t B, int32_t T_f, const void* pos_emb, void* workspace,
                            size_t workspace_bytes, float* enc_out, void* enc_act, int32_t flags,
                            void* stream);

/* ---- A11 + G1 assembled: ctc_lo GEMM (+ arg-max fused in its epilogue) -> collapse.
 *   enc_act [B*T][d] act; w_ctc [V][d] act; logits_ws >= B*T * 2*ceil(V/128) * 2 f32 scratch;
 *   ids [B][T] i32 per-frame argmax; tokens [B][T] i32 (-1 padded); out_lens [B] i32            */
int em_ctc_greedy(int dtype, const void* enc_act, const void* w_ctc, const float* b_ctc, int32_t B,
                  int32_t T, int32_t d, int32_t V, const int32_t* olens, int32_t blank,
                  int32_t sos_eos, float* logits_ws, int32_t* ids, int32_t* tokens,
                  int32_t* out_lens, void* stream);

/* ---- A14: attention-decoder step (TransformerDecoder.batch_score / forward_one_step,
 *      espnet2/asr/decoder/transformer_decoder.py:191-311; DecoderLayer.forward with cache,
 *      transformer/decoder_layer.py:73-179; MultiHeadedAttention, transformer/attention.py:77-265).
 *      K/V are true caches (identical values to the reference's per-step re-projection):
 *      self-attention K/V live in a token-tree cache indexed through `anc`, source-attention
 *      K / V^T of the encoder memory are computed once per utterance by em_search_init.          */
/*   x[r] = embed[tok_row[r]] * sqrt(d) + pe[pos]  (embedding.py:93);  embed [V][d], pe [>pos][d] f32 */
/*   pos_dev != NULL: the position is read from device memory (hipGraph-captured step) and
 *   tok_row is then the base of a [pos][n] token table; pe_len bounds the position.              */
int em_dec_embed_f32(const float* embed, const float* pe, const int32_t* tok_row, int32_t n,
                     int32_t V, int32_t d, int32_t pos, const int32_t* pos_dev, int32_t pe_len,
                     float* x, void* stream);
/*   qkv [n][3d] act (q|k|v of the token at position pos); kc,vc [Lmax][n][d] act (one layer);
 *   anc [n][Lmax] i32 slot of every prefix position; ctx [n][d] act out.  d/heads in {32, 64}.    */
/*   pos_dev != NULL: position from device memory; anc is then used at even and anc_odd at odd
 *   positions (the search double-buffers the table by step parity).  group (1..16) consecutive rows
 *   share a workgroup (the beam of one utterance: shared ancestors hit the same cache).
 *   tok_tab != NULL ([Lmax][n] token table): keys whose token id is 0 are masked
 *   (TransformerLM._target_mask, espnet2/lm/transformer_lm.py:54-57).                            */
int em_dec_self_attention(int dtype, const void* qkv, void* kc, void* vc, const int32_t* anc,
                          const int32_t* anc_odd, int32_t n, int32_t d, int32_t heads, int32_t Lmax,
                          int32_t pos, const int32_t* pos_dev, int32_t group, const int32_t* tok_tab,
                          void* ctx, void* stream);
/*   The same attention for rows that are B x W beams (W consecutive rows per utterance, no token mask): where the shape
 *   allows (EM_BF16, d / heads = 64, W <= 16, Lmax <= 512 with a key list within 64 KiB of LDS, n <= 65 535) the keys are the UNION of the beam's ancestors -
 *   every distinct cache row under the W hypotheses is gathered once and scored for all of them on the matrix cores, a
 *   hypothesis seeing the keys on its own path (csrc/decoder.hip dec_self_attn_tree_kernel); other shapes run
 *   em_dec_self_attention with group = (W + 1) / 2.  Same arguments otherwise; the result differs from the per-row
 *   kernel by bf16 round-off of the probabilities (MultiHeadedAttention over the cached prefix, attention.py:121-151). */
int em_dec_self_attention_beam(int dtype, const void* qkv, void* kc, void* vc, const int32_t* anc,
                               const int32_t* anc_odd, int32_t n, int32_t d, int32_t heads, int32_t Lmax,
                               int32_t pos, const int32_t* pos_dev, int32_t W, void* ctx, void* stream);
/*   TransformerLM input (espnet2/lm/transformer_lm.py:35, transformer/encoder.py:132-139):
 *   e[r] = embed[tok_row[r]] in the act dtype;  then (after the input Linear) in place on x f32:
 *   torch LayerNorm(eps 1e-5) -> ReLU -> optional x*sqrt(d) + pe[pos].  pos_dev as above.       */
int em_lm_embed(int dtype, const float* embed, const int32_t* tok_row, int32_t n, int32_t V,
                int32_t embed_unit, const int32_t* pos_dev, int32_t Lmax, void* e, void* stream);
int em_lm_input_norm_f32(float* x, const float* g, const float* b, const float* pe, int32_t n,
                         int32_t d, int32_t pos, const int32_t* pos_dev, int32_t Lmax, void* stream);
/*   qs [B*W][d] act; kmem: K rows of utterance b at kmem + (b*T + t)*ldk; vT [B][d][Tpad] act
 *   (zero padded, Tpad % 32 == 0); klens [B] valid memory frames; ctx [B*W][d] act out.           */
int em_dec_src_attention(int dtype, const void* qs, const void* kmem, int32_t ldk, const void* vT,
                         const int32_t* klens, int32_t B, int32_t W, int32_t d, int32_t heads,
                         int32_t T, int32_t Tpad, void* ctx, void* stream);
/*   ... with the pre-norm LayerNorm and the query projection in the kernel's prologue (bf16, d_k = 64, d = 256 | 512):
 *   qs = (LN(x; g, be, eps) . wq^T + bq) rounded to bf16, x [B*W][d] f32, wq [d][d] act - DecoderLayer.forward's
 *   norm2 + src_attn.linear_q (decoder_layer.py:119-121, attention.py:94) without a launch of their own; the same q bit
 *   for bit as em_ln_gemm followed by em_dec_src_attention.  EM_ERR_UNSUPPORTED for other shapes / EM_F32.            */
int em_dec_src_attention_lnq(int dtype, const float* x, const float* g, const float* be, float eps, const void* wq,
                             const float* bq, const void* kmem, int32_t ldk, const void* vT, const int32_t* klens,
                             int32_t B, int32_t W, int32_t d, int32_t heads, int32_t T, int32_t Tpad, void* ctx,
                             void* stream);
/*   The feed-forward sublayer of a decoder label step on fragment-major operands (bf16; csrc/dec_ffn.hip, two launches):
 *     x[n][d] f32  +=  w2 . relu(w1 . LN(x; ln_g, ln_b, eps) + b1) + b2
 *   = DecoderLayer.forward's norm3 -> feed_forward -> residual (decoder_layer.py:150-158, pre-norm),
 *   PositionwiseFeedForward.forward (positionwise_feed_forward.py:30-32).  w1f [ff][d] and w2f [d][ff] are bf16 and
 *   FRAGMENT-MAJOR: a [R][K] matrix as [R/16][K/32][lane = 16 * (k % 32 / 8) + r % 16][k % 8], the 16 rows x 32 k of one
 *   MFMA operand 1 KiB contiguous (host: espnet_amd.lib.pack_frag16; EmDecoderLayer.w1_frag / w2_frag).  hbuf: n * ff
 *   bf16 of scratch (the hidden activation, written fragment-major).  d = 256 | 512, ff % 128 == 0, n % 16 == 0;
 *   EM_ERR_UNSUPPORTED otherwise and for EM_F32 (em_dec_ffn_split tells beforehand).                                   */
int em_dec_ffn(int dtype, float* x, const float* ln_g, const float* ln_b, float eps, const void* w1f, const float* b1,
               const void* w2f, const float* b2, int32_t n, int32_t d, int32_t ff, void* hbuf, void* stream);
/*   hidden units per workgroup of em_dec_ffn's first launch for (n, d, ff); 0 = shape not covered (keep em_ln_gemm +
 *   em_gemm on the row-major matrices)                                                                               */
int em_dec_ffn_split(int32_t n, int32_t d, int32_t ff);
/*   LayerNorm + projection of a label step on a FRAGMENT-MAJOR weight (bf16; csrc/dec_ffn.hip):
 *     out[n][N] = epilogue(LN(x[n][d]; ln_g, ln_b, eps) . wf^T + bias),   x f32, wf [N][d] fragment-major with its rows
 *   ZERO-PADDED to a multiple of 512 (espnet_amd.lib.pack_frag16(w, pad_rows=512)), bias [N] f32;
 *   out_mode EM_LNF_RELU_FRAG: ReLU, bf16, written fragment-major (n % 16 == 0, N % 128 == 0: em_dec_ffn's hidden activation);
 *            EM_LNF_STORE: bf16 rows [n][N];  EM_LNF_STORE_F32: f32 rows [n][N] (the vocabulary logits).
 *   d = 256 | 512, N % 4 == 0; EM_ERR_UNSUPPORTED otherwise (callers keep em_ln_gemm / em_layernorm + em_gemm).          */
#define EM_LNF_RELU_FRAG 0
#define EM_LNF_STORE 1
#define EM_LNF_STORE_F32 2
int em_ln_gemm_frag(int out_mode, const float* x, const float* ln_g, const float* ln_b, float eps, const void* wf,
                    const float* bias, void* out, int32_t n, int32_t N, int32_t d, void* stream);
/*   ... the same with wq FRAGMENT-MAJOR (espnet_amd.lib.pack_frag16; EmDecoderLayer.src_wq_frag): bit for bit the same context */
int em_dec_src_attention_lnq_frag(int dtype, const float* x, const float* g, const float* be, float eps, const void* wq_frag,
                                  const float* bq, const void* kmem, int32_t ldk, const void* vT, const int32_t* klens,
                                  int32_t B, int32_t W, int32_t d, int32_t heads, int32_t T, int32_t Tpad, void* ctx,
                                  void* stream);
/*   vT[b][c][t] = kv[(b*T + t)*2d + d + c]                                                       */
int em_dec_transpose_v(int dtype, const void* kv, int32_t B, int32_t T, int32_t d, int32_t Tpad,
                       void* vT, void* stream);

typedef struct EmDecoderLayer {
  const float *norm1_g, *norm1_b, *norm2_g, *norm2_b, *norm3_g, *norm3_b;
  const void* self_wqkv; /* [3d][d] act: self_attn.linear_q | linear_k | linear_v */
  const float* self_bqkv;
  const void* self_wout; /* [d][d] act */
  const float* self_bout;
  const void* src_wq; /* [d][d] act */
  const float* src_bq;
  const void* src_wkv; /* [2d][d] act: src_attn.linear_k | linear_v (used once per utterance) */
  const float* src_bkv;
  const void* src_wout;
  const float* src_bout;
  const void* w1; /* [ff][d] act (ReLU FFN) */
  const float* b1;
  const void* w2; /* [d][ff] act */
  const float* b2;
  /* round 6: bf16 fragment-major copies (espnet_amd.lib.pack_frag16) for the label step's projections on 1 KiB operand
   * loads, or NULL (then the row-major matrix above is used): w1 / w2 (em_dec_ffn), self_wqkv (em_ln_gemm_frag, rows
   * padded to a multiple of 512), self_wout / src_wout (mid_gemm with a fragment-major W)                            */
  const void *w1_frag, *w2_frag, *self_wqkv_frag, *self_wout_frag, *src_wout_frag;
  const void* src_wq_frag; /* src_wq fragment-major (em_dec_src_attention_lnq_frag), or NULL */
} EmDecoderLayer;

typedef struct EmDecoderWeights {
  int32_t d, heads, ff, num_blocks, vocab, pe_len;
  const float* embed; /* [V][d] f32: decoder.embed.0.weight */
  const float* pe;    /* [pe_len][d] f32 absolute sinusoid table (embedding.py:56-79) */
  const float *after_norm_g, *after_norm_b;
  const void* out_w; /* [V][d] act: decoder.output_layer */
  const float* out_b;
  const EmDecoderLayer* layers; /* [num_blocks], host array */
  const void* out_w_frag; /* round 6: out_w fragment-major, rows zero-padded to a multiple of 512 (em_ln_gemm_frag), or NULL */
} EmDecoderWeights;

/* ---- A12 + A13: label-synchronous joint CTC/attention beam search, batched over utterances
 *      (BatchBeamSearch.search / post_process, espnet2/legacy/nets/batch_beam_search.py:253-423;
 *      BeamSearch.forward, legacy/nets/beam_search.py:385-498; CTCPrefixScoreTH.__call__,
 *      legacy/nets/ctc_prefix_score.py:71-191; end_detect, legacy/nets/e2e_asr_common.py:14-44).
 *      n = B*W rows (row = b*W + k).  Every buffer is caller-owned device memory.               */
typedef struct EmSearchParams {
  int32_t B, W, V;
  int32_t T, Tpad;     /* memory frames of the batch (max) and its multiple-of-32 padding */
  int32_t S;           /* pre-beam size int(1.5*W), or V when there is no pre-beam (beam_search.py:105-119) */
  int32_t NC;          /* candidate slots per row: S+1 (the extra slot is <eos>) or V */
  int32_t Lmax;        /* token positions held per hypothesis (>= max maxlen + 2) */
  int32_t end_cap;     /* capacity of the per-utterance ended list */
  int32_t sos, eos, blank;
  int32_t use_end_detect; /* maxlenratio == 0 (beam_search.py:443) */
  float w_dec, w_ctc, w_len; /* scorer weights: decoder, ctc, length_bonus (asr_inference.py:310-316) */
  float w_lm;                /* language-model scorer weight (lm_weight), 0 = no LM */
  int32_t ldT;               /* frame stride of ctc_lpT / r_a / r_b; 0 = T.  A stream (B == 1) whose
                                visible length T grows block by block keeps a fixed capacity here */
} EmSearchParams;

/* TransformerLM used as a full scorer (espnet2/lm/transformer_lm.py:12-137):
 * nn.Embedding -> Encoder(input_layer="linear") -> nn.Linear head.  "act" = dtype of the call.  */
typedef struct EmLmLayer {
  const float *norm1_g, *norm1_b, *norm2_g, *norm2_b;
  const void* wqkv; /* [3d][d] act: self_attn.linear_q | linear_k | linear_v */
  const float* bqkv;
  const void* wout; /* [d][d] act */
  const float* bout;
  const void* w1; /* [ff][d] act (ReLU) */
  const float* b1;
  const void* w2; /* [d][ff] act */
  const float* b2;
} EmLmLayer;

typedef struct EmLmWeights {
  int32_t d, heads, ff, num_blocks, vocab, embed_unit;
  const float* embed;               /* [V][embed_unit] f32: lm.embed.weight */
  const void* in_w;                 /* [d][embed_unit] act: lm.encoder.embed.0 */
  const float *in_b, *in_ln_g, *in_ln_b; /* encoder.embed.0 bias, encoder.embed.1 (torch LayerNorm, eps 1e-5) */
  const float* pe;                  /* [>= Lmax][d] sinusoid table, or NULL (pos_enc=None) */
  const float *after_norm_g, *after_norm_b;
  const void* out_w;                /* [V][d] act: lm.decoder */
  const float* out_b;
  const EmLmLayer* layers;          /* [num_blocks], host array */
  /* kind >= EM_LM_LSTM: SequentialRNNLM (espnet2/lm/seq_rnn_lm.py:14-177; rnn_type lstm | gru | rnn_tanh |
   * rnn_relu, :40-58).  d = nhid padded
   * to the GEMM K step (ld of every hidden-state row), embed_unit = unit padded likewise (embed is
   * [V][embed_unit] with zero pad columns), num_blocks = nlayers, out_w [V][d] act, out_b; the
   * transformer-only fields are NULL.                                                              */
  int32_t kind;                     /* EM_LM_TRANSFORMER | EM_LM_LSTM | EM_LM_GRU | EM_LM_RNN_TANH | EM_LM_RNN_RELU */
  int32_t nhid;                     /* true hidden size (<= d) */
  const struct EmRnnLayer* rnn;     /* [num_blocks], host array */
} EmLmWeights;

#define EM_LM_TRANSFORMER 0
#define EM_LM_LSTM 1
#define EM_LM_GRU 2
#define EM_LM_RNN_TANH 3
#define EM_LM_RNN_RELU 4
/* One recurrent layer; G gate blocks of nhid rows each, in_pad = embed_unit (layer 0) or d.
 *   LSTM (G = 4): PyTorch order i | f | g | o, bias = bias_ih + bias_hh.
 *   GRU  (G = 4): r | z | n_x | n_h.  torch.nn.GRU's candidate n = tanh(W_in x + b_in + r * (W_hn h + b_hn))
 *        needs its input and hidden parts apart, so w_ih = [W_ir; W_iz; W_in; 0], w_hh = [W_hr; W_hz; 0; W_hn],
 *        bias = [b_ir + b_hr; b_iz + b_hz; b_in; b_hn] (packed by the host, lm/seq_rnn_lm.py).
 *   RNN  (G = 1): bias = bias_ih + bias_hh.                                                              */
typedef struct EmRnnLayer {
  const void* w_ih;  /* [G*nhid][in_pad] act */
  const void* w_hh;  /* [G*nhid][d] act */
  const float* bias; /* [G*nhid] */
} EmRnnLayer;

typedef struct EmSearchBuffers {
  const int32_t *xlens, *maxlens, *minlens; /* [B] valid memory frames, max / min output length */
  float* ctc_lpT;                           /* [V][B*T] CTC.log_softmax(enc) (scorers/ctc.py:96), TRANSPOSED:
                                               the T frames of one (utterance, label) are contiguous;
                                               written by em_search_init */
  int32_t *tok, *parent;                    /* [Lmax][n] token tree */
  int32_t *anc_a, *anc_b;                   /* [n][Lmax] ancestor slots, double buffered by step parity */
  int32_t* alive;                           /* [n] */
  float *run_score, *run_sdec, *run_sctc, *run_slen, *s_prev; /* [n] */
  float *r_a, *r_b;                         /* [n][T][2] CTC forward variables (r^n, r^b) */
  int32_t* cand_tok;                        /* [n][NC] */
  float *cand_full, *cand_psi, *cand_total; /* [n][NC] */
  int32_t* sel_idx;                         /* [n] */
  float* sel_total;                         /* [n] */
  int32_t *end_count;                       /* [B] */
  int32_t *end_pos, *end_slot, *end_forced; /* [B][end_cap] tree node of the last token; forced <eos> flag */
  float *end_score, *end_sdec, *end_sctc, *end_slen; /* [B][end_cap] */
  float *best_all, *best_by_len;            /* [B], [B][Lmax+2] end-detection statistics */
  int32_t* done;                            /* [B] */
  int32_t* step;                            /* [2] device step counter + arrival ticket, or NULL.  Non-NULL = graph
                                               mode: every step kernel reads the step index from step[0] (kernel
                                               arguments are frozen in a captured hipGraph) and the last
                                               kernel of a step advances it (step[1] counts its workgroups) */
  float* x;                                 /* [n][d] f32 decoder residual stream */
  void *xn, *qkv, *qs, *ctx, *hbuf;         /* act: [n][d], [n][3d], [n][d], [n][d], [n][ff] */
  float* dec_logp;                          /* [n][V] */
  void *self_k, *self_v;                    /* act [layers][Lmax][n][d] */
  void *mem_kv;                             /* act [layers][B*T][2d] */
  void *mem_vT;                             /* act [layers][B][d][Tpad], zero initialised by the caller */
  /* language-model scorer (all NULL when w_lm == 0) */
  const EmLmWeights* lm;                    /* host struct */
  void *lm_e, *lm_xn, *lm_qkv, *lm_ctx, *lm_h; /* act: [n][embed_unit], [n][d], [n][3d], [n][d], [n][ff] */
  float *lm_x, *lm_logp;                    /* f32: [n][d], [n][V] */
  void *lm_k, *lm_v;                        /* act [lm layers][Lmax][n][d] */
  float *run_slm, *end_slm;                 /* [n], [B][end_cap] accumulated LM score */
  /* recurrent language model (kind >= EM_LM_LSTM; NULL otherwise).  States of the rows of step i live in
   * ring slot i % 3; a row reads its parent's state (token-tree `parent`) from slot (i-1) % 3; the third
   * slot keeps step i-2 alive for the streaming search's one-step rewind.                             */
  void* rnn_hs;                             /* act [3][layers][n][d] hidden states (pad columns zero) */
  float* rnn_cs;                            /* f32 [3][layers][n][d] master state: LSTM c, GRU h */
  void* rnn_hin;                            /* act [layers][n][d] parent-gathered h, then this step's h */
  float* rnn_gates;                         /* f32 [n][G*nhid] */
  /* streaming search (em_search_online_*; NULL offline) */
  float *online_best;                       /* [n][8] valid, parent slot, token, total, dec, ctc, len, lm */
  float *online_psi;                        /* [n] log psi of the selected candidates (next s_prev) */
  float *online_snap;                       /* [n][8] per-row scalars of prev_hyps */
  /* round 6 (bf16, d_k = 64; both or neither; NULL: the source attention reads mem_kv / mem_vT): the memory's K and V^T
   * FRAGMENT-MAJOR - the 16 x 32 operand tiles of the source attention's MFMAs contiguous - written by em_search_init beside
   * mem_kv / mem_vT: mem_kf act [layers][B][heads][Tpad/16][2][64][8], mem_vf act [layers][B][heads][4][Tpad/32][64][8]
   * (B * d * Tpad elements per layer each), zero for frames >= T                                                              */
  void *mem_kf, *mem_vf;
} EmSearchBuffers;

/*   Projects the encoder memory (enc_act [B][T][d_model] act) to per-layer K | V and V^T, computes
 *   the transposed CTC log-probs (ctc_w [V][d_model] act, ctc_b [V] f32: ctc.ctc_lo; may be NULL
 *   when w_ctc == 0) and writes the initial search state (one alive <sos> hypothesis per
 *   utterance, CTC r_prev of the empty prefix).                                                  */
int em_search_init(int dtype, const EmSearchParams* p, const EmDecoderWeights* dw,
                   const EmSearchBuffers* b, const void* enc_act, int32_t d_model,
                   const void* ctc_w, const float* ctc_b, void* stream);
/*   Enqueues search steps i0 .. i1-1 (decoder step, pre-beam, CTC prefix scores, top-W, update).
 *   Utterances whose `done` flag is set are skipped; the host polls `done` between calls.
 *   With b->step set, i1 - i0 steps are enqueued starting at the device counter's value, so the
 *   call can be captured once into a hipGraph and replayed; steps past Lmax - 2 are no-ops.      */
int em_search_steps(int dtype, const EmSearchParams* p, const EmDecoderWeights* dw,
                    const EmSearchBuffers* b, int32_t i0, int32_t i1, void* stream);

/* ---- The reference's scorer interface, one call per search step.
 *      em_search_steps keeps a whole search on the device; a search written against
 *      BatchScorerInterface / BatchPartialScorerInterface (espnet2/legacy/nets/scorer_interface.py:29-190) --
 *      the reference's own BatchBeamSearch.search, legacy/nets/batch_beam_search.py:253-357 -- calls its
 *      scorers once per step instead.  These entry points are those calls on the same kernels; the host
 *      mirrors are espnet_amd TransformerDecoder.batch_score, CTCPrefixScorer.batch_score_partial /
 *      select_state, TransformerLM / SequentialRNNLM.batch_score.                                        */
/*   K | V and V^T of the encoder memory for every decoder layer (MultiHeadedAttention.forward_qkv of
 *   src_attn, transformer/attention.py:77-119; the reference recomputes it per hypothesis and step):
 *   enc_act [B][T][d] act -> mem_kv act [layers][B*T][2d], mem_vT act [layers][B][d][Tpad] (zeroed by
 *   the caller, Tpad % 32 == 0).                                                                        */
int em_decoder_memory(int dtype, const EmDecoderWeights* dw, const void* enc_act, int32_t B, int32_t T,
                      int32_t Tpad, void* mem_kv, void* mem_vT, void* stream);
/*   TransformerDecoder.forward_one_step / batch_score (espnet2/asr/decoder/transformer_decoder.py:191-311)
 *   for n = B*W hypotheses (W rows per memory): consumes the token at position `pos` of every row, appends
 *   that position's self-attention K/V to the caller's cache and writes the next-token LOGITS [n][V]
 *   (log-softmax: em_log_softmax_rows_f32).  The cache is the persistent handle: self_k / self_v act
 *   [layers][Lmax][n][d], `anc` [n][Lmax] names the cache slot holding position j of row r's prefix
 *   (identity r for independent rows), tok [Lmax][n] the token table (row `pos` is read).              */
typedef struct EmDecoderStepArgs {
  int32_t B, W, T, Tpad, Lmax, pos;
  const int32_t *tok, *anc, *xlens;  /* [Lmax][n], [n][Lmax], [B] valid memory frames */
  void *self_k, *self_v;
  const void *mem_kv, *mem_vT;       /* em_decoder_memory */
  float* x;                          /* [n][d] f32 workspace */
  void *xn, *qkv, *qs, *ctx, *hbuf;  /* act workspaces [n][d], [n][3d], [n][d], [n][d], [n][ff] */
  float* logits;                     /* [n][V] out */
} EmDecoderStepArgs;
int em_decoder_step(int dtype, const EmDecoderWeights* dw, const EmDecoderStepArgs* a, void* stream);
/*   The language-model stage of search step i on caller-provided search state (TransformerLM.batch_score
 *   espnet2/lm/transformer_lm.py:103-137, SequentialRNNLM.batch_score espnet2/lm/seq_rnn_lm.py:140-177):
 *   reads p->{B,W,V,Lmax} and b->{lm, tok, anc_a, anc_b, parent, lm_*, rnn_*}; writes LOGITS to
 *   b->lm_logp [n][V] and the step's K/V (or ring-slot i % 3 recurrent state).  b->step must be NULL.     */
int em_lm_step(int dtype, const EmSearchParams* p, const EmSearchBuffers* b, int32_t i, void* stream);
/*   CTC.log_softmax(enc) (espnet2/asr/ctc.py:197-205, scorers/ctc.py:96-98) TRANSPOSED: lpT [V][B*T] f32. */
int em_ctc_log_probs_t(int dtype, const void* enc_act, int32_t B, int32_t T, int32_t d_model,
                       const void* ctc_w, const float* ctc_b, int32_t V, float* lpT, void* stream);
/*   r_prev of the empty prefix (CTCPrefixScoreTH.__call__ state None, ctc_prefix_score.py:87-99):
 *   r0 [B][T][2] = (logzero, cumsum of the blank log-probs).                                            */
int em_ctc_prefix_init(const float* lpT, const int32_t* xlens, int32_t B, int32_t T, int32_t blank,
                       float* r0, void* stream);
/*   CTCPrefixScoreTH.__call__ (legacy/nets/ctc_prefix_score.py:71-191) for n = B*W prefixes of length
 *   out_len (without <sos>): r_prev [n][T][2], s_prev [n], last_ids [n]; cand_ids [n][S] (the pre-beam
 *   ids; slot S of every row is <eos>, which the reference always scores, :184-186) or NULL = all labels
 *   (S == V).  log_psi [n][S+1] (or [n][V]) out; scores = log_psi - s_prev (may be NULL).  The forward
 *   variables of the chosen candidates come from em_ctc_prefix_state (the reference materialises r for
 *   every candidate and indexes it in select_state, scorers/ctc.py:40-63).                              */
int em_ctc_prefix_score(const float* lpT, const int32_t* xlens, const float* r_prev, const float* s_prev,
                        const int32_t* last_ids, const int32_t* cand_ids, int32_t B, int32_t W, int32_t S,
                        int32_t T, int32_t V, int32_t out_len, int32_t eos, int32_t blank, float* log_psi,
                        float* scores, void* stream);
/*   r of prefix(rows[k]) + toks[k] for k < m (ctc_prefix_score.py:131-132,158-164): r_out [m][T][2]; frames
 *   below max(out_len,1)-1 are not written (logzero in the reference; the caller pre-fills).             */
int em_ctc_prefix_state(const float* lpT, const int32_t* xlens, const float* r_prev, const int32_t* last_ids,
                        const int32_t* rows, const int32_t* toks, int32_t m, int32_t B, int32_t W, int32_t T,
                        int32_t out_len, int32_t blank, float* r_out, void* stream);
/*   CTCPrefixScoreTH.extend_state (legacy/nets/ctc_prefix_score.py:248-270; driven by CTCPrefixScorer.extend_state,
 *   legacy/nets/scorers/ctc.py:141-157) for n stand-alone hypothesis states of ONE utterance whose visible memory grew
 *   from T_old to T_new frames: r_old [n][T_old][2] -> r_new [n][T_new][2].  Frames below max(T_old, 1) are copied;
 *   the new frames continue along the blank path only, r[t] = (logzero, r[t-1][1] + lpT[blank][t]), accumulated
 *   left to right like the reference's loop.  lpT [V][T_new] = em_ctc_log_probs_t of the T_new frames (B = 1).   */
int em_ctc_prefix_extend(const float* lpT, int32_t T_new, int32_t blank, const float* r_old, int32_t n,
                         int32_t T_old, float* r_new, void* stream);

/* ---- §8(f) rank 3: block-synchronous streaming search, BatchBeamSearchOnline
 *      (espnet2/legacy/nets/batch_beam_search_online.py:155-534).  The host mirrors the reference's
 *      control flow (block loop :296-376, process_one_block :394-493: repetition / local-<eos> breaks,
 *      rewind, ended lists); these entry points are its device operations on ONE stream (B == 1,
 *      b->step NULL).  State layout as em_search_init; p->T / p->Tpad = frames visible to the current
 *      block, p->ldT = frame capacity; xlens[0] = p->T; maxlens[0] >= Lmax (no forced <eos> on the
 *      device: the host owns the length logic).  em_search_init starts the stream on the first block. */
/*   extend (:518-534): source-attention memory re-projected for the p->T visible frames, CTC
 *   log-probs of all of them (extend_prob, legacy/nets/scorers/ctc.py:128-139), and the forward
 *   variables of the running rows (state before step i) continued over [t_old, T) along the blank
 *   path (extend_state, legacy/nets/ctc_prefix_score.py:248-270).  enc_act [T][d_model] act.        */
int em_search_online_extend(int dtype, const EmSearchParams* p, const EmDecoderWeights* dw,
                            const EmSearchBuffers* b, const void* enc_act, int32_t d_model,
                            const void* ctc_w, const float* ctc_b, int32_t i, int32_t t_old, void* stream);
/*   best = search(running_hyps, h) (:400, BatchBeamSearch.search batch_beam_search.py:253-357) for
 *   step i: scorers, pre-beam, CTC prefix scores, top-W -> online_best.  No search state is modified. */
int em_search_online_core(int dtype, const EmSearchParams* p, const EmDecoderWeights* dw,
                          const EmSearchBuffers* b, int32_t i, void* stream);
/*   prev_hyps = running_hyps; running_hyps = post_process(best) (:459-462): snapshot, forward
 *   variables of the winners, tree / ancestor / score update; rows ending in <eos> leave the beam.    */
int em_search_online_commit(int dtype, const EmSearchParams* p, const EmSearchBuffers* b, int32_t i,
                            void* stream);
/*   running_hyps = prev_hyps (:484-487); the caller decrements its position.                          */
int em_search_online_rewind(const EmSearchParams* p, const EmSearchBuffers* b, void* stream);

/* ---- A16: streaming (contextual block) Conformer encoder step
 *      (ContextualBlockConformerEncoder.forward_infer, espnet2/asr/encoder/
 *      contextual_block_conformer_encoder.py:386-600; ContextualBlockEncoderLayer.forward_infer,
 *      legacy/nets/pytorch_backend/conformer/contextual_block_encoder_layer.py:197-310).
 *      Buffering across calls is host logic (espnet_amd/asr/encoder/
 *      contextual_block_conformer_encoder.py); these entry points do the arithmetic of one call on
 *      static shapes, so the host can capture a steady-state call in a hipGraph.                 */
/*   out[j] = xs[j]*sqrt(d) + pe[start+j]  (StreamPositionalEncoding.forward, embedding.py:376-389) */
int em_stream_pos_enc_f32(const float* xs, const float* pe, int32_t start, int32_t n, int32_t d,
                          float* out, void* stream);
/*   Block assembly (:512-536): xs [total][d] f32 subsampled frames, pe [..][d] sinusoid table,
 *   prev_addin [d] context carried from the previous call or NULL; n_proc = blocks processed by
 *   earlier calls, read from n_proc_dev (device i32) when that is non-NULL -- a hipGraph-captured
 *   step needs it in device memory; x [n_blk][bs+2][d] f32 out, addin_out [d] = context of the
 *   last block.                                                                                  */
int em_cb_build_blocks_f32(const float* xs, const float* pe, const float* prev_addin, int32_t n_proc,
                           const int32_t* n_proc_dev, int32_t n_blk, int32_t total, int32_t bs,
                           int32_t hs, int32_t d, float* x, float* addin_out, void* stream);
/*   MultiHeadedAttention inside each block (attention.py:121-151): qkv [n_blk*L][3d] act,
 *   mask_mode 1 = contextual mask (:539-544), 0 = none; ctx [n_blk*L][d] act.  L <= 64.          */
int em_block_mha(int dtype, const void* qkv, int32_t n_blk, int32_t L, int32_t d, int32_t heads,
                 int32_t mask_mode, void* ctx, void* stream);
/*   Context hand-over after a layer (layer :292-304), in place on x [n_blk][L][d] f32.           */
int em_cb_propagate_ctx_f32(float* x, const float* past_ctx, float* next_ctx, int32_t n_blk,
                            int32_t L, int32_t d, void* stream);
size_t em_cb_workspace_bytes(int dtype, const EmConformerWeights* w, int32_t n_blk, int32_t L);
/*   All layers on x [n_blk][L][d] f32 in place.  EmConformerLayer is reused (norm_mha = norm1,
 *   norm_ff = norm2, pos_u/pos_v unused; ReLU feed-forward).  past_ctx / next_ctx [num_blocks][d]: give them DIFFERENT
 *   buffers - with one block per stream the hand-over rides in the block launches (layer l + 1 reads past_ctx[l] after layer
 *   l wrote next_ctx[l]); a call with past_ctx == next_ctx is still correct, it falls back to the hand-over launch.  */
int em_cb_encode_blocks(int dtype, const EmConformerWeights* w, float* x, int32_t n_blk, int32_t L,
                        int32_t mask_mode, const float* past_ctx, float* next_ctx, void* workspace,
                        size_t workspace_bytes, void* stream);

/*   Batch of n_streams LOCK-STEP streams (a server feeding equal chunks to many streams at once): the same two calls
 *   with a leading stream dimension - xs [n_streams][total][d], prev_addin / addin_out [n_streams][d] (prev_addin NULL
 *   on the streams' first call), x [n_streams][n_blk][bs+2][d]; past_ctx / next_ctx [n_streams][num_blocks][d];
 *   workspace for n_streams * n_blk blocks.  The dense operators see n_streams * n_blk independent blocks; only block
 *   assembly and the context hand-over between layers know the streams.                                         */
int em_cb_build_blocks_batch_f32(const float* xs, const float* pe, const float* prev_addin, int32_t n_proc,
                                 int32_t n_streams, int32_t n_blk, int32_t total, int32_t bs, int32_t hs, int32_t d,
                                 float* x, float* addin_out, void* stream);
/*   ... for streams that did NOT start together (a server's connections join and leave: the reference keeps the state
 *   per object, espnet2/bin/asr_inference_streaming.py:205-336, so nothing forbids it): the number of blocks a stream
 *   has already processed - it sets the positional-encoding offsets of its blocks - per stream, n_proc_rows [n_streams]
 *   i32 on the device.  Everything else as em_cb_build_blocks_batch_f32.                                            */
int em_cb_build_blocks_rows_f32(const float* xs, const float* pe, const float* prev_addin, const int32_t* n_proc_rows,
                                int32_t n_streams, int32_t n_blk, int32_t total, int32_t bs, int32_t hs, int32_t d,
                                float* x, float* addin_out, void* stream);
int em_cb_encode_blocks_batch(int dtype, const EmConformerWeights* w, float* x, int32_t n_streams, int32_t n_blk,
                              int32_t L, int32_t mask_mode, const float* past_ctx, float* next_ctx, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---- optional per-launch timing of the GEMM kernel family (measurement only; bench.py's
 *      `roofline` leg).  While a profile is attached to the calling thread every em_gemm launch
 *      (direct or from em_conformer_encode / em_ctc_greedy) is bracketed by hipEventRecord on its
 *      stream.  em_profile_read waits for the last event, returns per-launch milliseconds and
 *      2*M*N*K flops, and resets the profile.                                                    */
typedef struct EmProfile EmProfile;
EmProfile* em_profile_create(int32_t capacity);
void em_profile_destroy(EmProfile* prof);
void em_profile_attach(EmProfile* prof); /* NULL detaches */
int em_profile_read(EmProfile* prof, float* ms, double* flops, int32_t max_n, int32_t* count);
/*   the same with the kernel family of every record: EM_PROF_GEMM (em_gemm), EM_PROF_BLOCK
 *   (em_conformer_block_fused: flops of its GEMM-shaped stages), EM_PROF_ATTN (em_relpos_attention2_bf16:
 *   6 B h T^2 d_k for AC, BD and P.V).  tags may be NULL.                                          */
#define EM_PROF_GEMM 0
#define EM_PROF_BLOCK 1
#define EM_PROF_ATTN 2
#define EM_PROF_ROWS 3 /* em_ffn_rows_fused: flops of its feed-forward / projection / GLU GEMMs */
int em_profile_read2(EmProfile* prof, float* ms, double* flops, int32_t* tags, int32_t max_n, int32_t* count);

/* f32 -> act dtype copy (lets reference-shaped f32 entry points feed the act-dtype GEMMs) */
int em_cast_f32(int dtype, const float* src, size_t n, void* dst, void* stream);

/* ---- host audio reader of the decode CLI (host code only, no GPU needed):
 *      `soundfile.read(path, dtype=float32)` of espnet2/fileio/sound_scp.py:13-155 / the `sound` entry of
 *      espnet2/train/iterable_dataset.py:44-67, fused with CommonCollateFn's zero padding
 *      (espnet2/train/collate_fn.py:17-95): every file is decoded straight into its row of the batch matrix.
 *      Handled: RIFF/WAVE, mono, PCM 8/16/24/32-bit (value / 2^(bits-1), 8-bit unsigned offset 128) and IEEE
 *      float 32/64; FLAC (RFC 9639; the recipes' default audio_format, egs2/TEMPLATE/asr1/asr.sh:56), mono with
 *      the length in STREAMINFO, every subframe type, CRC-8 / CRC-16 verified, value / 2^(bits-1).  Anything
 *      else sets that file's status to EM_ERR_UNSUPPORTED (EM_ERR_IO if unreadable or a CRC fails) and the
 *      caller reads it through the Python reader (espnet_amd/fileio/sound_scp.py), which names the problem. */
#define EM_AUDIO_FORMAT_FLAC 0xF1AC /* EmWavInfo.format of a FLAC stream (WAVE tags are 1 = PCM, 3 = float) */
typedef struct EmWavInfo {
  int64_t frames;      /* samples per channel in the data chunk (clipped to the file size) */
  int64_t data_offset; /* byte offset of the first sample */
  int32_t rate, channels, bits;
  int32_t format;      /* WAVE format tag: 1 PCM, 3 IEEE float (WAVE_FORMAT_EXTENSIBLE resolved); EM_AUDIO_FORMAT_FLAC */
  int32_t status;      /* EM_OK or the EM_ERR_* code of this file */
} EmWavInfo;
/*   Parse the headers of n files (paths: NUL-terminated strings) with up to `threads` threads.  Returns EM_OK
 *   or one of the per-file error codes; info[i].status tells which files.                                    */
int em_wav_probe(const char* const* paths, int32_t n, EmWavInfo* info, int32_t threads);
/*   out [n][ld] f32: row i <- the info[i].frames samples of file i, zero filled up to ld (>= every frames).   */
int em_wav_load_rows(const char* const* paths, const EmWavInfo* info, int32_t n, float* out, int64_t ld,
                     int32_t threads);

#ifdef __cplusplus
}
#endif
#endif /* ESPNET_AMD_H_ */
