"""ctypes binding of libespnet_amd.so (the C ABI declared in include/espnet_amd.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol is absent this
module raises at first use, and every wrapper raises on a non-zero status.  Status codes are mapped
to the exceptions the reference raises for the same condition (TooShortUttError for < 7 feature
frames, espnet2/legacy/nets/pytorch_backend/transformer/subsampling.py:14-49).
"""
import ctypes as C
import os
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libespnet_amd.so"

EM_OK = 0
EM_LNF_RELU_FRAG, EM_LNF_STORE, EM_LNF_STORE_F32 = 0, 1, 2
EM_ERR_UNSUPPORTED, EM_ERR_BAD_ARG, EM_ERR_TOO_SHORT, EM_ERR_LAUNCH, EM_ERR_WORKSPACE, EM_ERR_IO = -1, -2, -3, -4, -5, -6
EM_F32, EM_BF16 = 0, 1
(EM_EPI_STORE, EM_EPI_SWISH, EM_EPI_RELU, EM_EPI_RESID_F32, EM_EPI_SCALE_F32, EM_EPI_GLU,
 EM_EPI_STORE_F32, _EM_EPI_UNUSED_7, _EM_EPI_UNUSED_8, EM_EPI_ARGMAX_PART, EM_EPI_GELU, EM_EPI_QK_HEADS,
 EM_EPI_VT_HEADS) = range(13)
EM_A_PLAIN, EM_A_CONV2 = 0, 1

EM_DW_SWISH, EM_DW_LINEAR, EM_DW_GATE, EM_DW_SELFRES = range(4)
EM_ENC_ISOLATE_UTTS = 1  # em_conformer_encode flags (include/espnet_amd.h)
EM_ENC_NO_FUSED = 2
EM_ENC_POS_PROJECTED = 4
EM_ENC_FOLD_C = 8
EM_ENC_SPLIT_ATT = 16
EM_ENC_POS_PACKED = 32


def EM_ENC_IN_FLIGHT(n):
    return (int(n) & 15) << 8


EM_ENC_PLAN_FUSED, EM_ENC_PLAN_CTC_IDS = 1, 2
EM_BLOCK_C, EM_BLOCK_D, EM_BLOCK_A, EM_BLOCK_FINAL, EM_BLOCK_CTC, EM_BLOCK_RELU, EM_BLOCK_ATT = 1, 2, 4, 8, 16, 32, 64
EM_BLOCK_PARAM_GROUP = 1792
EM_BLOCK_CTC_MAX_UNITS = 88  # vocabularies up to 5 632 labels take the fused CTC stage (the sizes the GPU tests cover); larger ones keep the arg-max GEMM
EM_PROF_GEMM, EM_PROF_BLOCK, EM_PROF_ATTN, EM_PROF_ROWS = 0, 1, 2, 3
DTYPES = {"float32": EM_F32, "fp32": EM_F32, "f32": EM_F32, "bfloat16": EM_BF16, "bf16": EM_BF16}


class TooShortUttError(Exception):
    """Same contract as the reference's TooShortUttError (subsampling.py:14-29)."""

    def __init__(self, message, actual_size, limit, indices=None):
        super().__init__(message)
        self.actual_size = actual_size
        self.limit = limit
        self.indices = indices  # batch rows that are too short (utterance-batched entry only)


class EspnetAmdError(RuntimeError):
    pass


class EmGemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int32), ("ldc", C.c_int32), ("scale", C.c_float),
                ("T1", C.c_int32), ("F1", C.c_int32), ("T2", C.c_int32), ("F2", C.c_int32),
                ("d", C.c_int32), ("conv_k", C.c_int32), ("conv_s", C.c_int32)]


_LAYER_PTRS = ["norm_ff_mac_g", "norm_ff_mac_b", "norm_mha_g", "norm_mha_b", "norm_conv_g",
               "norm_conv_b", "norm_ff_g", "norm_ff_b", "norm_final_g", "norm_final_b",
               "ffm_w1", "ffm_b1", "ffm_w2", "ffm_b2", "wqkv", "bqkv", "pos_u", "pos_v", "wout",
               "bout", "pw1", "pw1_b", "dw_w", "dw_b", "pw2", "pw2_b", "ff_w1", "ff_b1", "ff_w2",
               "ff_b2", "pw1f", "ffm_w2p", "ff_w2p", "woutp", "pw2p", "ff_w1p", "ffm_w1p", "wqkvp", "fp_c", "fp_da", "fp_a"]


class EmConformerLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _LAYER_PTRS]


class EmBlockArgs(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "T", "Tpad", "d", "ff", "kernel")] + [("eps", C.c_float)] + \
               [(n, C.c_void_p) for n in ("x", "ctx", "glu", "qh", "kh", "vt", "enc_out", "enc_act", "tlens", "wout",
                                          "pw1f", "pw2", "ff_w1", "ff_w2", "dw_w", "dw_b", "ffm_w1", "ffm_w2", "wqkv",
                                          "params", "ctc_w", "ctc_b", "ctc_ids")] + [("ctc_units", C.c_int32)] + \
               [("x_out", C.c_void_p), ("params_c", C.c_void_p), ("ffm_b1g", C.c_void_p), ("ff_b1g", C.c_void_p),
                ("row0_src", C.c_void_p), ("last_dst", C.c_void_p), ("row_stride", C.c_int32), ("ldp", C.c_int32),
                ("pos", C.c_void_p), ("pos_u", C.c_void_p), ("pos_v", C.c_void_p), ("klens", C.c_void_p), ("kv_frag", C.c_int32),
                ("ffn_split", C.c_int32), ("ffn_part", C.c_void_p), ("ffn_ticket", C.c_void_p), ("att_mask", C.c_int32)]


EM_ROWS_FFN, EM_ROWS_GLU = 0, 1


class EmFfnRowsArgs(C.Structure):
    """include/espnet_amd.h EmFfnRowsArgs (csrc/ffn_rows.hip)."""
    _fields_ = [(n, C.c_void_p) for n in ("xn_in", "x", "w1p", "w2p", "b1", "b2", "g1", "be1", "g2", "be2", "xn_out",
                                          "out_f32")] + \
               [(n, C.c_int32) for n in ("M", "d", "ff", "ln_mode")] + [("scale", C.c_float), ("eps", C.c_float)] + \
               [(n, C.c_void_p) for n in ("pre_in", "pre_w", "pre_b", "pre_g", "pre_be")] + [("main", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("post_w", "post_b", "post_ids")] + [("post_chunks", C.c_int32), ("post_vocab", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("post_q", "post_k", "post_vt")] + [("post_T", C.c_int32), ("post_Tpad", C.c_int32)]


class EmConformerWeights(C.Structure):
    _fields_ = [("d", C.c_int32), ("heads", C.c_int32), ("ff", C.c_int32),
                ("num_blocks", C.c_int32), ("kernel", C.c_int32), ("n_mels", C.c_int32),
                ("conv1_w", C.c_void_p), ("conv1_b", C.c_void_p), ("conv2_w", C.c_void_p),
                ("conv2_b", C.c_void_p), ("embed_w", C.c_void_p), ("embed_b", C.c_void_p),
                ("wpos_all", C.c_void_p), ("after_norm_g", C.c_void_p),
                ("after_norm_b", C.c_void_p), ("layers", C.POINTER(EmConformerLayer)),
                ("subsample", C.c_int32), ("conv3_w", C.c_void_p), ("conv3_b", C.c_void_p),
                ("legacy_relpos", C.c_int32), ("ctc_w", C.c_void_p), ("ctc_b", C.c_void_p),
                ("ctc_ids", C.c_void_p), ("ctc_units", C.c_int32), ("conv1_wf", C.c_void_p), ("conv2_wf", C.c_void_p)]


# order of include/espnet_amd.h EmEBranchformerLayer
_EBF_LAYER_PTRS = ["norm_ff_mac_g", "norm_ff_mac_b", "norm_mha_g", "norm_mha_b", "norm_mlp_g", "norm_mlp_b",
                   "norm_ff_g", "norm_ff_b", "norm_final_g", "norm_final_b", "ffm_w1", "ffm_w2", "ff_w1", "ff_w2",
                   "ffm_b1", "ffm_b2", "ff_b1", "ff_b2", "wqkv", "bqkv", "pos_u", "pos_v", "wout", "bout",
                   "proj1_w", "proj1_b", "csgu_norm_g", "csgu_norm_b", "csgu_conv_w", "csgu_conv_b", "proj2_w",
                   "proj2_b", "merge_conv_w", "merge_conv_b", "merge_w", "merge_b", "pool_w", "pool_b", "wproj_w",
                   "wproj_b", "ffm_w1p", "ffm_w2p", "ff_w1p", "ff_w2p", "wqkvp"]
EM_MERGE_CONCAT, EM_MERGE_LEARNED_AVE = 0, 1


class EmEBranchformerLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _EBF_LAYER_PTRS]


class EmEBranchformerWeights(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d", "heads", "ff", "cg", "num_blocks", "cg_kernel", "merge_kernel",
                                         "n_mels")] + \
               [(n, C.c_void_p) for n in ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "embed_w", "embed_b",
                                          "wpos_all", "after_norm_g", "after_norm_b")] + \
               [("layers", C.POINTER(EmEBranchformerLayer)), ("use_ffn", C.c_int32), ("merge_conv", C.c_int32),
                ("subsample", C.c_int32), ("conv3_w", C.c_void_p), ("conv3_b", C.c_void_p),
                ("legacy_relpos", C.c_int32), ("merge_method", C.c_int32), ("conv1_wf", C.c_void_p),
                ("conv2_wf", C.c_void_p)]


class EmWavInfo(C.Structure):
    _fields_ = [("frames", C.c_int64), ("data_offset", C.c_int64), ("rate", C.c_int32), ("channels", C.c_int32),
                ("bits", C.c_int32), ("format", C.c_int32), ("status", C.c_int32)]


_DEC_LAYER_PTRS = ["norm1_g", "norm1_b", "norm2_g", "norm2_b", "norm3_g", "norm3_b", "self_wqkv",
                   "self_bqkv", "self_wout", "self_bout", "src_wq", "src_bq", "src_wkv", "src_bkv",
                   "src_wout", "src_bout", "w1", "b1", "w2", "b2", "w1_frag", "w2_frag", "self_wqkv_frag",
                   "self_wout_frag", "src_wout_frag", "src_wq_frag"]


def pack_frag16(w, pad_rows=16):
    """[R][K] -> fragment-major [R/16][K/32][lane = 16 * (k % 32 // 8) + r % 16][k % 8] (csrc/dec_ffn.hip: the 16 rows x
    32 k of one MFMA operand contiguous, so that a wave-wide 16-byte-per-lane load reads 1 KiB in one piece).  Rows are
    zero-padded to a multiple of `pad_rows` (em_ln_gemm_frag wants 512: whole workgroup column slices)."""
    import torch

    R, K = w.shape
    assert pad_rows % 16 == 0 and K % 32 == 0, (R, K, pad_rows)
    Rp = (R + pad_rows - 1) // pad_rows * pad_rows
    if Rp != R:
        w = torch.cat([w, w.new_zeros(Rp - R, K)], 0)
    return w.reshape(Rp // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()


class EmDecoderLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _DEC_LAYER_PTRS]


class EmDecoderWeights(C.Structure):
    _fields_ = [("d", C.c_int32), ("heads", C.c_int32), ("ff", C.c_int32),
                ("num_blocks", C.c_int32), ("vocab", C.c_int32), ("pe_len", C.c_int32),
                ("embed", C.c_void_p), ("pe", C.c_void_p), ("after_norm_g", C.c_void_p),
                ("after_norm_b", C.c_void_p), ("out_w", C.c_void_p), ("out_b", C.c_void_p),
                ("layers", C.POINTER(EmDecoderLayer)), ("out_w_frag", C.c_void_p)]


class EmDecoderStepArgs(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "W", "T", "Tpad", "Lmax", "pos")] + \
               [(n, C.c_void_p) for n in ("tok", "anc", "xlens", "self_k", "self_v", "mem_kv", "mem_vT", "x",
                                          "xn", "qkv", "qs", "ctx", "hbuf", "logits")]


class EmSearchParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "W", "V", "T", "Tpad", "S", "NC", "Lmax", "end_cap",
                                         "sos", "eos", "blank", "use_end_detect")] + \
               [(n, C.c_float) for n in ("w_dec", "w_ctc", "w_len", "w_lm")] + [("ldT", C.c_int32)]


_LM_LAYER_PTRS = ["norm1_g", "norm1_b", "norm2_g", "norm2_b", "wqkv", "bqkv", "wout", "bout", "w1", "b1",
                  "w2", "b2"]


class EmLmLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _LM_LAYER_PTRS]


class EmRnnLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_ih", "w_hh", "bias")]


EM_LM_TRANSFORMER, EM_LM_LSTM, EM_LM_GRU, EM_LM_RNN_TANH, EM_LM_RNN_RELU = 0, 1, 2, 3, 4


class EmLmWeights(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d", "heads", "ff", "num_blocks", "vocab", "embed_unit")] + \
               [(n, C.c_void_p) for n in ("embed", "in_w", "in_b", "in_ln_g", "in_ln_b", "pe",
                                          "after_norm_g", "after_norm_b", "out_w", "out_b")] + \
               [("layers", C.POINTER(EmLmLayer)), ("kind", C.c_int32), ("nhid", C.c_int32),
                ("rnn", C.POINTER(EmRnnLayer))]


SEARCH_BUFFERS = ["xlens", "maxlens", "minlens", "ctc_lpT", "tok", "parent", "anc_a", "anc_b",
                  "alive", "run_score", "run_sdec", "run_sctc", "run_slen", "s_prev", "r_a", "r_b",
                  "cand_tok", "cand_full", "cand_psi", "cand_total", "sel_idx", "sel_total",
                  "end_count", "end_pos", "end_slot", "end_forced", "end_score", "end_sdec",
                  "end_sctc", "end_slen", "best_all", "best_by_len", "done", "step", "x", "xn", "qkv", "qs",
                  "ctx", "hbuf", "dec_logp", "self_k", "self_v", "mem_kv", "mem_vT",
                  "lm", "lm_e", "lm_xn", "lm_qkv", "lm_ctx", "lm_h", "lm_x", "lm_logp", "lm_k", "lm_v",
                  "run_slm", "end_slm", "rnn_hs", "rnn_cs", "rnn_hin", "rnn_gates",
                  "online_best", "online_psi", "online_snap", "mem_kf", "mem_vf"]


class EmSearchBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in SEARCH_BUFFERS]


_i32, _f32, _vp, _sz = C.c_int32, C.c_float, C.c_void_p, C.c_size_t
_SIGNATURES = {
    "em_version": (C.c_int, []),
    "em_error_string": (C.c_char_p, [C.c_int]),
    "em_frontend_logmel_f32": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp,
                                         _i32, _vp, _vp]),
    "em_utt_mvn_partial_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "em_utt_mvn_apply_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "em_global_mvn_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "em_conv2d_sub1": (C.c_int, [C.c_int, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "em_conv2d_sub12_bf16": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp]),
    "em_gemm": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(EmGemmArgs), _vp]),
    "em_layernorm": (C.c_int, [C.c_int, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp]),
    "em_layernorm2": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp]),
    "em_layernorm_inplace_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _f32, _vp]),
    "em_relpos_attention": (C.c_int, [C.c_int, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32,
                                      _i32, _vp, _vp]),
    "em_legacy_relpos_attention": (C.c_int, [C.c_int, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32,
                                      _i32, _vp, _vp]),
    "em_dwconv_bn_swish": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "em_argmax_rows_f32": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "em_argmax_partials": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "em_log_softmax_rows_f32": (C.c_int, [_vp, _i32, _i32, _vp]),
    "em_ctc_collapse": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "em_conformer_workspace_bytes": (_sz, [C.c_int, C.POINTER(EmConformerWeights), _i32, _i32]),
    "em_ebranchformer_workspace_bytes": (_sz, [C.c_int, C.POINTER(EmEBranchformerWeights), _i32, _i32]),
    "em_ebranchformer_encode": (C.c_int, [C.c_int, C.POINTER(EmEBranchformerWeights), _vp, _vp, _vp, _vp,
                                          _i32, _i32, _vp, _vp, _sz, _vp, _vp, _i32, _vp]),
    "em_wav_probe": (C.c_int, [C.POINTER(C.c_char_p), _i32, C.POINTER(EmWavInfo), _i32]),
    "em_wav_load_rows": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(EmWavInfo), _i32, _vp, C.c_int64, _i32]),
    "em_branch_learned_ave": (C.c_int, [C.c_int, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "em_dec_ffn": (C.c_int, [C.c_int, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "em_dec_ffn_split": (C.c_int, [_i32, _i32, _i32]),
    "em_ln_gemm_frag": (C.c_int, [C.c_int, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "em_ln_gemm": (C.c_int, [C.c_int, C.c_int, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "em_dwconv": (C.c_int, [C.c_int, C.c_int, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp,
                            _i32, _vp]),
    "em_row_stats": (C.c_int, [C.c_int, _vp, _i32, _i32, _i32, _f32, _vp, _vp]),
    "em_dwconv_ln_gate": (C.c_int, [C.c_int, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32,
                                    _vp, _i32, _vp]),
    "em_layernorm_act": (C.c_int, [C.c_int, _vp, _i32, _vp, _vp, _i32, _i32, _f32, _vp, _i32, _vp]),
    "em_conformer_encode": (C.c_int, [C.c_int, C.POINTER(EmConformerWeights), _vp, _vp, _vp, _vp,
                                      _i32, _i32, _vp, _vp, _sz, _vp, _vp, _i32, _vp]),
    "em_stream_pos_enc_f32": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "em_cb_build_blocks_f32": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp,
                                         _vp]),
    "em_block_mha": (C.c_int, [C.c_int, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "em_cb_propagate_ctx_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "em_cb_workspace_bytes": (_sz, [C.c_int, C.POINTER(EmConformerWeights), _i32, _i32]),
    "em_cb_encode_blocks": (C.c_int, [C.c_int, C.POINTER(EmConformerWeights), _vp, _i32, _i32, _i32,
                                      _vp, _vp, _vp, _sz, _vp]),
    "em_cb_build_blocks_batch_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "em_cb_build_blocks_rows_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "em_cb_encode_blocks_batch": (C.c_int, [C.c_int, C.POINTER(EmConformerWeights), _vp, _i32, _i32, _i32, _i32,
                                            _vp, _vp, _vp, _sz, _vp]),
    "em_profile_create": (_vp, [_i32]),
    "em_profile_destroy": (None, [_vp]),
    "em_profile_attach": (None, [_vp]),
    "em_profile_read": (C.c_int, [_vp, _vp, _vp, _i32, _vp]),
    "em_profile_read2": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp]),
    "em_conformer_block_fused": (C.c_int, [C.c_int, C.POINTER(EmBlockArgs), _vp]),
    "em_dev_switches_reload": (None, []),
    "em_relpos_pos_fragments": (C.c_int, [C.c_int32]),
    "em_relpos_pack_pos_bf16": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    "em_ffn_rows_fused": (C.c_int, [C.POINTER(EmFfnRowsArgs), _vp]),
    "em_relpos_attention2_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp,
                                            _vp]),
    "em_cast_f32": (C.c_int, [C.c_int, _vp, _sz, _vp, _vp]),
    "em_dec_embed_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp]),
    "em_dec_self_attention": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                        _i32, _vp, _i32, _vp, _vp, _vp]),
    "em_dec_self_attention_beam": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                             _i32, _vp, _i32, _vp, _vp]),
    "em_lm_embed": (C.c_int, [C.c_int, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp]),
    "em_lm_input_norm_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp]),
    "em_dec_src_attention": (C.c_int, [C.c_int, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32,
                                       _i32, _i32, _vp, _vp]),
    "em_dec_src_attention_lnq": (C.c_int, [C.c_int, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32,
                                          _i32, _i32, _i32, _vp, _vp]),
    "em_dec_src_attention_lnq_frag": (C.c_int, [C.c_int, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32,
                                          _i32, _i32, _i32, _vp, _vp]),
    "em_dec_transpose_v": (C.c_int, [C.c_int, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "em_search_init": (C.c_int, [C.c_int, C.POINTER(EmSearchParams), C.POINTER(EmDecoderWeights),
                                 C.POINTER(EmSearchBuffers), _vp, _i32, _vp, _vp, _vp]),
    "em_search_steps": (C.c_int, [C.c_int, C.POINTER(EmSearchParams), C.POINTER(EmDecoderWeights),
                                  C.POINTER(EmSearchBuffers), _i32, _i32, _vp]),
    "em_search_online_extend": (C.c_int, [C.c_int, C.POINTER(EmSearchParams), C.POINTER(EmDecoderWeights),
                                          C.POINTER(EmSearchBuffers), _vp, _i32, _vp, _vp, _i32, _i32, _vp]),
    "em_search_online_core": (C.c_int, [C.c_int, C.POINTER(EmSearchParams), C.POINTER(EmDecoderWeights),
                                        C.POINTER(EmSearchBuffers), _i32, _vp]),
    "em_search_online_commit": (C.c_int, [C.c_int, C.POINTER(EmSearchParams), C.POINTER(EmSearchBuffers),
                                          _i32, _vp]),
    "em_search_online_rewind": (C.c_int, [C.POINTER(EmSearchParams), C.POINTER(EmSearchBuffers), _vp]),
    "em_decoder_memory": (C.c_int, [C.c_int, C.POINTER(EmDecoderWeights), _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "em_decoder_step": (C.c_int, [C.c_int, C.POINTER(EmDecoderWeights), C.POINTER(EmDecoderStepArgs), _vp]),
    "em_lm_step": (C.c_int, [C.c_int, C.POINTER(EmSearchParams), C.POINTER(EmSearchBuffers), _i32, _vp]),
    "em_ctc_log_probs_t": (C.c_int, [C.c_int, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "em_ctc_prefix_init": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "em_ctc_prefix_score": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                      _i32, _vp, _vp, _vp]),
    "em_ctc_prefix_state": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp,
                                      _vp]),
    "em_conformer_encode_plan": (C.c_int, [C.c_int, C.POINTER(EmConformerWeights), _i32]),
    "em_conformer_encode_plan_for": (C.c_int, [C.c_int, C.POINTER(EmConformerWeights), _i32, _i32, _i32]),
    "em_ctc_prefix_extend": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "em_ctc_greedy": (C.c_int, [C.c_int, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32,
                                _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def library_path() -> Path:
    return Path(os.environ.get("ESPNET_AMD_LIB", str(_LIB_PATH)))


def exported_symbols():
    return sorted(_SIGNATURES)


def load():
    """Load the shared library (once) and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own libamdhip64; it must be the HIP runtime this library binds to (streams
    # and device pointers come from torch).  Importing torch first makes the dynamic linker resolve
    # our DT_NEEDED libamdhip64 to the copy torch already loaded instead of a second runtime.
    import torch  # noqa: F401

    path = library_path()
    if not path.exists():
        raise EspnetAmdError(
            f"{path} not found: the HIP extension is not built (run `python -m espnet_amd.build`). "
            "espnet_amd has no CPU fallback.")
    lib = C.CDLL(str(path))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc == EM_OK:
        return
    msg = load().em_error_string(rc).decode()
    if rc == EM_ERR_UNSUPPORTED:
        raise NotImplementedError(f"{what}: {msg}")
    if rc == EM_ERR_BAD_ARG:
        raise ValueError(f"{what}: {msg}")
    raise EspnetAmdError(f"{what}: {msg} (code {rc})")


def ptr(t):
    """Raw device/host pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(t, name="tensor"):
    if not t.is_cuda:
        raise EspnetAmdError(
            f"{name} lives on {t.device}: espnet_amd runs only on an MI355X (HIP) device and has "
            "no CPU fallback")
