// Developer switches of the library in ONE place (VERDICT r05 item 9 / ADVICE r05): the environment is read ONCE, into
// this struct, the first time a launcher asks - no getenv() on a call path (the label step used to read one per layer and
// step, which also raced a concurrent setenv from a Python thread).  They select between device code paths for A/B timing
// and diagnostics; none of them selects a CPU path.  em_dev_switches_reload() (C-ABI) reads the environment again: the
// tests that flip a switch in-process call it after changing os.environ.  DESIGN.md lists what each one does.
#pragma once

struct EmSwitches {
  bool attn2_stamps, block_stamps, ffn_stamps, sub2_stamps;          // EM_*_STAMPS: print s_memtime stamps of one workgroup
  bool block_no_helpers;                                             // ESPNET_AMD_BLOCK_NO_HELPERS
  int sa_split, sa_group;                                            // ESPNET_AMD_SA_SPLIT / _SA_GROUP (0: automatic)
  bool no_sa_tree;                                                   // ESPNET_AMD_NO_SA_TREE
  int sa_tree_min_rows;                                              // ESPNET_AMD_SA_TREE_MIN_ROWS (default 200)
  bool no_rows_qkv;                                                  // ESPNET_AMD_NO_ROWS_QKV (512-wide: q | k | v as two GEMM launches instead of the walk behind the macaron launch)
  bool no_attn2_large, no_ffn_rows, no_rows_ctc;                     // ESPNET_AMD_NO_ATTN2_LARGE / _NO_FFN_ROWS / _NO_ROWS_CTC
  bool frontend_v1;                                                  // ESPNET_AMD_FRONTEND_V1
  int gemm_stages, gemm_bm;                                          // ESPNET_AMD_GEMM_STAGES / _GEMM_BM (0: automatic; 64 | 128: the M tile)
  bool no_mid_gemm;                                                  // ESPNET_AMD_NO_MID_GEMM
  int mid_tile, lng_rt, lng_wide;                                    // ESPNET_AMD_MID_TILE / _LNG_RT (0: automatic) / _LNG_WIDE (256)
  bool no_src_lnq, no_tail_fusion;                                   // ESPNET_AMD_NO_SRC_LNQ / _NO_TAIL_FUSION
  bool stream_no_fused, stream_mha_v1, stream_no_ctx_fold, stream_no_ln_gemm;  // ESPNET_AMD_STREAM_*
  int stream_fused_min;                                              // ESPNET_AMD_STREAM_FUSED_MIN (default 1; 8 in rounds 4 - 5)
  bool stream_split_att;                                             // ESPNET_AMD_STREAM_SPLIT_ATT
  int stream_ffn_split;                                              // ESPNET_AMD_STREAM_FFN_SPLIT (0: automatic; 1: off; n: forced)
  bool no_sub12;                                                     // ESPNET_AMD_NO_SUB12
  int ffn_rows_min_fill;                                             // ESPNET_AMD_FFN_ROWS_MIN_FILL (percent of the CUs a round of 64-row workgroups must fill; 0 = automatic: 72 / batches in flight)
  int dec_ffn_rows;                                                  // ESPNET_AMD_DEC_FFN_ROWS (16 | 32 rows per workgroup of ln_frag_gemm_kernel; 0: automatic, 32 from 320 rows)
  int dec_ffn_split;                                                 // ESPNET_AMD_DEC_FFN_SPLIT (0: automatic; 1: off - LayerNorm + two projections; n: forced)
};

const EmSwitches& em_sw();  // csrc/api.hip
