"""CPU-only tests of the host layer: the C ABI library loads and exports every declared symbol
(no compute calls without a GPU), registry/config handling, state-dict compatibility, lengths."""
import re
from pathlib import Path

import pytest
import torch

from espnet_amd import lib as L
from tests.helpers import golden_state_dict, load_golden

REPO = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    hdr = (REPO / "include" / "espnet_amd.h").read_text()
    declared = set(re.findall(r"\b(em_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes found"
    lib = L.load()  # raises if not built
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/espnet_amd.h but not exported"
    assert declared == set(L.exported_symbols()), declared ^ set(L.exported_symbols())
    assert lib.em_version() >= 1
    assert b"short" in lib.em_error_string(L.EM_ERR_TOO_SHORT)


def test_no_cpu_fallback():
    from espnet_amd.tasks.asr import ASRTask

    g = load_golden("tiny_blocks")
    model = ASRTask.build_model(g["config"])
    with pytest.raises(L.EspnetAmdError):
        model.encode(torch.zeros(1, 16000), torch.tensor([16000]))


@pytest.mark.parametrize("name", ["tiny_blocks", "small_10s", "large_10s", "tiny_beam5", "ebf_tiny_blocks",
                                  "ebf_small_5s", "bf_tiny_blocks", "bf_small_4s", "stream_search_a",
                                  "bf_learned_ave_4s", "bf_fixed_ave_4s"])
def test_state_dict_table_equals_reference(name):
    from espnet_amd.tasks.asr import ASRTask

    g = load_golden(name)
    cfg = g["config"]
    if cfg["encoder_conf"]["output_size"] // cfg["encoder_conf"]["attention_heads"] != 64:
        with pytest.raises(NotImplementedError):
            ASRTask.build_model(cfg)
        return
    model = ASRTask.build_model(cfg)
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert mine == g["shapes"]
    model.load_state_dict(golden_state_dict(g), strict=True)
    assert model.sos == model.eos == int(g["vocab"]) - 1 and model.blank_id == 0


@pytest.mark.parametrize("name", ["tiny_beam5_lm", "tiny_beam4_lm_posenc", "tiny_beam5_rnnlm", "tiny_beam4_rnnlm_nhid", "tiny_beam5_gru", "tiny_beam4_gru_nhid",
                                  "tiny_beam4_rnn_tanh", "tiny_beam4_rnn_relu"])
def test_lm_state_dict_table_equals_reference(name):
    """LM scorers expose the reference's own state-dict keys/shapes (espnet2/lm/{transformer_lm,seq_rnn_lm}.py),
    through LMTask's registry names (espnet2/tasks/lm.py:36-44)."""
    import json

    from espnet_amd.lm.transformer_lm import ESPnetLanguageModel
    from espnet_amd.tasks.lm import LMTask
    from oracle.weights import token_list

    g = load_golden(name)
    V = int(g["vocab"])
    lm_name = str(g["lm_name"]) if "lm_name" in g else "transformer"
    model = LMTask.build_model(dict(lm=lm_name, lm_conf=json.loads(str(g["lm_conf"])), token_list=token_list(V)))
    assert isinstance(model, ESPnetLanguageModel)
    ref = {"lm." + k: tuple(v) for k, v in json.loads(str(g["lm_state_shapes"])).items()}
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == ref


def test_unsupported_choices_raise():
    from espnet_amd.tasks.asr import ASRTask

    g = load_golden("tiny_blocks")
    cfg = dict(g["config"])
    cfg["encoder"] = "transformer"
    with pytest.raises(NotImplementedError):
        ASRTask.build_model(cfg)
    cfg = dict(g["config"])
    cfg["encoder_conf"] = dict(cfg["encoder_conf"], zero_triu=True)
    with pytest.raises(NotImplementedError):
        ASRTask.build_model(cfg)
    cfg = dict(g["config"])
    cfg["encoder_conf"] = dict(cfg["encoder_conf"], pos_enc_layer_type="abs_pos", selfattention_layer_type="selfattn")
    with pytest.raises(NotImplementedError):
        ASRTask.build_model(cfg)
    cfg = dict(g["config"])  # rel_pos_type legacy IS on the fast path (LegacyRelPositionMultiHeadedAttention)
    cfg["encoder_conf"] = dict(cfg["encoder_conf"], rel_pos_type="legacy", output_size=128, attention_heads=2)
    assert ASRTask.build_model(cfg).encoder.legacy_relpos


def test_lengths_match_oracle():
    import random

    from espnet_amd.nets_utils import conv2d_subsampled_lengths, stft_frame_lengths
    from oracle.conformer import subsampled_lengths

    assert stft_frame_lengths([30, 15], 4, 2) == [16, 8]  # test/espnet2/layers/test_stft.py:10-16
    rng = random.Random(0)
    for _ in range(300):
        tmax = rng.randint(7, 400)
        fl = [rng.randint(1, tmax) for _ in range(4)] + [tmax]
        assert conv2d_subsampled_lengths(fl, tmax) == subsampled_lengths(torch.tensor(fl), tmax).tolist()


def test_mel_band_packing_is_exact():
    from espnet_amd.layers.log_mel import mel_filterbank, pack_banded

    g = load_golden("small_10s")
    m = mel_filterbank(16000, 512, 80, 0, 8000)
    assert (m == g["melmat"]).all()
    packed, lo, maxlen = pack_banded(torch.from_numpy(m))
    dense = torch.zeros(257, 80)
    for j in range(80):
        for s in range(maxlen):
            if lo[j] + s < 257:
                dense[lo[j] + s, j] += packed[s, j]
    assert torch.equal(dense, torch.from_numpy(m))


def test_fragment_major_weight_units_follow_the_header():
    """pack_k_units / pack_w1 / pack_w2 against the layouts include/espnet_amd.h states at EmBlockArgs.  K unit: lane
    16 lg + lr of wave nf finds the MFMA operand (W row 16 nf + lr of the 64-row group, k = 32 ks + 8 lg .. + 7) at
    byte 8192 nf + 1024 ks + 16 lane of the unit.  FFN second matrix: per pair of hidden chunks and wave, 16
    output-fragment lines whose eight elements per lane are the wave's OWN hidden columns of the two chunks."""
    from espnet_amd.asr.encoder.conformer_encoder import pack_k_units, pack_w1, pack_w2

    w = torch.arange(192 * 256, dtype=torch.float32).reshape(192, 256)
    flat = pack_k_units(w).reshape(-1)
    for u in range(3):
        for nf in range(4):
            for ks in range(8):
                for lane in (0, 5, 16, 37, 63):
                    lg, lr = lane >> 4, lane & 15
                    off = u * 16384 + nf * 4096 + ks * 512 + lane * 8  # in elements: 32 KiB unit = 16 384 bf16
                    want = w[64 * u + 16 * nf + lr, 32 * ks + 8 * lg: 32 * ks + 8 * lg + 8]
                    assert torch.equal(flat[off:off + 8], want), (u, nf, ks, lane)
    w2 = torch.arange(256 * 256, dtype=torch.float32).reshape(256, 256) + 1.0
    flat = pack_w2(w2).reshape(-1)
    for p in range(2):
        for wv in range(4):
            for f in (0, 3, 8, 15):
                for lane in (0, 9, 31, 48, 63):
                    lg, lr = lane >> 4, lane & 15
                    off = p * 32768 + wv * 8192 + f * 512 + lane * 8  # elements: 64 KiB pair = 32 768 bf16
                    cols = [64 * (2 * p + (e >> 2)) + 16 * wv + 4 * lg + (e & 3) for e in range(8)]
                    assert torch.equal(flat[off:off + 8], w2[16 * f + lr, cols]), (p, wv, f, lane)
    # a permutation: nothing lost, nothing duplicated
    assert torch.equal(pack_k_units(w).reshape(-1).sort().values, w.reshape(-1))
    assert torch.equal(pack_w2(w2).reshape(-1).sort().values, w2.reshape(-1).sort().values)
    # an odd chunk count is padded with one zero chunk on both matrices
    w1o, w2o = torch.ones(192, 256), torch.ones(256, 192)
    assert pack_w1(w1o).shape == (256, 256) and float(pack_w1(w1o).sum()) == 192 * 256
    assert pack_w2(w2o).shape == (2, 32768) and float(pack_w2(w2o).sum()) == 192 * 256
    with pytest.raises(AssertionError):
        pack_k_units(torch.zeros(60, 256))


def test_row_block_operand_streams_follow_the_header():
    """pack_ffn_rows_w1 / _w2, pack_rows_proj, pack_rows_glu (the 512-wide model's row-block launches, csrc/ffn_rows.hip)
    against the layouts include/espnet_amd.h states at EmFfnRowsArgs: wave w's fragment i of a 128-row chunk is the 1 KiB at
    chunk + 8 KiB * i + 1 KiB * w, lane 16 lg + lr holding 8 consecutive k (W1, the projection) or its own hidden columns (W2)."""
    from espnet_amd.asr.encoder.conformer_encoder import (glu_chunk_order, pack_ffn_rows_w1, pack_ffn_rows_w2,
                                                          pack_rows_glu, pack_rows_proj)

    ff = 256
    w1 = torch.arange(ff * 512, dtype=torch.float32).reshape(ff, 512)
    flat = pack_ffn_rows_w1(w1).reshape(-1)
    for c in range(ff // 128):
        for ks in (0, 7, 15):
            for wv in (0, 3, 7):
                for lane in (0, 5, 16, 37, 63):
                    lg, lr = lane >> 4, lane & 15
                    off = c * 65536 + ks * 4096 + wv * 512 + lane * 8  # elements: 128 KiB chunk = 65 536 bf16
                    want = w1[128 * c + 16 * wv + lr, 32 * ks + 8 * lg: 32 * ks + 8 * lg + 8]
                    assert torch.equal(flat[off:off + 8], want), (c, ks, wv, lane)
    w2 = torch.arange(512 * ff, dtype=torch.float32).reshape(512, ff) + 1.0
    flat = pack_ffn_rows_w2(w2).reshape(-1)
    for c in range(ff // 128):
        for s in range(4):
            for cf in range(4):
                for wv in (0, 2, 7):
                    for lane in (0, 9, 31, 48, 63):
                        lg, lr = lane >> 4, lane & 15
                        off = c * 65536 + (s * 4 + cf) * 4096 + wv * 512 + lane * 8
                        cols = [128 * c + 32 * s + 16 * (e >> 2) + 4 * lg + (e & 3) for e in range(8)]
                        assert torch.equal(flat[off:off + 8], w2[64 * wv + 16 * cf + lr, cols]), (c, s, cf, wv, lane)
    wp = torch.arange(512 * 512, dtype=torch.float32).reshape(512, 512)
    flat = pack_rows_proj(wp).reshape(-1)
    for ks in (0, 9, 15):
        for cf in range(4):
            for wv in (0, 5, 7):
                for lane in (0, 21, 63):
                    lg, lr = lane >> 4, lane & 15
                    off = (ks * 4 + cf) * 4096 + wv * 512 + lane * 8
                    assert torch.equal(flat[off:off + 8], wp[64 * wv + 16 * cf + lr, 32 * ks + 8 * lg: 32 * ks + 8 * lg + 8])
    # GLU: chunk 2 j = the value rows of output columns 128 j .., chunk 2 j + 1 their gate rows (F.glu: value | gate halves)
    order = glu_chunk_order(512)
    assert order[:128].tolist() == list(range(128)) and order[128:256].tolist() == list(range(512, 640))
    assert order[256:384].tolist() == list(range(128, 256)) and sorted(order.tolist()) == list(range(1024))
    pw1 = torch.arange(1024 * 512, dtype=torch.float32).reshape(1024, 512)
    assert torch.equal(pack_rows_glu(pw1), pack_ffn_rows_w1(pw1[order]))
    for f, wt in ((pack_ffn_rows_w1, w1), (pack_ffn_rows_w2, w2), (pack_rows_proj, wp)):  # permutations
        assert torch.equal(f(wt).reshape(-1).sort().values, wt.reshape(-1).sort().values)


def test_rel_pos_table_matches_oracle():
    from espnet_amd.asr.encoder.conformer_encoder import rel_pos_table
    from oracle.conformer import rel_pos_emb

    for T, d in [(24, 64), (249, 256)]:
        assert torch.equal(rel_pos_table(T, d), rel_pos_emb(T, d))


@pytest.mark.parametrize("layer", ["conv2d", "conv2d6", "conv2d8"])
def test_subsampled_lengths_match_mask_slicing(layer):
    """Host length formulas of the three Conv2dSubsampling variants == counting the reference's mask slices
    (subsampling.py:448-449, :758, :851) on the padded mask, for every (length, padded length) pair."""
    from espnet_amd.nets_utils import conv2d_subsampled_lengths, conv_out_size
    from oracle import conformer as oc

    for tmax in (15, 16, 17, 40, 101, 250):
        lens = list(range(1, tmax + 1))
        want = oc.subsampled_lengths(torch.tensor(lens), tmax, layer).tolist()
        assert conv2d_subsampled_lengths(lens, tmax, layer) == want
        # an unpadded utterance keeps exactly the conv stack's output frames
        assert conv2d_subsampled_lengths([tmax], tmax, layer)[0] == conv_out_size(tmax, layer)


def test_collect_backtraces_token_tree_on_host():
    """BatchBeamSearch._collect: the ended list (node position / slot, forced-<eos> flag, scores) is turned
    into Hypothesis objects by walking `parent` through the token tree — host-only logic, checked on a hand
    built tree: two utterances, shared prefixes, a forced <eos>, best-first ordering."""
    from espnet_amd.nets.batch_beam_search import BatchBeamSearch
    from espnet_amd.nets.scorers.ctc import CTCPrefixScorer

    W, B, Lmax, cap, EOS = 2, 2, 6, 4, 9
    n = B * W
    bs = BatchBeamSearch(beam_size=W, weights=dict(ctc=1.0), scorers=dict(ctc=CTCPrefixScorer(ctc=None, eos=EOS)),
                         sos=EOS, eos=EOS, vocab_size=10, token_list=[str(i) for i in range(10)])
    tok = torch.full((Lmax, n), -7, dtype=torch.int32)
    par = torch.full((Lmax, n), -1, dtype=torch.int32)
    tok[0] = EOS  # <sos> == <eos> id
    # utterance 0 (rows 0,1): sos-3-4-eos (ends at pos 3, slot 0) and sos-3-5 forced eos at pos 2, slot 1
    tok[1, 0], par[1, 0] = 3, 0
    tok[1, 1], par[1, 1] = 3, 0
    tok[2, 0], par[2, 0] = 4, 0
    tok[2, 1], par[2, 1] = 5, 1
    tok[3, 0], par[3, 0] = EOS, 0
    # utterance 1 (rows 2,3): sos-6-eos at pos 2, slot 3 (parent row 2)
    tok[1, 2], par[1, 2] = 6, 2
    tok[2, 3], par[2, 3] = EOS, 2
    z = lambda *s: torch.zeros(*s)
    bufs = dict(tok=tok, parent=par, end_count=torch.tensor([2, 1], dtype=torch.int32),
                end_pos=torch.tensor([[2, 3, 0, 0], [2, 0, 0, 0]], dtype=torch.int32),
                end_slot=torch.tensor([[1, 0, 0, 0], [3, 0, 0, 0]], dtype=torch.int32),
                end_forced=torch.tensor([[1, 0, 0, 0], [0, 0, 0, 0]], dtype=torch.int32),
                end_score=torch.tensor([[-5.0, -2.0, 0, 0], [-1.5, 0, 0, 0]]), end_sdec=z(B, cap),
                end_sctc=torch.tensor([[-5.0, -2.0, 0, 0], [-1.5, 0, 0, 0]]), end_slen=z(B, cap))
    out = bs._collect(bufs, B, W, [3, 3])
    assert [h.yseq.tolist() for h in out[0]] == [[EOS, 3, 4, EOS], [EOS, 3, 5, EOS]]  # best (-2.0) first
    assert [float(h.score) for h in out[0]] == [-2.0, -5.0]
    assert float(out[0][0].scores["ctc"]) == -2.0 and set(out[0][0].scores) == {"ctc"}
    assert [h.yseq.tolist() for h in out[1]] == [[EOS, 6, EOS]]
    bs.normalize_length = True  # beam_search.py:453-459: score / (len - 1)
    out = bs._collect(bufs, B, W, [3, 3])
    assert [float(h.score) for h in out[0]] == [-2.0, -5.0]  # -2/3 > -5/3


# reference callable (tests/golden/api_signatures.json, dumped from /root/reference by make_golden.py api_signatures)
# -> (espnet_amd module, attribute)
_API_MAP = {
    "espnet2.bin.asr_inference:Speech2Text.__init__": ("espnet_amd.bin.asr_inference", "Speech2Text.__init__"),
    "espnet2.bin.asr_inference:Speech2Text.__call__": ("espnet_amd.bin.asr_inference", "Speech2Text.__call__"),
    "espnet2.bin.asr_inference:inference": ("espnet_amd.bin.asr_inference", "inference"),
    "espnet2.bin.asr_inference_streaming:Speech2TextStreaming.__init__":
        ("espnet_amd.bin.asr_inference_streaming", "Speech2TextStreaming.__init__"),
    "espnet2.bin.asr_inference_streaming:Speech2TextStreaming.__call__":
        ("espnet_amd.bin.asr_inference_streaming", "Speech2TextStreaming.__call__"),
    "espnet2.bin.asr_inference_streaming:inference": ("espnet_amd.bin.asr_inference_streaming", "inference"),
    "espnet2.asr.espnet_model:ESPnetASRModel.__init__": ("espnet_amd.asr.espnet_model", "ESPnetASRModel.__init__"),
    "espnet2.asr.espnet_model:ESPnetASRModel.encode": ("espnet_amd.asr.espnet_model", "ESPnetASRModel.encode"),
    "espnet2.asr.frontend.default:DefaultFrontend.__init__": ("espnet_amd.asr.frontend.default", "DefaultFrontend.__init__"),
    "espnet2.asr.encoder.conformer_encoder:ConformerEncoder.__init__":
        ("espnet_amd.asr.encoder.conformer_encoder", "ConformerEncoder.__init__"),
    "espnet2.asr.encoder.e_branchformer_encoder:EBranchformerEncoder.__init__":
        ("espnet_amd.asr.encoder.e_branchformer_encoder", "EBranchformerEncoder.__init__"),
    "espnet2.asr.encoder.branchformer_encoder:BranchformerEncoder.__init__":
        ("espnet_amd.asr.encoder.e_branchformer_encoder", "BranchformerEncoder.__init__"),
    "espnet2.asr.encoder.contextual_block_conformer_encoder:ContextualBlockConformerEncoder.__init__":
        ("espnet_amd.asr.encoder.contextual_block_conformer_encoder", "ContextualBlockConformerEncoder.__init__"),
    "espnet2.asr.decoder.transformer_decoder:TransformerDecoder.__init__":
        ("espnet_amd.asr.decoder.transformer_decoder", "TransformerDecoder.__init__"),
    "espnet2.asr.ctc:CTC.__init__": ("espnet_amd.asr.ctc", "CTC.__init__"),
    "espnet2.lm.transformer_lm:TransformerLM.__init__": ("espnet_amd.lm.transformer_lm", "TransformerLM.__init__"),
    "espnet2.lm.seq_rnn_lm:SequentialRNNLM.__init__": ("espnet_amd.lm.seq_rnn_lm", "SequentialRNNLM.__init__"),
    "espnet2.legacy.nets.beam_search:BeamSearch.__init__": ("espnet_amd.nets.batch_beam_search", "BatchBeamSearch.__init__"),
    "espnet2.legacy.nets.batch_beam_search_online:BatchBeamSearchOnline.__init__":
        ("espnet_amd.nets.batch_beam_search_online", "BatchBeamSearchOnline.__init__"),
}
# the only deliberate default differences: there is no CPU path
_API_DEFAULT_EXCEPTIONS = {("Speech2Text.__init__", "device"), ("Speech2TextStreaming.__init__", "device")}


def test_public_signatures_cover_the_reference():
    """Drop-in at the API level: every parameter of the reference callables on the path is accepted by its
    espnet_amd counterpart under the same name with the same default (keyword-only extras such as
    `compute_dtype` are allowed; options of other decoding modes may be taken by `**kwargs`, which the
    constructors check against the reference defaults)."""
    import importlib
    import inspect
    import json

    ref = json.loads((REPO / "tests" / "golden" / "api_signatures.json").read_text())
    assert set(ref) == set(_API_MAP)
    for key, params in ref.items():
        mod, attr = _API_MAP[key]
        obj = importlib.import_module(mod)
        for part in attr.split("."):
            obj = getattr(obj, part)
        mine = {n: p for n, p in inspect.signature(obj).parameters.items() if n != "self"}
        has_kw = any(p.kind == p.VAR_KEYWORD for p in mine.values())
        for name, r in params.items():
            if name not in mine:
                assert has_kw, f"{key}: parameter {name!r} is not accepted"
                continue
            if "default" not in r or (attr, name) in _API_DEFAULT_EXCEPTIONS:
                continue
            d = mine[name].default
            assert d is not inspect.Parameter.empty, f"{key}: {name} is optional in the reference"
            assert (list(d) if isinstance(d, tuple) else d) == r["default"], (key, name, d, r["default"])


def test_fused_subsampling_operand_packing():
    """Host packers of the fused conv1 + conv2 kernel (em_conv2d_sub12_bf16): the fragment layouts stated in
    include/espnet_amd.h, emulated on the CPU, reproduce torch's convolutions (conv1 at f32-class accuracy through the
    split-bf16 operands)."""
    import torch
    import torch.nn.functional as F

    from espnet_amd.asr.encoder.conformer_encoder import pack_conv1_frags, pack_conv2_frags

    g = torch.Generator().manual_seed(5)
    d = 256
    w1 = torch.randn(d, 9, generator=g) / 3
    b1 = torch.randn(d, generator=g) * 0.1
    x = torch.randn(4, 9, generator=g) * 3 - 1          # 4 map positions x 9 taps (f32 inputs)
    a = pack_conv1_frags(w1, b1).reshape(8, 2, 4, 16, 8)  # [cc][f][lg][lr][e]

    def hi(v):
        return v.to(torch.bfloat16).to(torch.float32)

    xh, xl = hi(x), hi(x - hi(x))
    bop = torch.zeros(4, 32)                              # the kernel's position operand, k-slots as in the header
    bop[:, 0:9], bop[:, 9:18], bop[:, 18:27], bop[:, 27], bop[:, 28] = xh, xl, xh, 1.0, 1.0
    got = torch.empty(4, d)
    for c in range(d):
        cc, f, lr = c // 32, (c % 32) // 16, c % 16
        wk = a[cc, f, :, lr, :].reshape(32)               # k = 8 lg + e
        got[:, c] = (bop * wk[None, :]).sum(1)
    ref = x @ w1.t() + b1
    assert (got - ref).abs().max().item() < 2e-4 * ref.abs().max().item()

    w2 = torch.randn(d, 9 * d, generator=g)
    u = pack_conv2_frags(w2).reshape(8, 9, 4, 4, 4, 16, 8)  # [cc][tap][w][j][lg][lr][e]
    for (cc, tap, w, j, lg, lr, e) in [(0, 0, 0, 0, 0, 0, 0), (7, 8, 3, 3, 3, 15, 7), (3, 4, 1, 2, 2, 9, 5)]:
        assert u[cc, tap, w, j, lg, lr, e] == w2[64 * w + 16 * (lr // 4) + 4 * j + lr % 4, tap * 256 + 32 * cc + 8 * lg + e]
