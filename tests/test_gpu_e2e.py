"""GPU end-to-end parity: frontend -> Conformer encoder -> greedy CTC through the drop-in host API
(espnet_amd.tasks.asr.ASRTask / ESPnetASRModel), against the golden fixtures the reference produced
(tests/golden/make_golden.py) and against the oracle.

Stated tolerances:
  float32 mode : encoder activations abs 2e-3 (f32 round-off through 12 blocks, |x| ~ 1 after the
                 final LayerNorm); per-frame argmax ids bit-exact except where the REFERENCE's own
                 top-2 log-prob margin is below 1e-3 (a tie at f32 resolution); G1 tokens exact
                 when no such frame exists.
  bfloat16 mode: encoder activations abs 4e-2 / mean-abs 6e-3 (bf16 operands, f32 accumulate; 2x what is
                 measured).  Random-init posteriors are nearly flat, so frame ids / tokens CAN flip - but only
                 where the REFERENCE's own top-2 log-prob margin is below BF16_MARGIN: every mismatching frame
                 is checked for that, the rate is bounded at 3 %, and on the fixture with a fitted (peaked) CTC
                 head (`small_10s_peaked`) the bf16 path must return the reference's G1 tokens EXACTLY.
"""
import numpy as np
import pytest
import torch

from oracle import conformer as oc
from tests.helpers import golden_speech, golden_state_dict, hparams, load_golden

pytestmark = pytest.mark.gpu


def build(g, dtype):
    from espnet_amd.tasks.asr import ASRTask

    cfg = dict(g["config"])
    cfg["compute_dtype"] = dtype
    model = ASRTask.build_model(cfg)
    model.load_state_dict(golden_state_dict(g), strict=True)
    return model.cuda().eval()


# frames whose arg-max may differ from the reference's because the reference's own top-2 log-prob margin is
# below 1e-3 (f32 round-off decides them): bounded per fixture, printed by the test
EXCUSED_FRAMES = {"tiny_blocks": 0, "small_ragged": 2, "small_10s": 2, "large_10s": 2, "sub6_small_6s": 2,
                  "sub8_small_6s": 2, "legacy_small_5s": 2, "legacy_small_12s": 2}


@pytest.mark.parametrize("name", ["tiny_blocks", "small_ragged", "small_10s", "large_10s", "sub6_small_6s",
                                  "sub8_small_6s", "legacy_small_5s", "legacy_small_12s"])
def test_encode_float32_matches_reference(name):
    g = load_golden(name)
    model = build(g, "float32")
    speech, lens = golden_speech(g)
    st = model.encode_device(speech.cuda(), lens.tolist())
    torch.cuda.synchronize()
    assert st.flens == g["feats_lens"].tolist()
    assert st.olens == g["enc_olens"].tolist()
    np.testing.assert_allclose(st.feats.cpu().numpy(), g["feats"], atol=3e-4, rtol=0)
    ke = int(g["enc_keep_every"])
    enc = st.enc_out.cpu().numpy()[:, ::ke]
    err = np.abs(enc - g["enc_out"]).max()
    assert err < 2e-3, f"{name}: encoder max abs err {err:.3e}"
    ids, tokens, tlens = model.greedy_ctc_device(st)
    ids = ids.cpu().numpy()
    valid = np.arange(ids.shape[1])[None, :] < g["enc_olens"][:, None]
    diff = (ids != g["ctc_ids"]) & valid
    assert (g["ctc_margin"][diff] < 1e-3).all(), "argmax differs on a frame that is not a near-tie"
    n_excused = int(diff.sum())
    print(f"[{name}] f32 greedy ids: {n_excused} of {int(valid.sum())} frames differ, all with a reference "
          f"top-2 margin < 1e-3")
    assert n_excused <= EXCUSED_FRAMES[name], (name, n_excused)
    # G1 tokens are ALWAYS compared: the reference's per-frame ids with only the excused near-tie frames
    # replaced by the device's choice, collapsed by the reference rule (asr_inference.py:574-575), must
    # equal the device's tokens; with no excused frame this is the golden g1_tokens itself.
    blank, sos_eos = 0, int(g["vocab"]) - 1
    for b in range(ids.shape[0]):
        n_fr = int(g["enc_olens"][b])
        want_ids = np.where(diff[b], ids[b], g["ctc_ids"][b])[:n_fr]
        want = oc.g1_collapse(want_ids.tolist(), (blank, sos_eos))
        if not diff[b].any():
            assert want == g["g1_tokens"][b, : int(g["g1_lens"][b])].tolist()
        n = int(tlens[b])
        assert tokens[b, :n].cpu().tolist() == want, (name, b)


def _edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


# A bf16 frame-id flip is legitimate only on a frame the reference itself barely decides: its top-2 log-prob margin
# must be below this (measured on MI355X, round 3: the largest reference margin at a flipped frame is printed by the
# tests and by bench.py's `bf16_vs_f32.max_reference_margin_at_mismatch`; the bound is ~2x the largest seen).
BF16_MARGIN = 0.04


def margin_report(tag, margins_at_mismatch):
    """Histogram of the reference's top-2 margins at the mismatching frames (printed: run pytest -s)."""
    m = np.asarray(margins_at_mismatch, dtype=np.float64)
    edges = [0, 1e-3, 2e-3, 5e-3, 1e-2, 2e-2, 4e-2, 1e-1, 1e9]
    hist = np.histogram(m, bins=edges)[0].tolist() if m.size else [0] * (len(edges) - 1)
    print(f"[{tag}] {m.size} mismatching frames; reference top-2 margin there: max "
          f"{(m.max() if m.size else 0.0):.3e}; histogram over {edges[:-1]} + inf: {hist}")
    return float(m.max()) if m.size else 0.0


@pytest.mark.parametrize("fused", [True, False, "fold", "split"])
@pytest.mark.parametrize("name", ["small_ragged", "small_10s", "large_10s"])
def test_encode_bfloat16_within_tolerance(name, fused):
    """bf16 MFMA mode (the timed mode), both launch sequences: the fused per-block kernels (csrc/block.hip; they
    apply to the 256-wide fixtures) and the one-operator-per-launch sequence.  Falsifiable form (VERDICT r02
    item 3): every frame whose arg-max differs from the reference's must be a frame the reference itself decides
    by less than BF16_MARGIN; bounds at ~2x the measured values."""
    g = load_golden(name)
    model = build(g, "bfloat16")
    model.encoder.fused = bool(fused)
    model.encoder.fold_c = fused == "fold"  # block<C|D|...> (round 4; ragged batch: utterance boundaries inside the halo)
    model.encoder.split_att = fused == "split"  # attention + block<C> as two launches (rounds 2-5); default (round 6): block<ATT|C>
    speech, lens = golden_speech(g)
    st = model.encode_device(speech.cuda(), lens.tolist())
    ke = int(g["enc_keep_every"])
    enc = st.enc_out.cpu().numpy()[:, ::ke]
    err = np.abs(enc - g["enc_out"])
    print(f"[{name} fused={fused}] bf16 encoder err max {err.max():.3e} mean {err.mean():.3e}")
    assert err.max() < 4e-2 and err.mean() < 6e-3
    ids, tokens, tlens = model.greedy_ctc_device(st)
    valid = np.arange(ids.shape[1])[None, :] < g["enc_olens"][:, None]
    diff = (ids.cpu().numpy() != g["ctc_ids"]) & valid
    mism = diff.sum() / valid.sum()
    worst = margin_report(f"{name} fused={fused}", g["ctc_margin"][diff])
    assert worst < BF16_MARGIN, f"a frame the reference decides by {worst:.3e} flipped in bf16"
    # token level: edit distance between the device's G1 tokens and the reference's, per reference token
    dist = ref_len = 0
    for b in range(ids.shape[0]):
        want = g["g1_tokens"][b, : int(g["g1_lens"][b])].tolist()
        dist += _edit_distance(tokens[b, : int(tlens[b])].cpu().tolist(), want)
        ref_len += len(want)
    print(f"[{name} fused={fused}] bf16 greedy-id mismatch rate {mism:.3%}, G1 token edit distance "
          f"{dist} / {ref_len} reference tokens")
    assert mism < 0.03
    assert dist <= 0.03 * max(ref_len, 1) + 1


def test_large_rows_path_at_b64_matches_reference():
    """Round 4: the 512-wide model's row-block launches (csrc/ffn_rows.hip: three per block, the CTC arg-max walked behind the
    last one) are taken only when a batch's rows fill the chip - none of the small-batch fixtures reaches them.  Here the
    reference's `large_10s` utterance is encoded as a batch of 64 copies (M = 15 936 rows: the bench's shape): every row must
    meet the reference's encoder output and per-frame CTC ids under the same bounds as the B = 1 test, all rows must be
    bit-identical (a workgroup sees nothing but its own rows), and the per-operator sequence must agree to bf16 round-off."""
    g = load_golden("large_10s")
    model = build(g, "bfloat16")
    speech, lens = golden_speech(g)
    B = 64
    wav = speech[:1].repeat(B, 1).cuda()
    ln = [int(lens[0])] * B
    model.encoder.fused = True
    st = model.encode_device(wav, ln)
    assert model.encoder.last_ctc_ids is not None, "the row-block path (with its CTC walk) was not taken at B = 64"
    ke = int(g["enc_keep_every"])
    enc = st.enc_out.float().cpu()
    for k in (1, 31, 63):
        assert torch.equal(enc[0], enc[k]), k
    err = np.abs(enc[0].numpy()[::ke] - g["enc_out"][0])
    print(f"[large_10s x 64, row-block path] bf16 encoder err max {err.max():.3e} mean {err.mean():.3e}")
    assert err.max() < 4e-2 and err.mean() < 6e-3
    ids, tokens, tlens = model.greedy_ctc_device(st)
    ids = ids.cpu().numpy()
    assert (ids == ids[:1]).all()
    T = int(g["enc_olens"][0])
    diff = ids[0, :T] != g["ctc_ids"][0, :T]
    worst = margin_report("large_10s x 64 row-block", g["ctc_margin"][0, :T][diff])
    assert worst < BF16_MARGIN, f"a frame the reference decides by {worst:.3e} flipped in bf16"
    assert diff.mean() < 0.03
    model.encoder.fused = False  # the per-operator sequence on the same batch
    st2 = model.encode_device(wav, ln)
    assert model.encoder.last_ctc_ids is None
    d = (st2.enc_out.float().cpu()[0] - enc[0]).abs()
    print(f"[large_10s x 64] row-block vs per-operator bf16: max {d.max():.3e} mean {d.mean():.3e}")
    assert d.max() < 0.08 and d.mean() < 6e-3


@pytest.mark.parametrize("B,fused", [(64, True), (64, False), (1, True)])
def test_large_peaked_posteriors_ids_exact(B, fused):
    """`large_10s_peaked` (round 5): the 512-wide model with a CTC head fitted to the reference's encoder output (reference
    top-2 margins all > 4).  As a batch of 64 copies the encoder runs through the row-block launches (csrc/ffn_rows.hip)
    with the CTC arg-max walked behind the last one: per-frame ids and G1 tokens must be the reference's EXACTLY
    (asr/ctc.py:207-215, asr_inference.py:574-575), on every row; likewise through the per-operator sequence and at B = 1."""
    g = load_golden("large_10s_peaked")
    assert float(g["ctc_margin"].min()) > 1.0
    model = build(g, "bfloat16")
    model.encoder.fused = fused
    speech, lens = golden_speech(g)
    wav = speech[:1].repeat(B, 1).cuda()
    st = model.encode_device(wav, [int(lens[0])] * B)
    if B == 64 and fused:
        assert model.encoder.last_ctc_ids is not None, "the row-block path (with its CTC walk) was not taken at B = 64"
    ids, tokens, tlens = model.greedy_ctc_device(st)
    n_fr = int(g["enc_olens"][0])
    want_ids = g["ctc_ids"][0, :n_fr].tolist()
    want = g["g1_tokens"][0, : int(g["g1_lens"][0])].tolist()
    assert len(want) >= 20
    for b in sorted({0, B // 2, B - 1}):
        assert ids[b, :n_fr].cpu().tolist() == want_ids, b
        assert tokens[b, : int(tlens[b])].cpu().tolist() == want, b
    assert (ids.cpu() == ids[:1].cpu()).all()


@pytest.mark.parametrize("dtype,fused", [("float32", False), ("bfloat16", True), ("bfloat16", False), ("bfloat16", "fold"),
                                         ("bfloat16", "split")])
def test_peaked_posteriors_tokens_exact(dtype, fused):
    """`small_10s_peaked`: the small model with a CTC head fitted to the reference's encoder output (reference
    top-2 margins all > 1, tests/golden/make_golden.py::fit_peaked_ctc_head).  With posteriors like a trained
    model's, bf16 round-off has no frame to flip: the timed bf16 fused path - CTC arg-max inside the last block
    kernel - must return the reference's per-frame ids and G1 tokens exactly (asr_inference.py:574-575)."""
    g = load_golden("small_10s_peaked")
    assert float(g["ctc_margin"].min()) > 1.0
    model = build(g, dtype)
    model.encoder.fused = bool(fused)
    model.encoder.fold_c = fused == "fold"  # block<C|D|...>: the two-launch sequence (EM_ENC_FOLD_C)
    model.encoder.split_att = fused == "split"  # relpos_attn2 + block<C> (EM_ENC_SPLIT_ATT) instead of block<ATT|C>
    speech, lens = golden_speech(g)
    st = model.encode_device(speech.cuda(), lens.tolist())
    ids, tokens, tlens = model.greedy_ctc_device(st)
    if fused:
        assert model.encoder.last_ctc_ids is not None  # the arg-max really came from block<D|FINAL|CTC>
    n_fr = int(g["enc_olens"][0])
    assert ids[0, :n_fr].cpu().tolist() == g["ctc_ids"][0, :n_fr].tolist()
    want = g["g1_tokens"][0, : int(g["g1_lens"][0])].tolist()
    assert len(want) >= 20
    assert tokens[0, : int(tlens[0])].cpu().tolist() == want


def test_fused_blocks_match_per_operator_sequence_bf16():
    """The two bf16 launch sequences round at the same points (LayerNorm output, FFN hidden, GLU and conv
    outputs, q / k / v, attention probabilities and context) and differ only in f32 summation order: their
    encoder outputs agree far more closely than either agrees with the f32 reference."""
    for name in ("small_10s", "small_ragged", "tiny_blocks"):
        g = load_golden(name)
        model = build(g, "bfloat16")
        speech, lens = golden_speech(g)
        outs = []
        for fused in (True, False):
            model.encoder.fused = fused
            for isolate in (False, True):
                st = model.encode_device(speech.cuda(), lens.tolist(), isolate=isolate)
                outs.append(st.enc_out.float().cpu())
        for a, b in ((outs[0], outs[2]), (outs[1], outs[3])):
            d = (a - b).abs()
            print(f"[{name}] fused vs per-operator bf16: max {d.max():.3e} mean {d.mean():.3e}")
            assert d.max() < 0.08 and d.mean() < 6e-3


def test_position_rows_packed_by_the_host_or_by_the_call_are_the_same():
    """block<ATT|C> reads the position rows fragment-major.  `ConformerEncoder._pos_projected` packs them once per length behind
    the projected table (EM_ENC_POS_PACKED); a caller that hands over the projected table alone - or the raw table - has
    `em_conformer_encode` pack them per call: same launch sequence, bit-identical encoder outputs and CTC ids, ragged batch."""
    g = load_golden("small_ragged")
    model = build(g, "bfloat16")
    speech, lens = golden_speech(g)
    enc = model.encoder
    st = model.encode_device(speech.cuda(), lens.tolist())
    ref_out, ref_ids = st.enc_out.clone(), model.greedy_ctc_device(st)[0].clone()
    orig = enc._pos_projected
    try:
        enc._pos_projected = lambda T, dev, pk, pos: orig(T, dev, pk, pos).clone()  # (a clone carries no `_em_packed` mark)
        st2 = model.encode_device(speech.cuda(), lens.tolist())
        assert torch.equal(st2.enc_out, ref_out)
        assert torch.equal(model.greedy_ctc_device(st2)[0], ref_ids)
    finally:
        enc._pos_projected = orig


def test_reference_api_encode_and_ctc():
    """`model.encode(speech, lengths)` / `ctc.argmax` / `ctc.log_softmax` keep the reference
    signatures (espnet_model.py:380, ctc.py:197-215)."""
    g = load_golden("small_ragged")
    model = build(g, "float32")
    speech, lens = golden_speech(g)
    enc, olens = model.encode(speech.cuda(), lens.cuda())
    assert olens.tolist() == g["enc_olens"].tolist() and enc.shape == (3, 74, 256)
    ids = model.ctc.argmax(enc)
    assert ids.dtype == torch.int64
    logp = model.ctc.log_softmax(enc)
    np.testing.assert_allclose(logp[:, :4].cpu().numpy(), g["ctc_logp_head"], atol=2e-3)
    assert torch.equal(ids, logp.argmax(-1))


def test_too_short_raises():
    from espnet_amd.lib import TooShortUttError

    g = load_golden("tiny_blocks")
    model = build(g, "float32")
    with pytest.raises(TooShortUttError):
        model.encode_device(torch.zeros(1, 800).cuda(), [800])  # 6 frames < 7


def test_cpu_tensor_fails_loudly():
    from espnet_amd.lib import EspnetAmdError

    g = load_golden("tiny_blocks")
    model = build(g, "float32")
    with pytest.raises(EspnetAmdError):
        model.encode_device(torch.zeros(1, 16000), [16000])


def test_long_and_ragged_utterances_f32():
    """Sizes beyond the fixtures: a 30 s utterance (T = 749 encoder frames: several attention key
    tiles, rel-pos window > 256) batched with a 12.3 s one; f32 encoder vs the oracle, and the
    greedy tokens bit-exact wherever the oracle's own top-2 margin is not round-off."""
    import yaml

    from espnet_amd.tasks.asr import ASRTask
    from oracle.weights import synth_waveform

    g = load_golden("small_10s")
    sd = golden_state_dict(g)
    cfg = dict(g["config"])
    cfg["compute_dtype"] = "float32"
    model = ASRTask.build_model(cfg)
    model.load_state_dict(sd, strict=True)
    model.cuda().eval()
    lens = [480000, 196800]
    speech = torch.zeros(2, max(lens))
    for i, n in enumerate(lens):
        speech[i, :n] = synth_waveform(40 + i, n)
    hp = hparams(g)
    with torch.no_grad():
        ref_enc, ref_olens = oc.encode(sd, speech, torch.tensor(lens), hp["heads"], hp["num_blocks"],
                                       hp["n_fft"], hp["win_length"], hp["hop"])
        ref_logp = oc.ctc_log_softmax(sd, ref_enc)
    st = model.encode_device(speech.cuda(), lens)
    assert st.olens == ref_olens.tolist() and st.olens[0] == 749
    for b in range(2):
        err = (st.enc_out[b, : st.olens[b]].cpu() - ref_enc[b, : st.olens[b]]).abs().max().item()
        assert err < 2e-3, (b, err)
    ids, tokens, tlens = model.greedy_ctc_device(st)
    top2 = ref_logp.topk(2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1])
    ref_ids = ref_logp.argmax(-1)
    for b in range(2):
        n = st.olens[b]
        diff = ids[b, :n].cpu().long() != ref_ids[b, :n]
        assert (margin[b, :n][diff] < 1e-4).all()
        assert diff.float().mean().item() < 0.01


def test_isolated_utterance_batching_is_transparent_f32():
    """`encode_device(..., isolate=True)` (what Speech2Text.batch_decode and the decode CLI use): every
    row of a ragged batch equals the ORACLE's encoding of that utterance decoded alone — reflect padding
    at the utterance's own end, depthwise conv blind to the padded frames — including a 0.56 s
    utterance (T = 13) next to a 2.6 s one.  The padded-batch mode (isolate=False) is the reference's
    `ESPnetASRModel.encode` on the padded batch and must differ at the short rows' tail."""
    from espnet_amd.tasks.asr import ASRTask
    from oracle.weights import synth_waveform

    g = load_golden("small_10s")
    sd = golden_state_dict(g)
    cfg = dict(g["config"])
    cfg["compute_dtype"] = "float32"
    model = ASRTask.build_model(cfg)
    model.load_state_dict(sd, strict=True)
    model.cuda().eval()
    lens = [41000, 9000, 23456, 41000, 1500]
    speech = torch.zeros(len(lens), max(lens))
    for i, n in enumerate(lens):
        speech[i, :n] = synth_waveform(60 + i, n)
    hp = hparams(g)
    st = model.encode_device(speech.cuda(), lens, isolate=True)
    st_pad = model.encode_device(speech.cuda(), lens, isolate=False)
    worst_pad = 0.0
    for b, n in enumerate(lens):
        with torch.no_grad():
            ref, ol = oc.encode(sd, speech[b : b + 1, :n], torch.tensor([n]), hp["heads"], hp["num_blocks"],
                                hp["n_fft"], hp["win_length"], hp["hop"])
        T = int(ol[0])
        assert st.olens[b] == T
        err = (st.enc_out[b, :T].cpu() - ref[0]).abs().max().item()
        assert err < 2e-3, (b, err)
        worst_pad = max(worst_pad, (st_pad.enc_out[b, :T].cpu() - ref[0]).abs().max().item())
    assert worst_pad > 1e-2  # the two semantics are genuinely different on ragged input
    # equal lengths: both modes are the same computation
    eq = speech[[0, 3]].cuda()
    a = model.encode_device(eq, [41000, 41000], isolate=True).enc_out
    b = model.encode_device(eq, [41000, 41000], isolate=False).enc_out
    assert torch.equal(a, b)


def test_two_minute_utterance_greedy_and_search_limits():
    """Maximum sizes: a 120 s utterance (T = 2 999 encoder frames) goes through frontend + encoder + greedy
    CTC (property checks: token ids valid, no repeats of the collapsed kind, deterministic), while the joint
    CTC/attention search refuses memories beyond its CTC frame capacity loudly instead of truncating."""
    from espnet_amd.tasks.asr import ASRTask
    from oracle.weights import synth_waveform

    g = load_golden("small_10s")
    cfg = dict(g["config"])
    cfg["compute_dtype"] = "bfloat16"
    model = ASRTask.build_model(cfg)
    model.load_state_dict(golden_state_dict(g), strict=True)
    model.cuda().eval()
    n = 120 * 16000
    wav = synth_waveform(77, n)[None].cuda()
    st = model.encode_device(wav, [n])
    assert st.olens == [2999]
    assert torch.isfinite(st.enc_out).all()
    ids, tokens, tlens = model.greedy_ctc_device(st)
    k = int(tlens[0])
    tok = tokens[0, :k].tolist()
    V = int(g["vocab"])
    assert 0 < k <= 2999 and all(0 < t < V - 1 for t in tok)  # blank / sos / eos never survive the collapse
    ids_h = ids[0, :2999].tolist()
    collapsed = [a for a, b in zip(ids_h, [None] + ids_h[:-1]) if a != b and a not in (0, V - 1)]
    assert collapsed == tok  # G1 = groupby + drop blank/eos, recomputed on the host from the frame ids
    ids2, tokens2, tlens2 = model.greedy_ctc_device(model.encode_device(wav, [n]))
    assert torch.equal(tokens, tokens2) and torch.equal(tlens, tlens2)
    from espnet_amd.nets.batch_beam_search import build_beam_search

    bs = build_beam_search(model, beam_size=2, ctc_weight=0.3, penalty=0.0, token_list=model.token_list)
    with pytest.raises(NotImplementedError):
        bs.search_batch(st.enc_act, st.olens)


@pytest.mark.parametrize("name", ["small_ragged", "small_10s"])
def test_fused_ctc_argmax_equals_ctc_head_on_same_rows(name):
    """The last fused block kernel also takes the CTC head's per-frame arg-max (EM_BLOCK_CTC: the logits never reach
    memory).  Same ids as `CTC.argmax` (the tiled GEMM + arg-max kernels) over the SAME encoder rows; the two sum
    their 256 products in different orders, so a frame may differ only where its top-2 logit margin is round-off."""
    g = load_golden(name)
    model = build(g, "bfloat16")
    speech, lens = golden_speech(g)
    st = model.encode_device(speech.cuda(), lens.tolist())
    assert st.ctc_ids is not None and st.ctc_ids.dtype == torch.int32
    ids_fused, tok_fused, tl_fused = model.greedy_ctc_device(st)
    assert ids_fused.data_ptr() == st.ctc_ids.data_ptr()
    blank, sos = model.blank_id, model.sos
    ids_ref, tok_ref, tl_ref = model.ctc.greedy_device(st.enc_act, st.olens_dev, blank, sos if sos == model.eos else -2)
    B, T = ids_ref.shape
    valid = (torch.arange(T)[None, :] < torch.tensor(st.olens)[:, None])
    logits = model.ctc.logits_device(st.enc_act).view(B, T, -1)
    top2 = logits.topk(2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1]).cpu()
    diff = (ids_fused.cpu() != ids_ref.cpu()) & valid
    assert bool((margin[diff] < 1e-3).all()), int(diff.sum())
    assert int(diff.sum()) <= 2
    assert int(ids_fused.cpu()[valid].min()) >= 0 and int(ids_fused.cpu()[valid].max()) < model.vocab_size
    if not diff.any():
        assert torch.equal(tok_fused.cpu(), tok_ref.cpu()) and torch.equal(tl_fused.cpu(), tl_ref.cpu())
    # the one-operator-per-launch sequence has no fused head
    model.encoder.fused = False
    assert model.encode_device(speech.cuda(), lens.tolist()).ctc_ids is None


@pytest.mark.parametrize("fused", [True, False])
def test_midmargin_posteriors_bf16_flips_only_below_margin(fused):
    """`small_10s_midmargin`: the fitted CTC head scaled down so that the REFERENCE's top-2 margins fill 0.04 .. 1.5
    (tests/golden/make_golden.py: peaked_level) - the band between the flat random-init fixtures (margins < 0.04)
    and `small_10s_peaked` (margins > 1.39), where a trained model's hard frames live.  The bf16 path
    may flip a frame only if the reference decides it by less than BF16_MARGIN; every frame above must carry the
    reference's id, and the G1 tokens must be the reference's wherever no excusable frame is involved."""
    g = load_golden("small_10s_midmargin")
    margin = g["ctc_margin"][0]
    n_fr = int(g["enc_olens"][0])
    band = int(((margin[:n_fr] >= BF16_MARGIN) & (margin[:n_fr] <= 1.0)).sum())
    assert band >= 150, band  # the band is populated (182 of 249 frames when generated)
    model = build(g, "bfloat16")
    model.encoder.fused = fused
    speech, lens = golden_speech(g)
    st = model.encode_device(speech.cuda(), lens.tolist())
    ids, tokens, tlens = model.greedy_ctc_device(st)
    ids = ids.cpu().numpy()
    diff = ids[0, :n_fr] != g["ctc_ids"][0, :n_fr]
    worst = margin_report(f"small_10s_midmargin fused={fused}", margin[:n_fr][diff])
    assert worst < BF16_MARGIN, f"a frame the reference decides by {worst:.3e} flipped in bf16"
    print(f"[small_10s_midmargin fused={fused}] {int(diff.sum())} of {n_fr} frames flipped (all below {BF16_MARGIN}); "
          f"{band} frames with a reference margin in [{BF16_MARGIN}, 1] all carry the reference's id")
    from oracle import conformer as oc

    want = oc.g1_collapse(np.where(diff, ids[0, :n_fr], g["ctc_ids"][0, :n_fr]).tolist(), (0, int(g["vocab"]) - 1))
    assert tokens[0, : int(tlens[0])].cpu().tolist() == want
