"""CPU: pin the oracle (oracle/conformer.py) against fixtures produced by the reference itself
(tests/golden/make_golden.py).  Tolerances are fp32 round-off of a different summation order."""
import numpy as np
import pytest
import torch

from oracle import conformer as oc
from oracle.mel import slaney_mel_filterbank
from tests.helpers import golden_speech, golden_state_dict, hparams, load_golden

torch.set_grad_enabled(False)


@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax", [(16000, 512, 80, 0, 8000), (16000, 400, 80, 0, 8000),
                                                       (22050, 2048, 128, 0, 11025), (8000, 256, 40, 50, 3800)])
def test_mel_filterbank_against_an_independent_implementation(sr, n_fft, n_mels, fmin, fmax):
    """librosa is absent, so the restatement of `librosa.filters.mel` is pinned against an independent
    implementation of the same published algorithm that is in the image (Hugging Face transformers' Slaney
    filterbank, itself tested against librosa upstream): equal to float32 round-off, same non-zero support; and
    against the value librosa's own documentation prints for its default example."""
    tau = pytest.importorskip("transformers.audio_utils")
    mine = slaney_mel_filterbank(sr=sr, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax)
    other = tau.mel_filter_bank(num_frequency_bins=1 + n_fft // 2, num_mel_filters=n_mels, min_frequency=fmin,
                                max_frequency=fmax, sampling_rate=sr, norm="slaney", mel_scale="slaney").T
    assert mine.dtype == np.float32 and mine.shape == other.shape == (n_mels, 1 + n_fft // 2)
    assert np.abs(mine.astype(np.float64) - other).max() < 5e-9
    assert ((mine > 0) == (other > 0)).all()
    if (sr, n_fft, n_mels) == (22050, 2048, 128):  # >>> librosa.filters.mel(sr=22050, n_fft=2048) -> [[0., 0.016, ...
        assert round(float(mine[0, 1]), 3) == 0.016 and mine[0, 0] == 0.0


def test_melmat_matches_reference_buffer():
    g = load_golden("small_10s")
    m = slaney_mel_filterbank(sr=16000, n_fft=512, n_mels=80, fmin=0, fmax=8000).T
    assert m.shape == (257, 80)
    np.testing.assert_array_equal(m, g["melmat"])
    # every rFFT bin feeds at most two neighbouring filters (triangles overlap by one)
    assert ((m > 0).sum(1) <= 2).all()


def test_stft_olens_formula():
    # test/espnet2/layers/test_stft.py:10-16: n_fft 4 / hop 2, ilens [30,15] -> olens [16,8]
    sp = torch.randn(2, 30)
    lens = torch.tensor([30, 15])
    mel = torch.rand(3, 2)
    feats, olens = oc.frontend_feats(sp, lens, mel, n_fft=4, win_length=4, hop=2)
    assert olens.tolist() == [16, 8]
    assert feats.shape == (2, 16, 2)
    assert (feats[1, 8:] == 0).all()


@pytest.mark.parametrize("name", ["tiny_blocks", "small_ragged", "small_10s", "small_10s_peaked", "small_10s_midmargin", "large_10s", "large_10s_peaked",
                                  "sub6_small_6s", "sub8_small_6s", "legacy_small_5s", "legacy_small_12s"])
def test_frontend_and_encoder_match_reference(name):
    g = load_golden(name)
    sd = golden_state_dict(g)
    hp = hparams(g)
    speech, lens = golden_speech(g)
    feats, flens = oc.frontend_feats(speech, lens, sd["frontend.logmel.melmat"], hp["n_fft"],
                                     hp["win_length"], hp["hop"])
    assert flens.tolist() == g["feats_lens"].tolist()
    np.testing.assert_allclose(feats.numpy(), g["feats"], atol=2e-4, rtol=0)
    enc, olens = oc.encode(sd, speech, lens, hp["heads"], hp["num_blocks"], hp["n_fft"],
                           hp["win_length"], hp["hop"],
                           rel_pos_type=g["config"]["encoder_conf"].get("rel_pos_type", "latest"))
    assert olens.tolist() == g["enc_olens"].tolist()
    ke = int(g["enc_keep_every"])
    np.testing.assert_allclose(enc[:, ::ke].numpy(), g["enc_out"], atol=5e-4, rtol=0)
    if "block_outs" in g and name.startswith("legacy"):
        feats2 = oc.utterance_mvn(feats, flens)
        _, _, blocks = oc.conformer_encoder(sd, feats2, flens, hp["heads"], hp["num_blocks"], return_blocks=True,
                                            rel_pos_type="legacy")
        for i, b in enumerate(blocks):
            np.testing.assert_allclose(b.numpy(), g["block_outs"][i], atol=1e-4, rtol=0)
        np.testing.assert_allclose(oc.legacy_pos_emb(blocks[0].size(1), blocks[0].size(2)).numpy(), g["pos_emb"][0],
                                   atol=1e-6)
    ids = oc.ctc_argmax(sd, enc).numpy()
    # argmax can only differ where the reference's own top-2 margin is within round-off
    diff = ids != g["ctc_ids"]
    assert (g["ctc_margin"][diff] < 1e-4).all()
    assert diff.mean() < 0.01
    if name.endswith("_peaked"):  # the fitted head: margins far above round-off, so the oracle has no excuse
        assert not diff.any() and g["ctc_margin"].min() > 1.0 and int(g["g1_lens"][0]) >= 20
    if not diff.any():
        eos = int(g["vocab"]) - 1
        toks = oc.greedy_ctc(sd, enc, olens, blank=0, sos_eos=eos)
        for b, t in enumerate(toks):
            assert t == g["g1_tokens"][b, : g["g1_lens"][b]].tolist()


def test_block_outputs_match_reference():
    g = load_golden("tiny_blocks")
    sd = golden_state_dict(g)
    hp = hparams(g)
    speech, lens = golden_speech(g)
    feats, flens = oc.frontend_feats(speech, lens, sd["frontend.logmel.melmat"], hp["n_fft"],
                                     hp["win_length"], hp["hop"])
    feats = oc.utterance_mvn(feats, flens)
    x = oc.conv2d_subsampling(sd, feats) * (64 ** 0.5)
    np.testing.assert_allclose(x.numpy(), g["embed_x"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(oc.rel_pos_emb(x.size(1), 64).numpy(), g["pos_emb"][0], atol=1e-6)
    _, _, blocks = oc.conformer_encoder(sd, feats, flens, hp["heads"], hp["num_blocks"],
                                        return_blocks=True)
    for i, b in enumerate(blocks):
        np.testing.assert_allclose(b.numpy(), g["block_outs"][i], atol=1e-4, rtol=0)


def test_rel_shift_identity():
    # rel_shift(BD)[i, j] = BD[i, T-1-i+j]
    T = 9
    bd = torch.randn(1, 2, T, 2 * T - 1)
    out = oc.rel_shift(bd)
    i = torch.arange(T)[:, None]
    j = torch.arange(T)[None, :]
    ref = bd[:, :, i, T - 1 - i + j]
    assert torch.equal(out, ref)


def test_too_short():
    g = load_golden("tiny_blocks")
    sd = golden_state_dict(g)
    with pytest.raises(oc.TooShortUttError):
        oc.conformer_encoder(sd, torch.zeros(1, 6, 80), torch.tensor([6]), 1, 2)


# --------------------------------------------------------------------------- beam search (A12-A14)
SEARCH_CASES = ["tiny_beam5", "tiny_beam3_attn_only", "tiny_beam4_early_eos", "tiny_beam4_minlen",
                "small_g2_3s", "large_beam10_3s", "large_beam10_3s_peaked"]


def run_oracle_search(g):
    from oracle import beam_search as ob

    sd = golden_state_dict(g)
    hp = hparams(g)
    speech, lens = golden_speech(g)
    enc, _ = oc.encode(sd, speech, lens, hp["heads"], hp["num_blocks"], hp["n_fft"],
                       hp["win_length"], hp["hop"])
    V = int(g["vocab"])
    dc = g["config"]["decoder_conf"]
    kw = {k: float(g[k]) for k in ("penalty", "maxlenratio", "minlenratio") if k in g}
    return ob.beam_search(sd, enc[0], dc["attention_heads"], dc["num_blocks"], int(g["beam"]),
                          float(g["ctc_weight"]), sos=V - 1, eos=V - 1, **kw)


@pytest.mark.parametrize("name", SEARCH_CASES)
def test_beam_search_oracle_matches_reference_speech2text(name):
    """The restated BatchBeamSearch + CTCPrefixScoreTH + K/V-cached decoder reproduces the
    reference's Speech2Text n-best lists: identical yseq, scores within fp32 round-off."""
    import json

    g = load_golden(name)
    res = run_oracle_search(g)
    keys = json.loads(str(g["score_keys"]))
    n = len(g["yseq_lens"])
    assert len(res) >= n
    for k in range(n):
        ref = g["yseq"][k, : g["yseq_lens"][k]].tolist()
        assert res[k]["yseq"] == ref, (name, k)
        tol = 1e-5 * max(1.0, abs(float(g["score"][k]))) + 1e-4
        assert abs(res[k]["score"] - float(g["score"][k])) < tol
        for j, kk in enumerate(keys):
            assert abs(res[k]["scores"][kk] - float(g["scores"][k, j])) < 1e-5 * abs(float(g["scores"][k, j])) + 1e-3


def test_ctc_prefix_scorer_properties():
    """Size-independent properties of the CTC prefix recurrence (no reference test pins
    CTCPrefixScoreTH numerics): restricting the candidates does not change their scores, and for
    the empty prefix psi(c) = log sum_t P(first non-blank label is c, emitted at t)."""
    from oracle import beam_search as ob

    torch.manual_seed(3)
    T, V = 23, 11
    logp = torch.log_softmax(torch.randn(T, V), -1)
    sc = ob.CtcPrefixScorer(logp, eos=V - 1)
    r0, s0 = sc.initial_state()
    last = torch.tensor([V - 1])
    full, r_full, psi_full = sc.score(0, last, r0, s0, None)
    ids = torch.tensor([[3, 7, 1, 5]])
    part, r_part, psi_part = sc.score(0, last, r0, s0, ids)
    np.testing.assert_allclose(part[0, ids[0]].numpy(), full[0, ids[0]].numpy(), rtol=0, atol=1e-5)
    # closed form for the empty prefix: psi(c) = logsumexp_t( sum_{u<t} logp[u,blank] + logp[t,c] )
    cum = torch.cat([torch.zeros(1), torch.cumsum(logp[:, 0], 0)[:-1]])
    want = torch.logsumexp(cum[:, None] + logp, 0)
    np.testing.assert_allclose(psi_full[0, 1:V - 1].numpy(), want[1:V - 1].numpy(), rtol=0, atol=1e-4)
    # eos entry = log P(prefix is complete) = total blank path; blank entry = logzero
    assert abs(float(psi_full[0, V - 1]) - float(torch.cumsum(logp[:, 0], 0)[-1])) < 1e-4
    assert float(psi_full[0, 0]) == ob.LOGZERO


# --------------------------------------------------------------------------- streaming encoder (A16)
@pytest.mark.parametrize("name", ["stream_tiny_4s", "stream_tiny_short", "stream_small_6s"])
def test_streaming_encoder_oracle_matches_reference(name):
    """oracle/streaming.py == ContextualBlockConformerEncoder.forward_infer fed chunk by chunk:
    same number of emitted frames per call, outputs within fp32 round-off; the one-shot
    (is_final) call agrees too."""
    from oracle.streaming import CBEncoderOracle
    from tests.helpers import load_stream_golden, stream_feats

    g = load_stream_golden(name)
    c = g["conf"]
    feats = stream_feats(int(g["utt_id"]), int(g["n_samples"]))
    assert feats.size(0) == int(g["n_feat_frames"])
    orc = CBEncoderOracle(g["sd"], c["attention_heads"], c["num_blocks"], c["block_size"],
                          c["hop_size"], c["look_ahead"])
    cf = int(g["chunk_frames"])
    outs, lens, state, pos = [], [], None, 0
    while pos < feats.size(0):
        nxt = min(feats.size(0), pos + cf)
        y, state = orc.forward_infer(feats[pos:nxt], state, nxt == feats.size(0))
        outs.append(y)
        lens.append(y.size(0))
        pos = nxt
    assert lens == g["out_lens"].tolist()
    ke = int(g["keep_every"])
    ys = torch.cat(outs, 0)
    np.testing.assert_allclose(ys[::ke].numpy(), g["ys"], atol=2e-4, rtol=0)
    y1, _ = orc.forward_infer(feats, None, True)
    np.testing.assert_allclose(y1[::ke].numpy(), g["ys_oneshot"], atol=2e-4, rtol=0)


@pytest.mark.parametrize("name", ["tiny_beam5_lm", "tiny_beam4_lm_posenc", "tiny_beam60_lm_v300", "e2e_beam5_lm",
                                  "tiny_beam5_rnnlm", "tiny_beam4_rnnlm_nhid", "tiny_beam5_gru", "tiny_beam4_gru_nhid",
                                  "tiny_beam4_rnn_tanh", "tiny_beam4_rnn_relu"])
def test_beam_search_with_lm_scorer_matches_reference(name):
    """TransformerLM as a full scorer (lm_weight) — oracle vs reference n-best."""
    import json

    from oracle import beam_search as ob
    from oracle.weights import recipe_state_dict

    g = load_golden(name)
    sd = golden_state_dict(g)
    lm_shapes = {"lm." + k: tuple(v) for k, v in json.loads(str(g["lm_state_shapes"])).items()}
    sd.update(recipe_state_dict(lm_shapes, int(g["wseed"]), skip=()))
    hp = hparams(g)
    speech, lens = golden_speech(g)
    enc, _ = oc.encode(sd, speech, lens, hp["heads"], hp["num_blocks"], hp["n_fft"], hp["win_length"], hp["hop"])
    V = int(g["vocab"])
    dc = g["config"]["decoder_conf"]
    res = ob.beam_search(sd, enc[0], dc["attention_heads"], dc["num_blocks"], int(g["beam"]),
                         float(g["ctc_weight"]), sos=V - 1, eos=V - 1, lm_weight=float(g["lm_weight"]),
                         lm_conf=json.loads(str(g["lm_conf"])))
    keys = json.loads(str(g["score_keys"]))
    for k in range(len(g["yseq_lens"])):
        assert res[k]["yseq"] == g["yseq"][k, : g["yseq_lens"][k]].tolist()
        assert abs(res[k]["score"] - float(g["score"][k])) < 1e-4 + 1e-5 * abs(float(g["score"][k]))
        for j, kk in enumerate(keys):
            assert abs(res[k]["scores"][kk] - float(g["scores"][k, j])) < 1e-3


@pytest.mark.parametrize("name", ["stream_search_a", "stream_search_b", "stream_search_c", "stream_search_lm",
                                  "stream_search_rnnlm", "stream_search_gru", "stream_search_peaked"])
def test_online_beam_search_matches_reference_per_call(name):
    """BatchBeamSearchOnline (block-synchronous search with CTC extend_prob /
    extend_state, repetition / local-<eos> breaks, rewind, end detection) — the oracle replays the
    encoder frames each `Speech2TextStreaming.__call__` handed to the search and must return the same
    n-best per call (token sequences exactly, scores to fp32 round-off)."""
    import json

    from oracle.beam_search_online import OnlineBeamSearchOracle

    g = load_golden(name)
    sd = golden_state_dict(g)
    V = int(g["vocab"])
    dc = g["config"]["decoder_conf"]
    lm_kw = {}
    if "lm_conf" in g and json.loads(str(g["lm_conf"])) is not None:
        from oracle.weights import recipe_state_dict

        shapes = {"lm." + k: tuple(v) for k, v in json.loads(str(g["lm_state_shapes"])).items()}
        sd.update(recipe_state_dict(shapes, int(g["wseed"]), skip=()))
        lm_kw = dict(lm_weight=float(g["lm_weight"]), lm_conf=json.loads(str(g["lm_conf"])))
    orc = OnlineBeamSearchOracle(sd, dc["attention_heads"], dc["num_blocks"], int(g["beam"]), float(g["ctc_weight"]),
                                 sos=V - 1, eos=V - 1, penalty=float(g["penalty"]),
                                 disable_repetition_detection=bool(g["disable_repetition_detection"]), **lm_kw)
    enc_all = torch.from_numpy(g["enc_all"])
    calls = json.loads(str(g["calls"]))
    lens = g["enc_lens"].tolist()
    assert len(calls) == len(lens)
    pos, nbest = 0, int(g["nbest"])
    seen_events = []
    for k, (call, n) in enumerate(zip(calls, lens)):
        orc.events = []
        res = orc.forward(enc_all[pos : pos + n], is_final=(k == len(calls) - 1))[:nbest]
        pos += n
        seen_events += orc.events
        assert [e for e in orc.events] == call["events"], (k, orc.events, call["events"])
        assert len(res) == len(call["hyps"]), (k, len(res), len(call["hyps"]))
        for mine, ref in zip(res, call["hyps"]):
            assert mine["yseq"] == ref["yseq"], k
            assert abs(mine["score"] - ref["score"]) < 1e-3 + 1e-5 * abs(ref["score"])
            for kk, v in ref["scores"].items():
                assert abs(mine["scores"][kk] - v) < 2e-3 + 1e-5 * abs(v)
    assert seen_events, "golden exercises no break / end event"


@pytest.mark.parametrize("name", ["ebf_tiny_blocks", "ebf_small_5s", "bf_tiny_blocks", "bf_small_4s", "ebf_sub6_4s",
                                  "ebf_legacy_4s", "bf_learned_ave_4s", "bf_fixed_ave_4s"])
def test_ebranchformer_encoder_matches_reference(name):
    """E-Branchformer (attention + cgMLP branches, depthwise-conv merge) — oracle vs
    the reference's `ESPnetASRModel.encode` with encoder=e_branchformer, incl. per-block outputs."""
    from oracle import ebranchformer as oe

    g = load_golden(name)
    sd = golden_state_dict(g)
    hp = hparams(g)
    speech, lens = golden_speech(g)
    enc, olens = oe.encode(sd, speech, lens, hp["heads"], hp["num_blocks"], hp["n_fft"], hp["win_length"], hp["hop"],
                           rel_pos_type=g["config"]["encoder_conf"].get("rel_pos_type", "latest"),
                           cgmlp_weight=g["config"]["encoder_conf"].get("cgmlp_weight", 0.5))
    assert olens.tolist() == g["enc_olens"].tolist()
    ke = int(g["enc_keep_every"])
    np.testing.assert_allclose(enc[:, ::ke].numpy(), g["enc_out"], atol=5e-4, rtol=0)
    if "block_outs" in g:
        feats, flens = oc.frontend_feats(speech, lens, sd["frontend.logmel.melmat"], hp["n_fft"], hp["win_length"],
                                         hp["hop"])
        feats = oc.utterance_mvn(feats, flens)
        _, _, blocks = oe.ebranchformer_encoder(sd, feats, flens, hp["heads"], hp["num_blocks"], return_blocks=True,
                                                cgmlp_weight=g["config"]["encoder_conf"].get("cgmlp_weight", 0.5))
        for i, b in enumerate(blocks):
            np.testing.assert_allclose(b.numpy(), g["block_outs"][i], atol=1e-4, rtol=0)
    ids = oc.ctc_argmax(sd, enc).numpy()
    diff = ids != g["ctc_ids"]
    assert (g["ctc_margin"][diff] < 1e-4).all() and diff.mean() < 0.01


def test_lockstep_noisy_searches_equal_separate_ones():
    """`beam_search(noise_seeds=[...])` - several perturbed searches of one utterance run in lock step (the path-noise
    model of tests/test_gpu_fullsize.py and bench.py) - returns, group by group, exactly what separate calls with
    noise_seed = s return: same hypotheses, same scores, groups ending at different steps."""
    from oracle import beam_search as ob
    from oracle import conformer as oc

    g = load_golden("tiny_beam5")
    sd = golden_state_dict(g)
    hp = hparams(g)
    speech, lens = golden_speech(g)
    with torch.no_grad():
        enc, olens = oc.encode(sd, speech, lens, hp["heads"], hp["num_blocks"], hp["n_fft"], hp["win_length"], hp["hop"])
    e = enc[0, : int(olens[0])]
    dc = g["config"]["decoder_conf"]
    V = int(g["vocab"])
    seeds = [0, 1, 2, 5]
    with torch.no_grad():
        multi = ob.beam_search(sd, e, dc["attention_heads"], dc["num_blocks"], 5, 0.3, sos=V - 1, eos=V - 1, noise=0.05,
                               noise_seeds=seeds)
        for s, m in zip(seeds, multi):
            one = ob.beam_search(sd, e, dc["attention_heads"], dc["num_blocks"], 5, 0.3, sos=V - 1, eos=V - 1, noise=0.05,
                                 noise_seed=s)
            assert [x["yseq"] for x in one] == [x["yseq"] for x in m]
            assert [x["score"] for x in one] == [x["score"] for x in m]
        losses, clean = ob.path_noise_losses(sd, e, dc["attention_heads"], dc["num_blocks"], 5, 0.3, V - 1, 0.05, seeds)
    assert len(losses) == len(seeds) and all(abs(x) < 50 for x in losses)
