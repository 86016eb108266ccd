"""GPU parity of the block-synchronous streaming beam search through the
C-ABI (em_search_init / em_search_online_extend / _core / _commit / _rewind).

tests/golden/stream_search_*.npz: the reference's `Speech2TextStreaming` fed 640 ms chunks; per call the
encoder frames it handed to `BatchBeamSearchOnline.forward` and the n-best that came back, plus the
reference's own log markers (repetition / local <eos> / end detection / forced <eos>)."""
import json
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import golden_state_dict, load_golden  # noqa: E402
from tests.test_gpu_search import _sub  # noqa: E402

CASES = ["stream_search_a", "stream_search_b", "stream_search_c", "stream_search_lm", "stream_search_rnnlm",
         "stream_search_gru", "stream_search_peaked"]


def build_online(g, sd, dtype="float32"):
    from espnet_amd.asr.ctc import CTC
    from espnet_amd.asr.decoder.transformer_decoder import TransformerDecoder
    from espnet_amd.nets.batch_beam_search_online import BatchBeamSearchOnline
    from espnet_amd.nets.scorers.ctc import CTCPrefixScorer
    from espnet_amd.nets.scorers.length_bonus import LengthBonus
    from oracle.weights import token_list

    V = int(g["vocab"])
    d = g["config"]["encoder_conf"]["output_size"]
    dec = TransformerDecoder(V, d, compute_dtype=dtype, **g["config"]["decoder_conf"])
    dec.load_state_dict(_sub(sd, "decoder."), strict=True)
    ctc = CTC(V, d, compute_dtype=dtype)
    ctc.load_state_dict(_sub(sd, "ctc."), strict=True)
    dec.cuda(), ctc.cuda()
    cw = float(g["ctc_weight"])
    # asr_inference_streaming.py:88-136
    scorers = dict(decoder=dec, ctc=CTCPrefixScorer(ctc=ctc, eos=V - 1), length_bonus=LengthBonus(V))
    weights = dict(decoder=1.0 - cw, ctc=cw, lm=0.0, length_bonus=float(g["penalty"]))
    if "lm_conf" in g and json.loads(str(g["lm_conf"])) is not None:  # asr_inference_streaming.py:97-102
        from tests.test_gpu_search import build_lm

        scorers["lm"] = build_lm(g, dtype)
        weights["lm"] = float(g["lm_weight"])
    return BatchBeamSearchOnline(beam_size=int(g["beam"]), weights=weights, scorers=scorers, sos=V - 1, eos=V - 1,
                                 vocab_size=V, token_list=token_list(V),
                                 pre_beam_score_key=None if cw == 1.0 else "full",
                                 disable_repetition_detection=bool(g["disable_repetition_detection"]),
                                 max_frames=256)


@pytest.mark.parametrize("name", CASES)
def test_online_search_f32_matches_reference_per_call(name):
    g = load_golden(name)
    sd = golden_state_dict(g)
    bs = build_online(g, sd, "float32")
    enc_all = torch.from_numpy(g["enc_all"]).cuda()
    calls = json.loads(str(g["calls"]))
    lens = g["enc_lens"].tolist()
    pos, nbest = 0, int(g["nbest"])
    n_checked = 0
    for k, (call, n) in enumerate(zip(calls, lens)):
        bs.events = []
        res = bs(enc_all[pos : pos + n], is_final=(k == len(calls) - 1))[:nbest]
        pos += n
        assert bs.events == call["events"], (k, bs.events, call["events"])
        assert len(res) == len(call["hyps"]), (k, len(res), len(call["hyps"]))
        for mine, ref in zip(res, call["hyps"]):
            assert mine.yseq.tolist() == ref["yseq"], k
            # fp32 round-off of a different summation order, accumulated over up to ~100 steps
            assert abs(float(mine.score) - ref["score"]) < 5e-3 + 2e-5 * abs(ref["score"])
            for kk, v in ref["scores"].items():
                assert abs(float(mine.scores[kk]) - v) < 5e-3 + 2e-5 * abs(v)
            n_checked += 1
    assert n_checked > 0


def test_online_search_restarts_cleanly_and_bf16_runs():
    """reset() between utterances (asr_inference_streaming.py:199-203) gives the same result again; the bf16
    mode runs the same control flow (near-tie flips are expected with random weights: structure only)."""
    g = load_golden("stream_search_b")
    sd = golden_state_dict(g)
    enc_all = torch.from_numpy(g["enc_all"]).cuda()
    lens = g["enc_lens"].tolist()

    def run(bs):
        pos, out = 0, None
        for k, n in enumerate(lens):
            out = bs(enc_all[pos : pos + n], is_final=(k == len(lens) - 1))
            pos += n
        bs.reset()
        return out

    bs = build_online(g, sd, "float32")
    a, b = run(bs), run(bs)
    assert [h.yseq.tolist() for h in a] == [h.yseq.tolist() for h in b]
    assert [float(h.score) for h in a] == [float(h.score) for h in b]
    V = int(g["vocab"])
    out = run(build_online(g, sd, "bfloat16"))
    assert len(out) >= 1
    # flat random-init posteriors: the bf16 best is held to the reference's best per token (the exact n-best is asked
    # of the peaked fixture below)
    ref_best = json.loads(str(g["calls"]))[-1]["hyps"][0]
    assert abs(float(out[0].score) - ref_best["score"]) <= BF16_ONLINE_EPS * (len(ref_best["yseq"]) - 1)
    for h in out:
        y = h.yseq.tolist()
        assert y[0] == V - 1 and y[-1] == V - 1
        tot = sum(float(v) * {"decoder": 1 - float(g["ctc_weight"]), "ctc": float(g["ctc_weight"]),
                              "length_bonus": float(g["penalty"])}[k] for k, v in h.scores.items())
        assert abs(tot - float(h.score)) < 2e-2 + 1e-4 * abs(tot)


# per scored token: |bf16 device score - f32 reference score| along the same token sequence (measured on MI355X, round 4:
# 1.2e-3 on the peaked fixture; d = 64 / 128 models, so relatively coarser than the 512-wide offline fixtures)
BF16_ONLINE_EPS = 5e-3


def test_online_search_bf16_peaked_head_exact():
    """The streaming search in bfloat16 on PEAKED posteriors (tests/golden/make_golden.py: stream_search_peaked - both
    heads fitted to a transcript; qualified against N(0, 0.05^2) noise on every log-probability): at every call the
    separated head of the reference's n-best (`sep_counts`: the leading hypotheses up to the first gap below 0.5) must
    come back exactly - same token sequences, same order - with the reference's break / end events, scores within
    BF16_ONLINE_EPS per token."""
    g = load_golden("stream_search_peaked")
    sd = golden_state_dict(g)
    bs = build_online(g, sd, "bfloat16")
    enc_all = torch.from_numpy(g["enc_all"]).cuda()
    calls = json.loads(str(g["calls"]))
    lens, sep = g["enc_lens"].tolist(), g["sep_counts"].tolist()
    pos, n_checked, worst = 0, 0, 0.0
    for k, (call, n) in enumerate(zip(calls, lens)):
        bs.events = []
        res = bs(enc_all[pos : pos + n], is_final=(k == len(calls) - 1))
        pos += n
        assert bs.events == call["events"], (k, bs.events, call["events"])
        for mine, ref in zip(res[: sep[k]], call["hyps"][: sep[k]]):
            assert mine.yseq.tolist() == ref["yseq"], k
            err = abs(float(mine.score) - ref["score"]) / (len(ref["yseq"]) - 1)
            worst = max(worst, err)
            assert err <= BF16_ONLINE_EPS, (k, float(mine.score), ref["score"])
            n_checked += 1
    print(f"[stream_search_peaked bf16] {n_checked} hypotheses exact over {len(calls)} calls; worst score error per "
          f"token {worst:.2e} (bound {BF16_ONLINE_EPS:.0e})")
    assert n_checked >= 3


@pytest.mark.parametrize("name", ["stream_search_b", "stream_search_lm"])
def test_online_search_graph_replay_equals_eager(name):
    """The label step replayed as a captured hipGraph (step index in device memory, one graph per number of visible
    frames) gives bit for bit what the eager launch sequence gives - tokens, scores, per-scorer scores, per call - over
    three utterances on the same search object: the first runs eagerly, graphs are captured on the way, the third is
    served by replays (including after the one-step rewind at the end of a block)."""
    g = load_golden(name)
    sd = golden_state_dict(g)
    enc_all = torch.from_numpy(g["enc_all"]).cuda()
    lens = g["enc_lens"].tolist()

    def run(bs):
        pos, outs = 0, []
        for k, n in enumerate(lens):
            res = bs(enc_all[pos : pos + n], is_final=(k == len(lens) - 1))
            outs.append([(h.yseq.tolist(), float(h.score), {kk: float(v) for kk, v in h.scores.items()}) for h in res])
            pos += n
        bs.reset()
        return outs

    eager = build_online(g, sd, "float32")
    eager.use_hipgraph = False
    ref = run(eager)
    assert eager.n_replays == 0
    bs = build_online(g, sd, "float32")
    bs.use_hipgraph = True  # (off by default: measured no faster)
    bs.graph_after = 0  # capture at the first use of a (buffers, T) key
    first = run(bs)
    n0 = bs.n_replays
    second = run(bs)
    assert first == ref and second == ref
    assert bs.n_replays - n0 == bs.n_steps // 2, (bs.n_replays, n0, bs.n_steps)  # the second utterance: replays only


@pytest.mark.parametrize("name", ["stream_search_a", "stream_search_b"])
def test_speech2text_streaming_online_end_to_end_f32(name, tmp_path):
    """Waveform chunks -> HIP frontend -> HIP contextual-block encoder -> device online search, behind the
    reference's Speech2TextStreaming API.  The encoder runs on the GPU here, so its ~1e-4 activation
    differences may move scores by ~1e-2: per call the same events and hypothesis count, scores within
    that band, and identical token sequences wherever the reference's own top-2 gap is clear."""
    from espnet_amd.bin.asr_inference_streaming import Speech2TextStreaming
    from oracle.weights import synth_waveform

    g = load_golden(name)
    sd = golden_state_dict(g)
    (tmp_path / "config.yaml").write_text(str(g["config_yaml"]))
    torch.save(sd, tmp_path / "model.pth")
    s2t = Speech2TextStreaming(str(tmp_path / "config.yaml"), str(tmp_path / "model.pth"), device="cuda",
                               dtype="float32", beam_size=int(g["beam"]), ctc_weight=float(g["ctc_weight"]),
                               penalty=float(g["penalty"]), nbest=int(g["nbest"]),
                               disable_repetition_detection=bool(g["disable_repetition_detection"]))
    assert s2t.search == "online"
    s2t.beam_search.max_frames = 256
    n, chunk = int(g["n_samples"]), int(g["chunk_samples"])
    wav = synth_waveform(int(g["utt_id"]), n)
    calls = json.loads(str(g["calls"]))
    pos, k, same_tokens = 0, 0, 0
    while pos < n:
        nxt = min(n, pos + chunk)
        s2t.beam_search.events = []
        res = s2t(wav[pos:nxt].numpy(), is_final=(nxt == n))
        ref = calls[k]
        if nxt < n:
            assert s2t.beam_search.events == ref["events"], (k, s2t.beam_search.events, ref["events"])
        assert len(res) == len(ref["hyps"]), (k, len(res), len(ref["hyps"]))
        if ref["hyps"]:
            text, token, token_int, hyp = res[0]
            r0 = ref["hyps"][0]
            assert abs(float(hyp.score) - r0["score"]) < 5e-2 + 1e-4 * abs(r0["score"]), (k, float(hyp.score), r0["score"])
            lower = [h["score"] for h in ref["hyps"] if h["score"] < r0["score"] - 1e-6]  # duplicates aside
            gap = r0["score"] - lower[0] if lower else 1.0
            if gap > 0.1:
                assert hyp.yseq.tolist() == r0["yseq"], k
                same_tokens += 1
        pos, k = nxt, k + 1
    assert k == len(calls) and same_tokens >= 1
