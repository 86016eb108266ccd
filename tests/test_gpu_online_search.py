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

CASES = ["stream_search_a", "stream_search_b", "stream_search_c"]


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
    for h in out:
        y = h.yseq.tolist()
        assert y[0] == V - 1 and y[-1] == V - 1
        tot = sum(float(v) * {"decoder": 1 - float(g["ctc_weight"]), "ctc": float(g["ctc_weight"]),
                              "length_bonus": float(g["penalty"])}[k] for k, v in h.scores.items())
        assert abs(tot - float(h.score)) < 2e-2 + 1e-4 * abs(tot)
