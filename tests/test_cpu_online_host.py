"""Host control flow of BatchBeamSearchOnline (espnet_amd/nets/batch_beam_search_online.py) without a GPU.

The product class takes every number from the device; here its four device hooks (`_setup`, `_see`, `_search`,
`_commit` / `_rewind`) are replaced by the CPU oracle's scorers, so the HOST logic — block loop, repetition and
local-<eos> breaks, rewind, ended lists, assemble order — is replayed against the reference's per-call n-best
(tests/golden/stream_search_*.npz) in the CPU suite.  This is a test harness: the stub lives in tests/ only.
"""
import json

import pytest
import torch

from tests.helpers import golden_state_dict, load_golden


def make_host_under_test(g, sd):
    from espnet_amd.nets.batch_beam_search_online import BatchBeamSearchOnline, _Rows
    from oracle.beam_search_online import OnlineBeamSearchOracle
    from oracle.weights import token_list

    V = int(g["vocab"])
    cw = float(g["ctc_weight"])
    dc = g["config"]["decoder_conf"]

    class _Marker:  # stands in for scorer objects: the stubbed hooks never touch them
        pass

    class HostOnly(BatchBeamSearchOnline):
        def _setup(self, x):
            self._dev = {}
            self.encbuffer = torch.empty(self.max_frames, x.size(-1))
            self.orc = OnlineBeamSearchOracle(sd, dc["attention_heads"], dc["num_blocks"], self.beam_size, cw,
                                              sos=V - 1, eos=V - 1, penalty=float(g["penalty"]))
            self.o_run = self.o_prev = self.o_best = None

        def _see(self, T):
            h = self.encbuffer[:T]
            if self.o_run is None:
                self.orc._set_memory(h)
                self.o_run = self.orc._init_hyp(h)
            self.orc._extend(h, self.o_run)
            self._visible = T

        def _search(self):
            self.n_steps += 1
            best = self.o_best = self.orc._search(self.o_run)
            keys = self._keys()
            return _Rows(list(range(len(best))), [y.tolist() for y in best.yseq], [float(s) for s in best.score],
                         [{k: float(best.scores[k][j]) for k in keys} for j in range(len(best))])

        def _commit(self):
            self.o_prev = self.o_run
            keep = torch.nonzero(self.o_best.yseq[:, -1] != self.eos).view(-1)
            self.o_run = self.o_best.batch_select(keep)

        def _rewind(self):
            self.o_run = self.o_prev

    scorers = dict(decoder=_Marker(), ctc=_Marker(), length_bonus=_Marker())
    weights = dict(decoder=1.0 - cw, ctc=cw, lm=0.0, length_bonus=float(g["penalty"]))
    from espnet_amd.nets.scorers.ctc import CTCPrefixScorer

    scorers["ctc"] = CTCPrefixScorer(ctc=None, eos=V - 1)  # only its type matters (partial scorer -> pre-beam)
    return HostOnly(beam_size=int(g["beam"]), weights=weights, scorers=scorers, sos=V - 1, eos=V - 1, vocab_size=V,
                    token_list=token_list(V), pre_beam_score_key=None if cw == 1.0 else "full",
                    disable_repetition_detection=bool(g["disable_repetition_detection"]), max_frames=256)


@pytest.mark.parametrize("name", ["stream_search_a", "stream_search_b", "stream_search_c"])
def test_online_host_control_flow_matches_reference_per_call(name):
    g = load_golden(name)
    sd = golden_state_dict(g)
    bs = make_host_under_test(g, sd)
    enc_all = torch.from_numpy(g["enc_all"])
    calls = json.loads(str(g["calls"]))
    lens = g["enc_lens"].tolist()
    pos, nbest = 0, int(g["nbest"])
    for k, (call, n) in enumerate(zip(calls, lens)):
        bs.events = []
        res = bs(enc_all[pos : pos + n], is_final=(k == len(calls) - 1))[:nbest]
        pos += n
        assert bs.events == call["events"], (k, bs.events, call["events"])
        assert len(res) == len(call["hyps"]), (k, len(res), len(call["hyps"]))
        for mine, ref in zip(res, call["hyps"]):
            assert mine.yseq.tolist() == ref["yseq"], k
            assert abs(float(mine.score) - ref["score"]) < 1e-3 + 1e-5 * abs(ref["score"])
            assert set(mine.scores) == set(ref["scores"])
    bs.reset()
    assert bs.running is None and bs.n_enc == 0 and bs.ended_hyps == []


@pytest.mark.parametrize("name", ["stream_frontend_gmvn", "stream_frontend_umvn"])
def test_streaming_apply_frontend_host_logic_on_cpu(name):
    """Speech2TextStreaming.apply_frontend's HOST side (waveform overlap buffer, residual samples, trimming of the
    frames that see artificial chunk edges, asr_inference_streaming.py:205-293) with the feature extraction
    stubbed by the CPU oracle: per-call frame counts and values equal the reference's.  (The same test with the
    HIP frontend is tests/test_gpu_streaming.py.)"""
    import types

    import numpy as np

    from espnet_amd.bin.asr_inference_streaming import Speech2TextStreaming
    from espnet_amd.nets_utils import stft_frame_lengths
    from oracle import conformer as oc
    from oracle.weights import synth_waveform
    from tests.helpers import GOLDEN

    z = np.load(GOLDEN / f"{name}.npz")
    fc = json.loads(str(z["frontend_conf"]))
    mel = torch.from_numpy(z["melmat"])

    class FE:
        def feature_lengths(self, ns):
            return stft_frame_lengths(ns, fc["n_fft"], fc["hop_length"])

        def forward_device(self, wav, flens_dev, wlens_dev=None):
            f, _ = oc.frontend_feats(wav, torch.tensor([wav.size(1)]), mel, fc["n_fft"], fc["win_length"],
                                     fc["hop_length"])
            return f

    class Norm:
        def forward_device(self, feats, flens_dev):
            if bool(z["use_global_mvn"]):
                return oc.global_mvn(feats, flens_dev.long(), torch.from_numpy(z["gmvn_mean"]),
                                     torch.from_numpy(z["gmvn_std"]))
            return oc.utterance_mvn(feats, flens_dev.long())

    s2t = object.__new__(Speech2TextStreaming)
    s2t.asr_model = types.SimpleNamespace(frontend=FE(), normalize=Norm())
    s2t.device, s2t.n_fft, s2t.hop_length, s2t.win_length = "cpu", fc["n_fft"], fc["hop_length"], fc["win_length"]
    n, cs = int(z["n_samples"]), int(z["chunk_samples"])
    wav = synth_waveform(int(z["utt_id"]), n)
    feats, lens, state, pos = [], [], None, 0
    while pos < n:
        nxt = min(n, pos + cs)
        f, fl, state = s2t.apply_frontend(wav[pos:nxt], state, is_final=(nxt == n))
        lens.append(-1 if f is None else int(f.size(1)))
        if f is not None:
            assert int(fl[0]) == f.size(1)
            feats.append(f[0])
        pos = nxt
    assert lens == z["feat_lens"].tolist()
    np.testing.assert_allclose(torch.cat(feats, 0).numpy(), z["feats"], atol=3e-4, rtol=0)
