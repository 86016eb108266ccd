"""Shared test helpers: load a golden fixture and rebuild the recipe weights it was made with."""
import json
from pathlib import Path

import numpy as np
import torch
import yaml

from oracle.weights import recipe_state_dict, synth_waveform

GOLDEN = Path(__file__).resolve().parent / "golden"


def load_golden(name):
    z = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    g = {k: z[k] for k in z.files}
    g["config"] = yaml.safe_load(str(g["config_yaml"]))
    g["shapes"] = {k: tuple(v) for k, v in json.loads(str(g["state_shapes"])).items()}
    return g


def golden_state_dict(g):
    sd = recipe_state_dict(g["shapes"], int(g["wseed"]))
    sd["frontend.logmel.melmat"] = torch.from_numpy(g["melmat"]).clone()
    if "tweaks" in g:  # e.g. a biased <eos> logit (tests/golden/make_golden.py)
        for key, idx, delta in json.loads(str(g["tweaks"])):
            sd[key][idx] += delta
    if "ctc_rows" in g:  # a fitted (peaked) CTC head: make_golden.py::fit_peaked_ctc_head
        rows = torch.from_numpy(np.asarray(g["ctc_rows"]))
        sd["ctc.ctc_lo.weight"] = torch.zeros_like(sd["ctc.ctc_lo.weight"])
        sd["ctc.ctc_lo.bias"] = torch.zeros_like(sd["ctc.ctc_lo.bias"])
        sd["ctc.ctc_lo.weight"][rows] = torch.from_numpy(np.asarray(g["ctc_w_rows"]))
        sd["ctc.ctc_lo.bias"][rows] = torch.from_numpy(np.asarray(g["ctc_b_rows"]))
    for key, pre in (("ctc.ctc_lo", "ctcov"), ("decoder.output_layer", "decov")):
        if pre + "_rows" in g:  # fitted rows laid over the recipe weights: make_golden.py::fit_peaked_search_heads
            rows = torch.from_numpy(np.asarray(g[pre + "_rows"]))
            sd[key + ".weight"][rows] = torch.from_numpy(np.asarray(g[pre + "_w"]))
            sd[key + ".bias"][rows] = torch.from_numpy(np.asarray(g[pre + "_b"]))
    return sd


def golden_speech(g):
    if "utt_ids" in g:
        ids, lens = g["utt_ids"].tolist(), g["lengths"].tolist()
    else:
        ids, lens = [int(g["utt_id"])], [int(g["n_samples"])]
    speech = torch.zeros(len(ids), max(lens))
    for i, (u, n) in enumerate(zip(ids, lens)):
        speech[i, :n] = synth_waveform(u, n)
    return speech, torch.tensor(lens, dtype=torch.long)


def hparams(g):
    c = g["config"]
    fc = c.get("frontend_conf") or {}
    return dict(
        heads=c["encoder_conf"]["attention_heads"],
        num_blocks=c["encoder_conf"]["num_blocks"],
        n_fft=fc.get("n_fft", 512),
        win_length=fc.get("win_length", None),
        hop=fc.get("hop_length", 128),
    )


def stream_feats(utt_id, n_samples):
    """Encoder input of the streaming fixtures (same recipe as tests/golden/make_golden.py)."""
    from oracle import conformer as oc
    from oracle.mel import slaney_mel_filterbank

    mel = torch.from_numpy(slaney_mel_filterbank(sr=16000, n_fft=512, n_mels=80, fmin=0, fmax=8000).T.copy())
    wav = synth_waveform(utt_id, n_samples)
    f, fl = oc.frontend_feats(wav[None], torch.tensor([n_samples]), mel, 512, 512, 128)
    return oc.utterance_mvn(f, fl)[0]


def load_stream_golden(name):
    z = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    g = {k: z[k] for k in z.files}
    g["conf"] = json.loads(str(g["enc_conf"]))
    g["shapes"] = {k: tuple(v) for k, v in json.loads(str(g["state_shapes"])).items()}
    g["sd"] = recipe_state_dict(g["shapes"], int(g["wseed"]), skip=())
    return g


def oracle_rescore(sd, enc, yseq, heads, num_blocks, ctc_weight, eos, maxlen=None):
    """Teacher-forced score of ONE given hypothesis under the oracle's scorers (oracle/beam_search.py:
    DecoderOracle.step and CtcPrefixScorer.score, the same calls `beam_search` makes, only along a fixed
    token path instead of inside the pruned search).  enc (T, d) valid frames; yseq incl. sos and eos.
    Returns dict(decoder, ctc, score).  The <eos> the search force-appends at maxlen carries no score
    (batch_beam_search.py:393-410), so at most `maxlen` (= T for maxlenratio 0) tokens are scored."""
    import torch.nn.functional as F

    from oracle import beam_search as ob

    T = enc.size(0)
    maxlen = T if maxlen is None else maxlen
    n_scored = min(len(yseq) - 1, maxlen)
    dec = ob.DecoderOracle(sd, enc, heads, num_blocks, maxlen)
    cache = dec.init_cache()
    logp = F.log_softmax(F.linear(enc, sd["ctc.ctc_lo.weight"], sd["ctc.ctc_lo.bias"]), dim=-1)
    ctc = ob.CtcPrefixScorer(logp, eos)
    r_prev, s_prev = ctc.initial_state()
    s_dec = s_ctc = 0.0
    with torch.no_grad():
        for i in range(n_scored):
            last = torch.tensor([yseq[i]], dtype=torch.long)
            nxt = int(yseq[i + 1])
            lp, cache = dec.step(last, i, cache)
            s_dec += float(lp[0, nxt])
            delta, r_new, log_psi = ctc.score(i, last, r_prev, s_prev, torch.tensor([[nxt]]))
            s_ctc += float(delta[0, nxt])
            r_prev, s_prev = r_new[:, :, :, 0], log_psi[:, nxt]
    return {"decoder": s_dec, "ctc": s_ctc, "score": (1.0 - ctc_weight) * s_dec + ctc_weight * s_ctc}


def oracle_rescore_batch(sd, enc, yseqs, heads, num_blocks, ctc_weight, eos, maxlen=None, lm_conf=None):
    """oracle.beam_search.rescore_batch (kept under this name for the tests)."""
    from oracle.beam_search import rescore_batch

    return rescore_batch(sd, enc, yseqs, heads, num_blocks, ctc_weight, eos, maxlen=maxlen, lm_conf=lm_conf)
