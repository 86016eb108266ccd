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
