"""Minimal stand-in for `soundfile` (libsndfile is not in this image) so the reference's `sound`
data type runs for the decode-CLI golden: 16-bit PCM wav through the stdlib `wave` module, samples
scaled by 1/32768 exactly as libsndfile does for float reads.  Only what
espnet2/train/iterable_dataset.py:45 calls (`soundfile.read(path)[0]`)."""
import wave

import numpy as np


def read(path, dtype="float64", always_2d=False):
    with wave.open(str(path), "rb") as w:
        assert w.getsampwidth() == 2, "shim reads 16-bit PCM only"
        nch, rate = w.getnchannels(), w.getframerate()
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    x = pcm.astype(np.float64) / 32768.0
    if nch > 1 or always_2d:
        x = x.reshape(-1, nch)
    return x.astype(dtype), rate
