"""ORACLE (test infrastructure, never shipped or measured): Slaney mel filterbank.

Restates `librosa.filters.mel` (librosa>=0.10.2, an un-vendored third-party dependency of the
reference: /root/reference/pyproject.toml:40) as called from
/root/reference/espnet2/layers/log_mel.py:38-52 with
`sr=fs, n_fft, n_mels, fmin=0, fmax=fs/2, htk=False` (=> Slaney scale, norm="slaney", float32).

PARITY UNPINNED against librosa itself for the matrix values: no reference test pins them and librosa
is not installed here.  What does pin them (tests/test_oracle_golden.py): an INDEPENDENT implementation of
the same published algorithm that ships in this image — `transformers.audio_utils.mel_filter_bank(norm=
"slaney", mel_scale="slaney")`, which its own project tests against librosa — agrees to float32 round-off
(1.6e-9) with identical supports on four configurations, and the value librosa's documentation prints for
`librosa.filters.mel(sr=22050, n_fft=2048)[0, 1]` (0.016) is reproduced.  Mitigation beyond that: the matrix
is a persistent buffer (`frontend.logmel.melmat`) in every reference checkpoint and the HIP frontend takes it
as an INPUT, so both sides of every parity test use the very same matrix.

Published algorithm (librosa/core/convert.py `hz_to_mel`/`mel_to_hz`, librosa/filters.py `mel`):
  * Slaney mel scale: linear 200/3 Hz per mel below 1 kHz, logarithmic above with
    step ln(6.4)/27 per mel;
  * n_mels+2 band edges equally spaced in mel between fmin and fmax;
  * triangular ramps over the rFFT bin centre frequencies;
  * area normalisation 2 / (f[i+2] - f[i]).
"""
import numpy as np


def _hz_to_mel(freq, htk=False):
    freq = np.asanyarray(freq, dtype=np.float64)
    if htk:
        return 2595.0 * np.log10(1.0 + freq / 700.0)
    f_min, f_sp = 0.0, 200.0 / 3
    mels = (freq - f_min) / f_sp
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if freq.ndim:
        log_t = freq >= min_log_hz
        mels[log_t] = min_log_mel + np.log(freq[log_t] / min_log_hz) / logstep
    elif freq >= min_log_hz:
        mels = min_log_mel + np.log(freq / min_log_hz) / logstep
    return mels


def _mel_to_hz(mels, htk=False):
    mels = np.asanyarray(mels, dtype=np.float64)
    if htk:
        return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    f_min, f_sp = 0.0, 200.0 / 3
    freqs = f_min + f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if mels.ndim:
        log_t = mels >= min_log_mel
        freqs[log_t] = min_log_hz * np.exp(logstep * (mels[log_t] - min_log_mel))
    elif mels >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (mels - min_log_mel))
    return freqs


def slaney_mel_filterbank(sr, n_fft, n_mels=80, fmin=0.0, fmax=None, htk=False):
    """Return the (n_mels, 1 + n_fft//2) float32 filterbank (librosa orientation).

    The reference stores its transpose: `melmat = mel(...).T` (log_mel.py:50-52).
    """
    if fmax is None:
        fmax = float(sr) / 2
    n_mels = int(n_mels)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)), dtype=np.float32)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    min_mel = _hz_to_mel(fmin, htk=htk)
    max_mel = _hz_to_mel(fmax, htk=htk)
    mel_f = _mel_to_hz(np.linspace(min_mel, max_mel, n_mels + 2), htk=htk)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights
