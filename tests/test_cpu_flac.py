"""FLAC reading of the decode CLI (the recipes' default `audio_format=flac`, egs2/TEMPLATE/asr1/asr.sh:56; the
reference reads it through soundfile/libsndfile, espnet2/fileio/sound_scp.py).  Two decoders written from the format
specification (RFC 9639) are under test — the Python one (fileio/sound_scp.py:read_flac, every stream) and the native
one (csrc/host_io.cpp behind em_wav_probe / em_wav_load_rows, mono streams of the batched fast path) — against
  * the specification's own self-checking example stream (CRC-8, CRC-16 and the MD5 of the decoded audio in
    STREAMINFO must all verify),
  * streams from the independent test encoder tests/flac_writer.py covering every subframe type, Rice / Rice2 /
    escaped residual partitions, wasted bits, all channel assignments, block-size and sample-size codes, and
  * each other, bit for bit."""
import hashlib
import struct

import numpy as np
import pytest
import torch

from espnet_amd.fileio.sound_scp import WavBatchReader, read_flac, read_wav, write_wav_pcm16
from espnet_amd.train.iterable_dataset import IterableESPnetDataset, StreamingBatchIterator
from tests.flac_writer import encode_flac, lpc_coefficients

# RFC 9639 appendix D.1: the smallest complete stream (1 stereo 16-bit sample, verbatim subframes with wasted bits)
RFC_EXAMPLE_1 = bytes.fromhex(
    "664c614380000022100010000000 0f00000f0ac442f0000000013e84b41807dc690307586a3dad1a2e0f"
    "fff8691800 00bf0358fd03128baa9a".replace(" ", ""))


def _speechlike(n, seed, bits=16, amp=0.3):
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    x = sum(a * np.sin(2 * np.pi * f * t / 16000 + p) for a, f, p in
            [(0.5, 180, 0.3), (0.3, 410, 1.1), (0.15, 1290, 2.0), (0.05, 3100, 0.7)])
    x = amp * x * (0.6 + 0.4 * np.sin(2 * np.pi * t / 5000)) + 0.004 * rng.standard_normal(n)
    lim = 1 << (bits - 1)
    return np.clip(np.rint(x * lim), -lim, lim - 1).astype(np.int64)


def _native(path, threads=2):
    rd = WavBatchReader(threads)
    probed = rd.probe([str(path)])
    if probed is None:
        return None
    out, lens = rd.load(probed, [0])
    return out[0].numpy()[: lens[0]]


def test_specification_example_stream_verifies_crcs_and_md5(tmp_path):
    p = tmp_path / "rfc1.flac"
    p.write_bytes(RFC_EXAMPLE_1)
    x, rate = read_flac(p, always_2d=True, native=False)  # raises on any CRC mismatch
    assert rate == 44100 and x.shape == (1, 2)
    pcm = np.rint(x * 32768).astype("<i2")
    assert pcm.tolist() == [[25588, 10416]]
    assert hashlib.md5(pcm.tobytes()).digest() == RFC_EXAMPLE_1[26:42]  # MD5 of the audio, from STREAMINFO
    assert _native(p) is None  # stereo: left to the Python reader
    # flipping one payload bit must be caught by the frame CRC
    bad = bytearray(RFC_EXAMPLE_1)
    bad[-3] ^= 0x10  # a sample bit of the second subframe
    p.write_bytes(bytes(bad))
    with pytest.raises(RuntimeError, match="CRC"):
        read_flac(p)


def _plans(x, bits):
    """(name, blocksize, plan) covering the format's coding tools on one mono signal."""
    def fixed(order, po=0, method=0, escape=False):
        return lambda fi, ci, s, bps: (dict(kind="fixed", order=min(order, len(s)), po=po if len(s) % (1 << po) == 0
                                            and (len(s) >> po) >= order else 0, method=method, escape=escape)
                                       if len(s) > 4 else dict(kind="verbatim"))

    def lpc(order, prec, po=2, method=0):
        def plan(fi, ci, s, bps):
            if len(s) <= order or len(s) % (1 << po) or (len(s) >> po) < order:
                return dict(kind="verbatim")
            c, sh = lpc_coefficients(s, order, prec)
            return dict(kind="lpc", order=order, coefs=c, prec=prec, shift=sh, po=po, method=method)
        return plan

    def mixed(fi, ci, s, bps):  # what a real encoder does: a different tool per frame
        kinds = [dict(kind="verbatim"), fixed(2, 3)(fi, ci, s, bps), lpc(8, 12, 3)(fi, ci, s, bps),
                 fixed(4, 1, 1)(fi, ci, s, bps), lpc(12, 15, 4)(fi, ci, s, bps), fixed(1, 2, 0, True)(fi, ci, s, bps)]
        return kinds[fi % len(kinds)]

    out = [("verbatim_4096", 4096, None), ("fixed0_192", 192, fixed(0)), ("fixed1_576", 576, fixed(1, 2)),
           ("fixed2_1152", 1152, fixed(2, 3)), ("fixed3_256", 256, fixed(3, 4)), ("fixed4_1000", 1000, fixed(4, 3)),
           ("fixed2_rice2", 512, fixed(2, 2, method=1)), ("fixed2_escape", 512, fixed(2, 3, escape=True)),
           ("lpc1", 1024, lpc(1, 8)), ("lpc8_p12", 4096, lpc(8, 12, 4)), ("lpc12_p15", 2304, lpc(12, 15, 3)),
           ("lpc32_p14", 4608, lpc(32, 14, 5)), ("lpc8_rice2", 2048, lpc(8, 13, 3, method=1)),
           ("tiny_blocks_many_frames", 16, fixed(2)), ("mixed_4096", 4096, mixed), ("mixed_300", 300, mixed)]
    return out


@pytest.mark.parametrize("bits", [16, 24, 8, 12, 20])
def test_every_coding_tool_round_trips_through_both_decoders(tmp_path, bits):
    n = 9000 if bits == 16 else 5000
    x = _speechlike(n, bits, bits)
    want = (x.astype(np.float64) / (1 << (bits - 1))).astype(np.float32)
    assert np.array_equal(want.astype(np.float64) * (1 << (bits - 1)), x)  # <= 24 bits: float32 holds them exactly
    for name, bs, plan in _plans(x, bits):
        p = tmp_path / f"{name}.flac"
        p.write_bytes(encode_flac([x], bits, 16000, bs, plan))
        py, rate = read_flac(p, dtype="float32", native=False)
        assert rate == 16000 and py.dtype == np.float32 and py.shape == (n,), name
        assert np.array_equal(py.view(np.uint32), want.view(np.uint32)), name
        nat = _native(p)
        assert nat is not None and np.array_equal(nat.view(np.uint32), want.view(np.uint32)), name
        y64, _ = read_wav(p)  # the reference's default dtype; dispatch by magic number
        assert y64.dtype == np.float64 and np.array_equal(y64, want.astype(np.float64)), name


def test_silence_wasted_bits_noise_and_32_bit(tmp_path):
    rng = np.random.default_rng(3)
    cases = {
        "silence": (np.zeros(5000, np.int64), 16, lambda fi, ci, s, b: dict(kind="constant")),
        "dc": (np.full(3000, -1234, np.int64), 16, lambda fi, ci, s, b: dict(kind="constant")),
        # every sample a multiple of 8: 3 wasted bits, on top of a fixed predictor
        "wasted": (_speechlike(6000, 5) // 8 * 8, 16, lambda fi, ci, s, b: dict(kind="fixed", order=2, po=2)),
        # white noise at full scale: large Rice parameters (Rice2 codes them with 5 bits)
        "noise_rice2": (rng.integers(-32768, 32768, 4096), 16,
                        lambda fi, ci, s, b: dict(kind="fixed", order=1, po=3, method=1)),
        "noise_24": (rng.integers(-2 ** 23, 2 ** 23, 4096), 24,
                     lambda fi, ci, s, b: dict(kind="fixed", order=0, po=2, method=1)),
        "bits32": (rng.integers(-2 ** 31, 2 ** 31, 2048), 32, lambda fi, ci, s, b: dict(kind="verbatim")),
        "bits32_pred": (_speechlike(3000, 9, 32, amp=0.2), 32,
                        lambda fi, ci, s, b: dict(kind="fixed", order=2, po=2, method=1)),
    }
    for name, (x, bits, plan) in cases.items():
        p = tmp_path / f"{name}.flac"
        p.write_bytes(encode_flac([x], bits, 16000, 1024, plan))
        want64 = x.astype(np.float64) / (1 << (bits - 1))
        py, _ = read_flac(p, native=False)
        assert np.array_equal(py, want64), name
        nat = _native(p)
        assert nat is not None and np.array_equal(nat, want64.astype(np.float32)), name


@pytest.mark.parametrize("assignment", ["independent", "left_side", "right_side", "mid_side"])
def test_stereo_channel_assignments(tmp_path, assignment):
    left, right = _speechlike(5000, 11), _speechlike(5000, 12)
    right = (0.7 * left + 0.3 * right).astype(np.int64)  # correlated, like real stereo
    left[::7] |= 1  # odd sums: the mid/side rounding bit matters
    plan = lambda fi, ci, s, bps: dict(kind="fixed", order=2, po=2) if fi % 2 else dict(kind="verbatim")  # noqa: E731
    p = tmp_path / "st.flac"
    p.write_bytes(encode_flac([left, right], 16, 16000, 1152, plan, assignment=assignment))
    y, rate = read_flac(p, native=False)
    assert y.shape == (5000, 2)
    assert np.array_equal(y * 32768, np.stack([left, right], 1))
    assert _native(p) is None  # multi-channel windows go through the Python reader
    it = StreamingBatchIterator(IterableESPnetDataset([(_scp(tmp_path, [p]), "speech", "sound")]), batch_size=1)
    (keys, batch), = list(it)
    assert it.native_windows == 0 and batch["speech"].shape == (1, 5000, 2)


def _scp(tmp_path, files):
    scp = tmp_path / "wav.scp"
    scp.write_text("".join(f"utt{i} {f}\n" for i, f in enumerate(files)))
    return str(scp)


def test_metadata_blocks_unknown_length_and_trailers(tmp_path):
    x = _speechlike(3000, 21)
    want = (x / 32768.0).astype(np.float32)
    plan = lambda fi, ci, s, bps: dict(kind="fixed", order=2, po=1)  # noqa: E731
    vorbis = struct.pack("<I", 4) + b"test" + struct.pack("<I", 1) + struct.pack("<I", 7) + b"TITLE=x"
    blocks = [(4, vorbis), (3, b"\x00" * 18), (6, b"\x00" * 5000), (1, b"\x00" * 37)]  # comment, seek, picture, pad
    p = tmp_path / "meta.flac"
    p.write_bytes(encode_flac([x], 16, 16000, 512, plan, extra_blocks=blocks))
    assert np.array_equal(read_flac(p, dtype="float32", native=False)[0], want) and np.array_equal(_native(p), want)
    # a stream whose STREAMINFO leaves the length open, followed by an ID3v1-style tag: the Python reader decodes to
    # the end of the frames; the native reader does not guess and passes
    p.write_bytes(encode_flac([x], 16, 16000, 512, plan, total_in_header=False, trailer=b"TAG" + b"\x00" * 125))
    assert np.array_equal(read_flac(p, dtype="float32", native=False)[0], want) and _native(p) is None
    # known length + trailer: both stop after the announced samples
    p.write_bytes(encode_flac([x], 16, 16000, 512, plan, trailer=b"TAG" + b"\x00" * 125))
    assert np.array_equal(read_flac(p, dtype="float32", native=False)[0], want) and np.array_equal(_native(p), want)


def test_corruption_is_detected_never_silently_decoded(tmp_path):
    x = _speechlike(4000, 31)
    good = encode_flac([x], 16, 16000, 1024, lambda fi, ci, s, bps: dict(kind="fixed", order=2, po=2))
    p = tmp_path / "c.flac"
    rng = np.random.default_rng(0)
    rd = WavBatchReader(1)
    first_frame = good.index(b"\xff\xf8")
    caught = 0
    for trial in range(150):
        b = bytearray(good)
        if trial % 3 == 0:
            b = b[: int(rng.integers(first_frame, len(b)))]  # truncated mid stream
        else:
            pos = int(rng.integers(first_frame, len(b)))
            b[pos] ^= 1 << int(rng.integers(0, 8))
        p.write_bytes(bytes(b))
        with pytest.raises((RuntimeError, ValueError, EOFError)):
            read_flac(p, native=False)
        probed = rd.probe([str(p)])
        assert probed is not None  # the header is intact ...
        with pytest.raises(OSError):  # ... the damage is found while decoding (CRC / sync / short stream)
            rd.load(probed, [0])
        caught += 1
    assert caught == 150
    # a STREAMINFO announcing an absurd length is refused by the probe (no giant row is ever allocated)
    b = bytearray(good)
    b[21] |= 0x0F  # top bits of the 36-bit total-samples field
    b[22] = 0xFF
    p.write_bytes(bytes(b))
    assert rd.probe([str(p)]) is None
    with pytest.raises(RuntimeError, match="announces"):
        read_flac(p)
    # damage to the metadata: probe refuses or the frames no longer match; never a crash
    for trial in range(100):
        b = bytearray(good)
        b[int(rng.integers(0, first_frame))] = int(rng.integers(0, 256))
        p.write_bytes(bytes(b))
        probed = rd.probe([str(p)])
        if probed is not None:
            try:
                rd.load(probed, [0])
            except OSError:
                pass


def test_flac_scp_goes_through_the_native_batch_reader(tmp_path):
    """A wav.scp of FLAC files (what `format_wav_scp.sh` writes by default) mixed with a wav: the batched iterator
    serves it from the native reader with the same batches as the Python reader + collate."""
    files = []
    for i, n in enumerate([3000, 1200, 4096, 2500, 800]):
        x = _speechlike(n, 40 + i)
        f = tmp_path / f"u{i}.flac"
        plan = (lambda fi, ci, s, bps: dict(kind="fixed", order=2, po=0)) if i % 2 else None
        f.write_bytes(encode_flac([x], 16, 16000, 1024 if i % 2 else 4096, plan))
        files.append(f)
    write_wav_pcm16(tmp_path / "w.wav", (_speechlike(2000, 50) / 32768.0).astype(np.float32), 16000)
    files.append(tmp_path / "w.wav")
    spec = [(_scp(tmp_path, files), "speech", "sound")]
    runs = []
    for native in (True, False):
        it = StreamingBatchIterator(IterableESPnetDataset(spec), batch_size=4, bucket_window=2, num_workers=2,
                                    native_reader=native)
        runs.append(([(k, b["speech"].clone(), b["speech_lengths"].tolist()) for k, b in it], it.native_windows))
    (nat, nw), (ref, rw) = runs
    assert nw == 1 and rw == 0 and len(nat) == len(ref) == 2
    for (k1, s1, l1), (k2, s2, l2) in zip(nat, ref):
        assert k1 == k2 and l1 == l2 and torch.equal(s1, s2)
