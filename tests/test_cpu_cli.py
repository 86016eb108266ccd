"""Host side of the decode CLI: file formats, the bucketed iterator and the
parser — no GPU.  References: espnet2/fileio/datadir_writer.py, fileio/sound_scp.py,
train/iterable_dataset.py, train/collate_fn.py, bin/asr_inference.py:911-1137."""
import json
import struct
import wave
import warnings
from pathlib import Path

import numpy as np
import pytest

from espnet_amd.fileio.datadir_writer import DatadirWriter
from espnet_amd.fileio.read_text import read_2columns_text
from espnet_amd.fileio.sound_scp import SoundScpReader, read_wav, write_wav_pcm16
from espnet_amd.train.iterable_dataset import IterableESPnetDataset, StreamingBatchIterator, common_collate_fn

GOLD = Path(__file__).parent / "golden"


def test_datadir_writer_layout_and_guards(tmp_path):
    with DatadirWriter(tmp_path / "out") as w:
        w["1best_recog"]["token"]["utt1"] = "a b c"
        w["1best_recog"]["token"]["utt2"] = ""
        w["1best_recog"]["score"]["utt1"] = "-1.5"
        with pytest.raises(RuntimeError):
            w["1best_recog"]["x"] = "y"  # a directory node takes no lines
        with pytest.raises(RuntimeError):
            w["1best_recog"]["token"]["sub"]  # a file node has no children
        with pytest.warns(UserWarning, match="Duplicated"):
            w["1best_recog"]["token"]["utt1"] = "again"
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            w.close()
        assert any("mismatching" in str(r.message) for r in rec)  # token has utt2, score does not
    assert (tmp_path / "out/1best_recog/token").read_text() == "utt1 a b c\nutt2 \nutt1 again\n"
    assert (tmp_path / "out/1best_recog/score").read_text() == "utt1 -1.5\n"


def test_read_2columns_text(tmp_path):
    p = tmp_path / "wav.scp"
    p.write_text("k1 /a b/c.wav\nk2\nk3   x\n")
    assert read_2columns_text(p) == {"k1": "/a b/c.wav", "k2": "", "k3": "x"}
    p.write_text("k1 a\nk1 b\n")
    with pytest.raises(RuntimeError, match="duplicated"):
        read_2columns_text(p)


@pytest.mark.parametrize("width", [1, 2, 3, 4])
def test_read_wav_pcm_matches_stdlib_wave(tmp_path, width):
    """Integer PCM of every width: value / 2**(8w-1) (8-bit unsigned), mono and stereo."""
    rng = np.random.default_rng(width)
    for nch in (1, 2):
        n = 777
        if width == 1:
            ints = rng.integers(0, 256, size=(n, nch))
            raw = ints.astype(np.uint8).tobytes()
            want = (ints - 128) / 128.0
        else:
            lim = 2 ** (8 * width - 1)
            ints = rng.integers(-lim, lim, size=(n, nch))
            raw = b"".join(int(v).to_bytes(width, "little", signed=True) for v in ints.reshape(-1))
            want = ints / float(lim)
        f = tmp_path / f"w{width}_{nch}.wav"
        with wave.open(str(f), "wb") as w:
            w.setnchannels(nch), w.setsampwidth(width), w.setframerate(8000)
            w.writeframes(raw)
        x, rate = read_wav(f)
        assert rate == 8000 and x.dtype == np.float64
        assert x.shape == ((n,) if nch == 1 else (n, nch))
        np.testing.assert_array_equal(x.reshape(n, nch), want)


def test_read_wav_float_extra_chunks_and_errors(tmp_path):
    x = np.linspace(-1, 1, 101, dtype="<f4")
    body = x.tobytes()
    fmt = struct.pack("<HHIIHH", 3, 1, 16000, 16000 * 4, 4, 32)
    junk = b"LIST" + struct.pack("<I", 3) + b"abc" + b"\x00"  # odd-sized chunk + pad byte
    riff = b"WAVE" + junk + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(body)) + body
    f = tmp_path / "f32.wav"
    f.write_bytes(b"RIFF" + struct.pack("<I", len(riff)) + riff)
    y, rate = read_wav(f, dtype="float32")
    assert rate == 16000
    np.testing.assert_array_equal(y, x)
    (tmp_path / "a.ogg").write_bytes(b"OggS" + b"\0" * 40)
    with pytest.raises(NotImplementedError):
        read_wav(tmp_path / "a.ogg")
    (tmp_path / "a.flac").write_bytes(b"fLaC" + b"\0" * 40)  # FLAC is decoded (tests/test_cpu_flac.py); this is not one
    with pytest.raises(RuntimeError):
        read_wav(tmp_path / "a.flac")
    with pytest.raises(NotImplementedError):
        read_wav("sox a.wav -t wav - |")


def test_write_wav_roundtrip_and_scp_reader(tmp_path):
    rng = np.random.default_rng(0)
    x = (rng.integers(-32768, 32768, size=1000) / 32768.0).astype(np.float32)
    write_wav_pcm16(tmp_path / "a.wav", x, 16000)
    with wave.open(str(tmp_path / "a.wav"), "rb") as w:  # a standard reader agrees on the container
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 16000, 1000)
    (tmp_path / "wav.scp").write_text(f"u1 {tmp_path / 'a.wav'}\n")
    r = SoundScpReader(tmp_path / "wav.scp")
    rate, y = r["u1"]
    assert rate == 16000 and list(r) == ["u1"] and len(r) == 1 and "u1" in r
    np.testing.assert_array_equal(y, x.astype(np.float64))


def _make_set(tmp_path, lens):
    lines = []
    for i, n in enumerate(lens):
        np.save(tmp_path / f"u{i}.npy", np.full(n, i, dtype=np.float64))
        lines.append(f"u{i} {tmp_path / f'u{i}.npy'}")
    (tmp_path / "feats.scp").write_text("\n".join(lines) + "\n")
    return str(tmp_path / "feats.scp")


def test_iterable_dataset_order_keyfile_and_dtype(tmp_path):
    scp = _make_set(tmp_path, [5, 3, 9, 4])
    ds = IterableESPnetDataset([(scp, "speech", "npy")])
    got = list(ds)
    assert [u for u, _ in got] == ["u0", "u1", "u2", "u3"]
    assert all(d["speech"].dtype == np.float32 for _, d in got)  # float_dtype cast (:240-243)
    (tmp_path / "keys").write_text("u1\nu3\n")
    ds = IterableESPnetDataset([(scp, "speech", "npy")], key_file=str(tmp_path / "keys"))
    assert [u for u, _ in ds] == ["u1", "u3"]
    (tmp_path / "keys").write_text("u1\nzz\n")
    with pytest.raises(RuntimeError, match="not found"):
        list(IterableESPnetDataset([(scp, "speech", "npy")], key_file=str(tmp_path / "keys")))
    (tmp_path / "empty").write_text("")
    with pytest.raises(RuntimeError, match="No iteration"):
        list(IterableESPnetDataset([(scp, "speech", "npy")], key_file=str(tmp_path / "empty")))
    with pytest.raises(NotImplementedError):
        list(IterableESPnetDataset([(scp, "speech", "kaldi_ark")]))


def test_collate_pads_and_reports_lengths():
    data = [("a", {"speech": np.ones(3, np.float32)}), ("b", {"speech": np.ones(5, np.float32)})]
    keys, batch = common_collate_fn(data)
    assert keys == ["a", "b"] and batch["speech"].shape == (2, 5)
    assert batch["speech"][0].tolist() == [1, 1, 1, 0, 0] and batch["speech_lengths"].tolist() == [3, 5]


@pytest.mark.parametrize("batch_size,window,workers", [(1, 1, 1), (3, 2, 2), (4, 8, 3), (16, 1, 1)])
def test_bucketed_iterator_covers_all_and_sorts_within_window(tmp_path, batch_size, window, workers):
    lens = [7, 30, 12, 5, 22, 22, 9, 40, 3, 18, 27]
    scp = _make_set(tmp_path, lens)
    ds = IterableESPnetDataset([(scp, "speech", "npy")])
    it = StreamingBatchIterator(ds, batch_size=batch_size, bucket_window=window, num_workers=workers,
                                length_key="speech")
    seen, batches = [], []
    for keys, batch in it:
        assert len(keys) <= batch_size and batch["speech"].shape[0] == len(keys)
        bl = batch["speech_lengths"].tolist()
        assert bl == sorted(bl, reverse=True)  # longest first inside a batch
        for k, n, row in zip(keys, bl, batch["speech"]):
            i = int(k[1:])
            assert n == lens[i] and row[:n].eq(i).all() and row[n:].eq(0).all()
        seen += keys
        batches.append(bl)
    assert sorted(seen) == sorted(f"u{i}" for i in range(len(lens)))
    assert it.key_order == [f"u{i}" for i in range(len(lens))]
    w = batch_size * window
    for s in range(0, len(lens), w):  # a window's utterances come out before the next window's
        assert set(seen[s : s + w]) == {f"u{i}" for i in range(s, min(s + w, len(lens)))}
    if batch_size == 4 and window == 8:  # one window: globally sorted -> padding waste is minimal
        flat = [n for b in batches for n in b]
        assert flat == sorted(lens, reverse=True)


def test_iterator_surfaces_reader_errors(tmp_path):
    scp = _make_set(tmp_path, [4, 4])
    (tmp_path / "u1.npy").unlink()
    it = StreamingBatchIterator(IterableESPnetDataset([(scp, "speech", "npy")]), batch_size=2)
    with pytest.raises(FileNotFoundError):
        list(it)


def _riff(fmt_tag, nch, rate, bits, body, extensible=False, junk=b"", data_size=None):
    blk = nch * bits // 8
    if extensible:  # WAVE_FORMAT_EXTENSIBLE: the real tag sits in the SubFormat GUID
        fmt = struct.pack("<HHIIHH", 0xFFFE, nch, rate, rate * blk, blk, bits) + struct.pack("<HHI", 22, bits, 0) \
              + struct.pack("<H", fmt_tag) + b"\x00" * 14
    else:
        fmt = struct.pack("<HHIIHH", fmt_tag, nch, rate, rate * blk, blk, bits)
    riff = b"WAVE" + junk + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + \
           struct.pack("<I", len(body) if data_size is None else data_size) + body
    return b"RIFF" + struct.pack("<I", len(riff)) + riff


def _wav_zoo(tmp_path):
    """Mono wavs of every sample format the readers take, plus awkward containers; returns scp lines."""
    rng = np.random.default_rng(5)
    junk = b"LIST" + struct.pack("<I", 3) + b"abc" + b"\x00"
    files = {
        "pcm16_a": _riff(1, 1, 16000, 16, rng.integers(-32768, 32768, 4001).astype("<i2").tobytes()),
        "pcm16_ext": _riff(1, 1, 16000, 16, rng.integers(-32768, 32768, 2500).astype("<i2").tobytes(), extensible=True),
        "pcm8": _riff(1, 1, 8000, 8, rng.integers(0, 256, 1777).astype(np.uint8).tobytes(), junk=junk),
        "pcm24": _riff(1, 1, 16000, 24, b"".join(int(v).to_bytes(3, "little", signed=True)
                                                  for v in rng.integers(-2 ** 23, 2 ** 23, 1234))),
        "pcm32": _riff(1, 1, 16000, 32, rng.integers(-2 ** 31, 2 ** 31, 999).astype("<i4").tobytes()),
        "f32": _riff(3, 1, 16000, 32, rng.normal(0, 0.3, 3100).astype("<f4").tobytes()),
        "f64": _riff(3, 1, 16000, 64, rng.normal(0, 0.3, 2100).astype("<f8").tobytes()),
        "trunc": _riff(1, 1, 16000, 16, rng.integers(-32768, 32768, 700).astype("<i2").tobytes(), data_size=4000),
        "pcm16_b": _riff(1, 1, 16000, 16, rng.integers(-32768, 32768, 4001).astype("<i2").tobytes()),
        "empty": _riff(1, 1, 16000, 16, b""),
    }
    lines = []
    for k, blob in files.items():
        (tmp_path / f"{k}.wav").write_bytes(blob)
        lines.append(f"{k} {tmp_path / f'{k}.wav'}")
    return lines


@pytest.mark.parametrize("batch_size,window,workers", [(4, 8, 3), (3, 1, 1), (16, 1, 2)])
def test_native_wav_reader_equals_python_reader_bit_for_bit(tmp_path, batch_size, window, workers):
    """em_wav_probe / em_wav_load_rows (csrc/host_io.cpp) behind StreamingBatchIterator: same batches, keys,
    lengths and sample VALUES (incl. the zero padding) as read_wav + common_collate_fn, for every PCM width,
    IEEE float, WAVE_FORMAT_EXTENSIBLE, extra chunks, a truncated data chunk and an empty file."""
    (tmp_path / "wav.scp").write_text("\n".join(_wav_zoo(tmp_path)) + "\n")
    spec = [(str(tmp_path / "wav.scp"), "speech", "sound")]
    runs = []
    for native in (True, False):
        it = StreamingBatchIterator(IterableESPnetDataset(spec), batch_size=batch_size, bucket_window=window,
                                    num_workers=workers, native_reader=native)
        runs.append(([(k, {n: v.clone() for n, v in b.items()}) for k, b in it], it.key_order, it.native_windows))
    (nat, nat_order, nat_windows), (ref, ref_order, ref_windows) = runs
    assert nat_windows == -(-10 // (batch_size * window)) and ref_windows == 0
    assert nat_order == ref_order and len(nat) == len(ref)
    for (k1, b1), (k2, b2) in zip(nat, ref):
        assert k1 == k2 and set(b1) == set(b2) == {"speech", "speech_lengths"}
        assert b1["speech"].dtype == b2["speech"].dtype and b1["speech_lengths"].dtype == b2["speech_lengths"].dtype
        assert b1["speech_lengths"].tolist() == b2["speech_lengths"].tolist()
        assert b1["speech"].shape == b2["speech"].shape
        assert np.array_equal(b1["speech"].numpy().view(np.uint32), b2["speech"].numpy().view(np.uint32))


def test_native_wav_reader_leaves_other_inputs_to_the_python_reader(tmp_path):
    """Stereo / non-wav / missing files, a preprocessor that changes samples, npy entries: the window is read by
    the Python path (same results and the same descriptive errors as before)."""
    from espnet_amd.tasks.asr import ASRTask

    lines = _wav_zoo(tmp_path)[:3]
    st = np.random.default_rng(1).integers(-32768, 32768, (300, 2)).astype("<i2")
    (tmp_path / "stereo.wav").write_bytes(_riff(1, 2, 16000, 16, st.tobytes()))
    (tmp_path / "wav.scp").write_text("\n".join(lines + [f"st {tmp_path / 'stereo.wav'}"]) + "\n")
    spec = [(str(tmp_path / "wav.scp"), "speech", "sound")]
    it = StreamingBatchIterator(IterableESPnetDataset(spec), batch_size=1, bucket_window=8)
    got = dict((k[0], b["speech"]) for k, b in it)
    assert it.native_windows == 0 and got["st"].shape == (1, 300, 2)
    np.testing.assert_array_equal(got["st"][0].numpy(), st.astype(np.float32) / 32768)
    # volume normalisation rewrites the samples -> not eligible; the identity preprocessor is
    for vol, want in ((0.5, 0), (None, 1)):
        pre = ASRTask.build_preprocess_fn(dict(use_preprocessor=True, speech_volume_normalize=vol), False)
        (tmp_path / "wav.scp").write_text("\n".join(lines) + "\n")
        it = StreamingBatchIterator(IterableESPnetDataset(spec, preprocess=pre), batch_size=2, bucket_window=8,
                                    collate_fn=ASRTask.build_collate_fn(None, False))  # the CLI's partial
        out = list(it)
        assert it.native_windows == want and sum(len(k) for k, _ in out) == 3
    # a missing file surfaces the Python reader's error
    (tmp_path / "wav.scp").write_text(f"gone {tmp_path / 'gone.wav'}\n")
    with pytest.raises(FileNotFoundError):
        list(StreamingBatchIterator(IterableESPnetDataset(spec), batch_size=1))
    (tmp_path / "a.ogg").write_bytes(b"OggS" + b"\0" * 40)
    (tmp_path / "wav.scp").write_text(f"og {tmp_path / 'a.ogg'}\n")
    with pytest.raises(NotImplementedError):
        list(StreamingBatchIterator(IterableESPnetDataset(spec), batch_size=1))


def test_parser_matches_reference_option_table():
    """Every option of the reference CLI is accepted; defaults are the reference's except the documented
    ones (ngpu: no CPU path; token_type also lists "word").  --dtype keeps the reference's default (float32, the
    exact-f32 parity mode) and choices, plus bfloat16 for the bf16 MFMA mode."""
    from espnet_amd.bin.asr_inference import get_parser

    ref = json.loads((GOLD / "asr_inference_cli_options.json").read_text())
    mine = {a.dest: a for a in get_parser()._actions if a.dest != "help"}
    assert set(ref) <= set(mine), sorted(set(ref) - set(mine))
    assert set(mine) - set(ref) == {"bucket_window", "ctc_greedy"}
    for name, r in ref.items():
        a = mine[name]
        assert a.required == r["required"], name
        if name == "ngpu":
            continue
        if name == "dtype":
            assert set(r["choices"]) <= set(a.choices) and "bfloat16" in a.choices
        d = r["default"]
        if name == "hugging_face_decoder_conf":
            d = {}
        assert a.default == d, (name, a.default, d)
    assert mine["ngpu"].default == 1 and mine["dtype"].default == "float32"


def test_dtype_names_of_the_reference_map_onto_the_two_mfma_modes(caplog):
    from espnet_amd.bin.asr_inference import resolve_dtype

    assert resolve_dtype("float32") == "float32" and resolve_dtype("bfloat16") == "bfloat16"
    with caplog.at_level("WARNING"):
        assert resolve_dtype("float16") == "bfloat16" and resolve_dtype("float64") == "float32"
    assert "float16" in caplog.text and "float64" in caplog.text
    with pytest.raises(ValueError):
        resolve_dtype("int8")


def test_cli_refuses_cpu_and_more_gpus_than_present(tmp_path):
    from espnet_amd.bin.asr_inference import main

    base = ["--output_dir", str(tmp_path / "o"), "--data_path_and_name_and_type", "x.scp,speech,sound"]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        main(base + ["--ngpu", "0"])
    # --ngpu N spawns one rank per GPU (tests/test_distributed_cpu.py covers the sharding and merge); a box
    # with fewer GPUs says so instead of falling back
    with pytest.raises(RuntimeError, match="GPUs are visible"):
        main(base + ["--ngpu", "2"])


def test_inference_loop_orders_outputs_and_handles_short_utterances(tmp_path, monkeypatch):
    """The decode loop of `inference()` on the CPU with a stand-in Speech2Text (the device work is covered by the
    GPU suite): results written in INPUT order whatever the length bucketing did, one batch kept in flight,
    too-short utterances replaced by the reference's placeholder row (asr_inference.py:851-858) without taking
    their batch down, n-best files, RTF summary."""
    import argparse

    import torch

    from espnet_amd.bin import asr_inference as ai
    from espnet_amd.lib import TooShortUttError
    from espnet_amd.nets.beam_search import Hypothesis

    lens = [5000, 900, 16000, 300, 7000, 12000, 450, 8000, 9500]
    lines = []
    for i, n in enumerate(lens):
        write_wav_pcm16(tmp_path / f"u{i}.wav", np.full(n, (i + 1) / 64.0, dtype=np.float32), 16000)
        lines.append(f"utt{i} {tmp_path / f'u{i}.wav'}")
    (tmp_path / "wav.scp").write_text("\n".join(lines) + "\n")
    calls = []

    class FakeS2T:
        asr_train_args = argparse.Namespace(frontend_conf=dict(fs=16000), use_preprocessor=True, preprocessor_conf={})

        def _decode(self, speech, lengths):
            short = [i for i, n in enumerate(lengths) if n < 1000]
            if short:
                raise TooShortUttError("has 3 frames and is too short for subsampling", 3, 7, indices=short)
            out = []
            for row, n in zip(speech, lengths):
                uid = int(round(float(row[0]) * 64.0)) - 1  # the constant sample value encodes the utterance
                assert bool((row[:n] == row[0]).all()) and bool((row[n:] == 0).all())  # zero padded to the batch max
                ids = [uid + 3, n % 97 + 3]
                hyps = [Hypothesis(yseq=torch.tensor([49] + ids + [49]), score=torch.tensor(-float(k + uid)))
                        for k in range(2)]
                out.append([(f"t{ids[0]} t{ids[1]}", [f"t{v}" for v in ids], ids, h) for h in hyps])
            return out

        def batch_decode(self, speech, lengths):
            calls.append(("sync", list(lengths)))
            return self._decode(speech, lengths)

        def batch_decode_async(self, speech, lengths):
            calls.append(("async", list(lengths)))
            res = self._decode(speech, lengths)
            return ai._Done(res)

    monkeypatch.setattr(ai.Speech2Text, "from_pretrained", staticmethod(lambda **kw: FakeS2T()))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.Tensor, "pin_memory", lambda self, *a, **k: self)
    summary = ai.inference(output_dir=str(tmp_path / "out"), batch_size=4, bucket_window=2, num_workers=2, nbest=2,
                           ngpu=1, data_path_and_name_and_type=[(str(tmp_path / "wav.scp"), "speech", "sound")],
                           asr_train_config=None, asr_model_file=None, log_level="ERROR")
    assert summary["utterances"] == len(lens) and abs(summary["audio_seconds"] - sum(lens) / 16000) < 1e-9
    tok = (tmp_path / "out/1best_recog/token_int").read_text().splitlines()
    assert [ln.split()[0] for ln in tok] == [f"utt{i}" for i in range(len(lens))]  # input order
    for i, n in enumerate(lens):
        want = "2" if n < 1000 else f"{i + 3} {n % 97 + 3}"
        assert tok[i] == f"utt{i} {want}", tok[i]
    for name in ("token", "token_int", "score", "text"):
        for nb in (1, 2):
            assert len((tmp_path / f"out/{nb}best_recog/{name}").read_text().splitlines()) == len(lens)
    assert (tmp_path / "out/1best_recog/text").read_text().splitlines()[1] == "utt1  "  # placeholder text " "
    # batches were cut from length-sorted windows (descending), short ones retried as a smaller synchronous batch
    asyncs = [c[1] for c in calls if c[0] == "async"]
    assert all(b == sorted(b, reverse=True) for b in asyncs) and max(len(b) for b in asyncs) <= 4
    assert any(c[0] == "sync" for c in calls)


def test_streaming_parser_matches_reference_option_table():
    """Every option of the reference's streaming decode CLI (espnet2/bin/asr_inference_streaming.py:497-629) is
    accepted with the reference default, except the documented ones (ngpu, dtype, optional asr_model_file,
    token_type also lists "word")."""
    from espnet_amd.bin.asr_inference_streaming import get_parser

    ref = json.loads((GOLD / "asr_inference_streaming_cli_options.json").read_text())
    mine = {a.dest: a for a in get_parser()._actions if a.dest != "help"}
    assert set(ref) == set(mine), set(ref) ^ set(mine)
    for name, r in ref.items():
        a = mine[name]
        if name == "ngpu":
            continue
        if name == "dtype":
            assert set(r["choices"]) <= set(a.choices) and "bfloat16" in a.choices
        assert a.required == (r["required"] and name != "asr_model_file"), name
        assert a.default == r["default"], (name, a.default, r["default"])
    assert mine["ngpu"].default == 1 and mine["dtype"].default == "float32"


def test_dtype_names_of_the_reference_map_onto_the_two_mfma_modes(caplog):
    from espnet_amd.bin.asr_inference import resolve_dtype

    assert resolve_dtype("float32") == "float32" and resolve_dtype("bfloat16") == "bfloat16"
    with caplog.at_level("WARNING"):
        assert resolve_dtype("float16") == "bfloat16" and resolve_dtype("float64") == "float32"
    assert "float16" in caplog.text and "float64" in caplog.text
    with pytest.raises(ValueError):
        resolve_dtype("int8")


def test_streaming_inference_loop_chunks_like_the_reference(tmp_path, monkeypatch):
    """`asr_inference_streaming.inference()` on the CPU with a stand-in Speech2TextStreaming: input order, the
    sim_chunk_length split (full chunks with is_final=False, the remainder — possibly empty — as the final call,
    asr_inference_streaming.py:462-475), an utterance shorter than one chunk, the TooShortUttError placeholder."""
    import argparse

    import torch

    from espnet_amd.bin import asr_inference_streaming as st
    from espnet_amd.lib import TooShortUttError
    from espnet_amd.nets.beam_search import Hypothesis

    lens = [2500, 400, 3000, 90, 1000]
    lines = []
    for i, n in enumerate(lens):
        x = (np.arange(n) % 100 / 128.0 + i / 8.0 - 0.5).astype(np.float32)
        write_wav_pcm16(tmp_path / f"u{i}.wav", x, 16000)
        lines.append(f"utt{i} {tmp_path / f'u{i}.wav'}")
    (tmp_path / "wav.scp").write_text("\n".join(lines) + "\n")
    log = []

    class FakeStreaming:
        asr_train_args = argparse.Namespace(frontend_conf=dict(fs=16000), use_preprocessor=True, preprocessor_conf={})

        def __init__(self, **kw):
            self.kw, self.buf = kw, []

        def reset(self):
            self.buf = []

        def __call__(self, speech, is_final=True):
            assert isinstance(speech, torch.Tensor) and speech.dim() == 1
            self.buf.append(speech.clone())
            log.append((len(speech), is_final))
            if not is_final:
                return []
            whole = torch.cat(self.buf)
            self.buf = []
            if len(whole) < 100:
                raise TooShortUttError("has 3 frames and is too short for subsampling", 3, 7)
            ids = [len(whole) % 41 + 3, int(round(float(whole[0] + 0.5) * 8)) + 3]
            return [(f"t{ids[0]} t{ids[1]}", [f"t{v}" for v in ids], ids,
                     Hypothesis(yseq=torch.tensor([49] + ids + [49]), score=torch.tensor(-1.5)))]

    monkeypatch.setattr(st, "Speech2TextStreaming", FakeStreaming)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.Tensor, "pin_memory", lambda self, *a, **k: self)
    common = dict(data_path_and_name_and_type=[(str(tmp_path / "wav.scp"), "speech", "sound")],
                  asr_train_config=None, asr_model_file=None, log_level="ERROR", num_workers=2)
    for chunk in (0, 1000):
        log.clear()
        out = tmp_path / f"out{chunk}"
        s = st.inference(output_dir=str(out), sim_chunk_length=chunk, **common)
        assert s["utterances"] == len(lens) and abs(s["audio_seconds"] - sum(lens) / 16000) < 1e-9
        tok = (out / "1best_recog/token_int").read_text().splitlines()
        assert [ln.split()[0] for ln in tok] == [f"utt{i}" for i in range(len(lens))]
        for i, n in enumerate(lens):
            want = "2" if n < 100 else f"{n % 41 + 3} {i + 3}"  # the chunks were reassembled in order
            assert tok[i] == f"utt{i} {want}", (chunk, tok[i])
        assert (out / "1best_recog/text").read_text().splitlines()[3] == "utt3  "
        if chunk == 0:
            assert log == [(n, True) for n in lens]
        else:
            assert log == [(1000, False), (1000, False), (500, True), (400, True), (1000, False), (1000, False),
                           (1000, False), (0, True), (90, True), (1000, False), (0, True)]
    with pytest.raises(NotImplementedError):
        st.inference(output_dir=str(tmp_path / "x"), batch_size=2, **common)
    with pytest.raises(RuntimeError, match="MI355X"):
        st.inference(output_dir=str(tmp_path / "x"), ngpu=0, **common)


def test_native_wav_reader_agrees_with_python_reader_on_corrupted_headers(tmp_path):
    """Header fuzzing: random byte flips / truncations of a valid file.  The native reader must never accept a
    file the Python reader rejects (or the reverse for mono files), and when both accept, lengths and samples
    are identical — malformed input cannot crash it or change results."""
    from espnet_amd.fileio.sound_scp import WavBatchReader

    rng = np.random.default_rng(0)
    base = _riff(1, 1, 16000, 16, rng.integers(-32768, 32768, 300).astype("<i2").tobytes(),
                 junk=b"LIST" + struct.pack("<I", 3) + b"abc\x00")
    rd = WavBatchReader(2)
    p = tmp_path / "f.wav"
    both_ok = both_fail = 0
    for _ in range(800):
        b = bytearray(base)
        for _k in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, 70))] = int(rng.integers(0, 256))
        if rng.random() < 0.2:
            b = b[: int(rng.integers(0, len(b)))]
        p.write_bytes(bytes(b))
        try:
            x, _rate = read_wav(p, dtype="float32")
            py = x if x.ndim == 1 else None  # multi-channel stays with the Python reader
        except Exception:
            py = None
        probed = rd.probe([str(p)])
        if probed is None:
            assert py is None, bytes(b[:64]).hex()
            both_fail += 1
            continue
        out, lens = rd.load(probed, [0])
        assert py is not None, bytes(b[:64]).hex()
        assert lens[0] == len(py) and np.array_equal(out[0].numpy().view(np.uint32), py.view(np.uint32))
        both_ok += 1
    assert both_ok > 100 and both_fail > 100
