"""Decode CLI end to end on the MI355X against the reference's own CLI output.

tests/golden/cli_decode.npz holds the files the reference's `python -m espnet2.bin.asr_inference`
wrote for a ragged 7-utterance wav.scp (one below the subsampling limit -> TooShortUttError row).
"""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import golden_state_dict, load_golden  # noqa: E402


def _setup(tmp_path):
    from espnet_amd.fileio.sound_scp import write_wav_pcm16
    from oracle.weights import synth_waveform

    g = load_golden("cli_decode")
    (tmp_path / "config.yaml").write_text(str(g["config_yaml"]))
    torch.save(golden_state_dict(g), tmp_path / "model.pth")
    lines = []
    for key, u, n in json.loads(str(g["utts"])):
        write_wav_pcm16(tmp_path / f"{key}.wav", synth_waveform(u, n).numpy(), 16000)
        lines.append(f"{key} {tmp_path / (key + '.wav')}")
    (tmp_path / "wav.scp").write_text("\n".join(lines) + "\n")
    return g, json.loads(str(g["files"]))


def _run(tmp_path, out, *extra):
    from espnet_amd.bin.asr_inference import main

    g = load_golden("cli_decode")
    summary = main(["--output_dir", str(tmp_path / out), "--ngpu", "1",
                    "--data_path_and_name_and_type", f"{tmp_path / 'wav.scp'},speech,sound",
                    "--asr_train_config", str(tmp_path / "config.yaml"),
                    "--asr_model_file", str(tmp_path / "model.pth"), "--beam_size", str(int(g["beam"])),
                    "--ctc_weight", str(float(g["ctc_weight"])), "--nbest", str(int(g["nbest"])),
                    "--lm_weight", "0.0", *extra])
    files = {}
    for f in sorted((tmp_path / out).rglob("*")):
        if f.is_file():
            files[str(f.relative_to(tmp_path / out))] = f.read_text()
    return summary, files


def _table(text):
    return {ln.split(maxsplit=1)[0]: (ln.split(maxsplit=1)[1] if " " in ln.strip() else "")
            for ln in text.splitlines()}


def test_cli_f32_batched_matches_reference_files(tmp_path):
    g, ref = _setup(tmp_path)
    summary, mine = _run(tmp_path, "out", "--dtype", "float32", "--batch_size", "3", "--bucket_window", "1",
                         "--num_workers", "2")
    assert sorted(mine) == sorted(ref)  # same file set: {1,2}best_recog/{score,text,token,token_int}
    keys = [k for k, _, _ in json.loads(str(g["utts"]))]
    for name in mine:  # input order, whatever the bucketing did
        assert [ln.split()[0] for ln in mine[name].splitlines()] == keys, name
    assert summary["utterances"] == len(keys) and summary["rtf"] > 0
    s1, s2 = _table(ref["1best_recog/score"]), _table(ref["2best_recog/score"])
    m1 = _table(mine["1best_recog/score"])
    checked = 0
    for k in keys:
        if k == "uttC":  # too short: the reference's placeholder row, verbatim (:852-854)
            for name in ("token", "token_int", "text", "score"):
                assert _table(mine[f"1best_recog/{name}"])[k] == _table(ref[f"1best_recog/{name}"])[k]
            continue
        gap = float(s1[k].replace("tensor(", "").rstrip(")")) - float(s2[k].replace("tensor(", "").rstrip(")"))
        sc_ref = float(s1[k].replace("tensor(", "").rstrip(")"))
        sc = float(m1[k].replace("tensor(", "").rstrip(")"))
        # HIP frontend + encoder in exact-f32 mode: activation differences ~1e-4 move scores ~1e-2
        assert abs(sc - sc_ref) < 2e-2 + 5e-5 * abs(sc_ref), (k, sc, sc_ref)
        if gap > 5e-2:  # a clear winner must be the same token sequence, token for token
            for name in ("token", "token_int", "text"):
                assert _table(mine[f"1best_recog/{name}"])[k] == _table(ref[f"1best_recog/{name}"])[k], (k, name)
            checked += 1
    assert checked >= 3, checked


def test_cli_batching_is_transparent_and_bf16_runs(tmp_path):
    g, _ = _setup(tmp_path)
    _, one = _run(tmp_path, "b1", "--dtype", "float32", "--batch_size", "1")
    _, four = _run(tmp_path, "b4", "--dtype", "float32", "--batch_size", "4", "--bucket_window", "2")
    for name in ("1best_recog/token_int", "1best_recog/text", "2best_recog/token_int"):
        assert one[name] == four[name], name
    a, b = _table(one["1best_recog/score"]), _table(four["1best_recog/score"])
    for k in a:
        fa, fb = (float(x.replace("tensor(", "").rstrip(")")) for x in (a[k], b[k]))
        assert abs(fa - fb) < 2e-3, (k, fa, fb)
    _, bf = _run(tmp_path, "bf", "--batch_size", "4")  # default dtype bfloat16
    assert sorted(bf) == sorted(one)
    assert len(bf["1best_recog/token_int"].splitlines()) == len(one["1best_recog/token_int"].splitlines())
    _, gr = _run(tmp_path, "g1", "--batch_size", "4", "--ctc_greedy", "true", "--nbest", "1")
    assert set(gr) == {"1best_recog/score", "1best_recog/text", "1best_recog/token", "1best_recog/token_int"}


def test_streaming_cli_equals_the_object_api(tmp_path):
    """`python -m espnet_amd.bin.asr_inference_streaming` (espnet2/bin/asr_inference_streaming.py:362-494): the
    result files equal what feeding `Speech2TextStreaming` the same chunks by hand gives (the object API is what
    tests/test_gpu_online_search.py pins against the reference's per-call goldens), for whole-utterance calls and
    for simulated chunks, in input order."""
    from espnet_amd.bin.asr_inference_streaming import Speech2TextStreaming, main
    from espnet_amd.fileio.sound_scp import read_wav, write_wav_pcm16
    from oracle.weights import synth_waveform

    g = load_golden("stream_search_a")
    (tmp_path / "config.yaml").write_text(str(g["config_yaml"]))
    torch.save(golden_state_dict(g), tmp_path / "model.pth")
    n0, chunk = int(g["n_samples"]), int(g["chunk_samples"])
    utts = [("b_long", int(g["utt_id"]), n0), ("a_short", int(g["utt_id"]) + 1, n0 // 2 + 123)]
    lines = []
    for key, u, n in utts:
        write_wav_pcm16(tmp_path / f"{key}.wav", synth_waveform(u, n).numpy(), 16000)
        lines.append(f"{key} {tmp_path / (key + '.wav')}")
    (tmp_path / "wav.scp").write_text("\n".join(lines) + "\n")
    opts = dict(beam_size=int(g["beam"]), ctc_weight=float(g["ctc_weight"]), penalty=float(g["penalty"]),
                nbest=int(g["nbest"]), disable_repetition_detection=bool(g["disable_repetition_detection"]))
    s2t = Speech2TextStreaming(str(tmp_path / "config.yaml"), str(tmp_path / "model.pth"), device="cuda",
                               dtype="float32", **opts)
    for sim in (0, chunk):
        out = tmp_path / f"out{sim}"
        summary = main(["--output_dir", str(out), "--ngpu", "1", "--dtype", "float32",
                        "--data_path_and_name_and_type", f"{tmp_path / 'wav.scp'},speech,sound",
                        "--asr_train_config", str(tmp_path / "config.yaml"),
                        "--asr_model_file", str(tmp_path / "model.pth"), "--sim_chunk_length", str(sim),
                        "--beam_size", str(opts["beam_size"]), "--ctc_weight", str(opts["ctc_weight"]),
                        "--penalty", str(opts["penalty"]), "--nbest", str(opts["nbest"]),
                        "--disable_repetition_detection", str(opts["disable_repetition_detection"]),
                        "--log_level", "WARNING"])
        assert summary["utterances"] == 2
        tok = (out / "1best_recog/token_int").read_text().splitlines()
        score = (out / "1best_recog/score").read_text().splitlines()
        assert [ln.split()[0] for ln in tok] == ["b_long", "a_short"]  # input order
        for i, (key, _, n) in enumerate(utts):
            wav = torch.from_numpy(read_wav(tmp_path / f"{key}.wav", dtype="float32")[0])
            if sim == 0:
                res = s2t(wav, is_final=True)
            else:
                k = n // sim
                for j in range(k):
                    s2t(wav[j * sim : (j + 1) * sim], is_final=False)
                res = s2t(wav[k * sim :], is_final=True)
            assert res, key
            assert tok[i] == f"{key} " + " ".join(map(str, res[0][2])), (sim, key)
            import re

            written = float(re.search(r"-?\d+\.?\d*(e[-+]?\d+)?", score[i].split(maxsplit=1)[1]).group(0))
            assert abs(written - float(res[0][3].score)) < 1e-3 + 1e-5 * abs(written), (sim, key)
