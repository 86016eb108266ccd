#!/usr/bin/env python3
"""Recipe-level throughput of the decode CLI (developer tool; run on the GPU box): N synthetic 16-bit wavs of
ragged length on local disk -> `espnet_amd.bin.asr_inference.inference()` -> result files.  Reports the
CLI's own RTF summary next to where the time goes (reader threads vs the device)."""
import argparse
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import yaml

import bench
from espnet_amd.bin.asr_inference import inference
from espnet_amd.fileio.sound_scp import write_wav_pcm16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--window", type=int, default=8)
    ap.add_argument("--beam", type=int, default=0, help="0 = greedy CTC (G1)")
    ap.add_argument("--model", default="small")
    ap.add_argument("--profile", action="store_true", help="cProfile the main thread of the warm pass")
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        cfg = bench.model_config(a.model, "bfloat16")
        (td / "config.yaml").write_text(yaml.safe_dump(cfg))
        torch.manual_seed(0)
        lines = []
        t0 = time.perf_counter()
        for i in range(a.n):
            n = int(rng.integers(4, 16) * 16000)  # 4 .. 15 s
            write_wav_pcm16(td / f"u{i:05d}.wav", rng.normal(0, 0.1, n).astype(np.float32), 16000)
            lines.append(f"u{i:05d} {td / f'u{i:05d}.wav'}")
        (td / "wav.scp").write_text("\n".join(lines) + "\n")
        print(f"wrote {a.n} wavs in {time.perf_counter() - t0:.1f} s", flush=True)
        for rep in range(2):  # second pass: page cache and kernels warm
            prof = None
            if a.profile and rep == 1:
                import cProfile

                prof = cProfile.Profile()
                prof.enable()
            s = inference(output_dir=str(td / f"out{rep}"), batch_size=a.batch_size, dtype="bfloat16", ngpu=1,
                          beam_size=max(1, a.beam), ctc_weight=0.3, lm_weight=0.0, num_workers=a.workers,
                          data_path_and_name_and_type=[(str(td / "wav.scp"), "speech", "sound")],
                          asr_train_config=str(td / "config.yaml"), asr_model_file=None, log_level="WARNING",
                          ctc_greedy=a.beam == 0, bucket_window=a.window)
            if prof is not None:
                import pstats

                prof.disable()
                pstats.Stats(prof).sort_stats("tottime").print_stats(28)
                pstats.Stats(prof).sort_stats("cumulative").print_stats(40)
            print(f"pass {rep}: {s['utterances']} utts, {s['audio_seconds']:.0f} audio-s in {s['wall_seconds']:.2f} s "
                  f"-> {s['audio_seconds'] / s['wall_seconds']:.0f} audio-s/s (RTF {s['rtf']:.6f}); native reader "
                  f"windows {s['native_reader_windows']}, reader seconds "
                  f"{ {k: round(v, 3) for k, v in s['reader_seconds'].items()} }", flush=True)


if __name__ == "__main__":
    main()
