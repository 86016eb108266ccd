"""Where does the HOST spend a call of one stream (Speech2TextStreaming.__call__, hipGraph step)?  cProfile over the steady-state
calls of a few utterances; prints the top functions by cumulative and by own time.  `python tools/experiments/stream_one_host_profile.py`"""
import cProfile
import pstats
import sys
import tempfile
import time
from pathlib import Path

import torch
import yaml

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402
from espnet_amd.bin.asr_inference_streaming import Speech2TextStreaming  # noqa: E402

enc_conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12, input_layer="conv2d", normalize_before=True,
                activation_type="swish", macaron_style=True, use_cnn_module=True, cnn_module_kernel=15, block_size=40, hop_size=16,
                look_ahead=16, init_average=True, ctx_pos_enc=True)
cfg = dict(token_list=["<blank>", "<unk>"] + [f"t{i}" for i in range(bench.VOCAB - 3)] + ["<sos/eos>"], frontend="default",
           frontend_conf=dict(n_fft=512, hop_length=160, win_length=400), normalize="utterance_mvn", normalize_conf={},
           encoder="contextual_block_conformer", encoder_conf=enc_conf, decoder="transformer",
           decoder_conf=dict(attention_heads=4, linear_units=2048, num_blocks=6), model_conf=dict(ctc_weight=0.3))
torch.manual_seed(0)
with tempfile.TemporaryDirectory() as td:
    (Path(td) / "config.yaml").write_text(yaml.safe_dump(cfg))
    s2t = Speech2TextStreaming(str(Path(td) / "config.yaml"), None, device="cuda", dtype="bfloat16", beam_size=1, ctc_weight=0.3)
wav = bench.synth_batch(0, 1)[0]
chunks = [wav[p : p + 10240] for p in range(0, bench.N_SAMPLES, 10240)]


def utt():
    for k, c in enumerate(chunks):
        s2t(c, is_final=(k == len(chunks) - 1))


utt()
utt()
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(6):
    utt()
pr.disable()
print(f"{(time.perf_counter() - t0) / (6 * len(chunks)) * 1e3:.3f} ms per call under the profiler")
for key in ("cumulative", "tottime"):
    st = pstats.Stats(pr)
    st.sort_stats(key).print_stats(22)
