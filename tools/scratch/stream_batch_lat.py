import sys, time, torch
sys.path.insert(0, '.')
import bench
from bench import *
import yaml, tempfile
from pathlib import Path
from espnet_amd.bin.asr_inference_streaming import Speech2TextStreaming
enc_conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12, input_layer="conv2d", normalize_before=True, activation_type="swish", macaron_style=True, use_cnn_module=True, cnn_module_kernel=15, block_size=40, hop_size=16, look_ahead=16, init_average=True, ctx_pos_enc=True)
cfg = dict(token_list=["<blank>", "<unk>"] + [f"t{i}" for i in range(VOCAB - 3)] + ["<sos/eos>"], frontend="default", frontend_conf=dict(n_fft=512, hop_length=160, win_length=400), normalize="utterance_mvn", normalize_conf={}, encoder="contextual_block_conformer", encoder_conf=enc_conf, decoder="transformer", decoder_conf=dict(attention_heads=4, linear_units=2048, num_blocks=6), model_conf=dict(ctc_weight=0.3))
torch.manual_seed(0)
with tempfile.TemporaryDirectory() as td:
    (Path(td) / "config.yaml").write_text(yaml.safe_dump(cfg))
    s2t = Speech2TextStreaming(str(Path(td) / "config.yaml"), None, device="cuda", dtype="bfloat16", beam_size=1, ctc_weight=0.3, use_hipgraph=False)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
wav = synth_batch(0, S)
chunk = 10240
bounds = [(p, min(N_SAMPLES, p + chunk)) for p in range(0, N_SAMPLES, chunk)]
for rep in range(3):
    lat = []
    for k, (lo, hi) in enumerate(bounds):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = s2t.batch_call(wav[:, lo:hi], is_final=(k == len(bounds) - 1))
        torch.cuda.synchronize(); lat.append(round((time.perf_counter() - t0) * 1e3, 2))
    print(rep, lat)
# where does a slow tick spend its time?
import cProfile, pstats
s2t.reset()
pr = cProfile.Profile()
for k, (lo, hi) in enumerate(bounds):
    if k == 5: pr.enable()
    out = s2t.batch_call(wav[:, lo:hi], is_final=(k == len(bounds) - 1))
    if k == 5: torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
