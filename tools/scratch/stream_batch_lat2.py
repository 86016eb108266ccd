import sys, time, torch
sys.path.insert(0, '.')
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
chunks_pinned = None
chunk = 10240
bounds = [(p, min(N_SAMPLES, p + chunk)) for p in range(0, N_SAMPLES, chunk)]
chunks_pinned = [wav[:, lo:hi].contiguous().pin_memory() for lo, hi in bounds]
m = s2t.asr_model
def sync(): torch.cuda.synchronize(); return time.perf_counter()
import gc
gc.collect(); gc.freeze()
for rep in range(4):
    fst = est = None
    rows = []
    for k, (lo, hi) in enumerate(bounds):
        fin = k == len(bounds) - 1
        a0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
        t0 = sync()
        feats, fst = s2t.apply_frontend_batch(chunks_pinned[k], fst, is_final=fin)
        t1 = sync()
        if feats is not None:
            enc, y_len, est = m.encoder.forward_infer_batch(feats.contiguous(), est, fin)
            t2 = sync()
            ids = m.ctc.argmax(enc).cpu().tolist() if y_len > 0 else []
            t3 = sync()
        else:
            t2 = t3 = t1
        a1 = torch.cuda.memory_stats().get("num_device_alloc", 0)
        rows.append((round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2), round((t3 - t2) * 1e3, 2), a1 - a0))
    print(rep, rows)
