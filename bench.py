#!/usr/bin/env python3
"""bench.py — audio-seconds/sec (RTF^-1) of the ASR-inference hot path on MI355X.

    python bench.py --gpus 1 --steps 2000 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic utterances already resident in
HBM: HIP STFT/log-mel frontend -> HIP Conformer encoder (fused block kernels) -> greedy CTC (G1)
decode -> the hypotheses of ALL ranks collated by `espnet_amd.distributed.RecordRing` (fixed-shape records
written in place into a device ring; ONE RCCL all-gather when N > 1 and ONE pinned, asynchronous device->host
copy per 16 steps, delivered while the next steps run, all K delivered inside the timed region).  Round 6: `--in-flight 2`
(default) deals the K steps to two HIP streams - utterance batches are independent, every kernel takes the whole chip, and one
stream's launch fills the boundary between two dependent launches of the other (`StepPipeline`; `one_stream` in the line is the
same loop with one batch in flight); `ms_per_step` = timed region / K, i.e. per batch.  Workload =
BASELINE.json configs[1]: Conformer-small (12 x 256d, 4 heads, ff 1024), batch 32 x 10 s @ 16 kHz per
GPU (weak scaling: per-GPU batch fixed).  Random-init weights (torch.manual_seed(0)), synthetic
N(0, 0.1^2) waveforms.

One JSON line on rank 0 carries the throughput and, at N = 1: `roofline` (the dominant MFMA kernel family
measured live with HIP events around every launch, HBM traffic from two `rocprofv3 --pmc` passes of this
same script and - `rocprofv3_*` fields - the un-bracketed kernel table of a nested `rocprofv3 --kernel-trace
--stats` run of the same loop), `cpu_baseline` (the oracle port on this box's host cores + the reference figure it stands in
for), `pcie_inclusive` (waveforms arriving in pinned host memory, H2D overlapped with compute),
`f32_mode`, `bf16_vs_f32` (id mismatch rate of the timed mode on the bench batch), `frontend` (GB/s vs
HBM peak), `encoder_large_b64` (the Conformer-large encoder at configs[3]'s per-GPU batch with its MFMA families and the HBM
traffic of its dominant one), `encoder_ebranchformer_b32` (the E-Branchformer encoder with its roofline),
`beam` (configs[2] with the roofline of the search and `bf16_vs_oracle`: per-token error, best-score loss beside the
oracle search's own path noise, token edit distance), `beam_cfg3_per_gpu` (configs[3]'s per-GPU batch on one GPU,
with `bf16_vs_oracle`; the beam legs with eight (configs[2]) and four (configs[3] per GPU) joint searches in flight, a host thread each: `SearchLanes`) and `stream` (configs[4] + the 40 ms-per-call
stress case); `box_state` (three probes of the pool's slow state: a slow lease is labelled, not read as a regression).  `--quick`
keeps only the main line, `roofline` and `cpu_baseline`.
"""
import argparse
import csv
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

AUDIO_SEC = 10.0
N_SAMPLES = 160000
VOCAB = 5000
MFMA_PEAK_TFLOPS = {"bfloat16": 2500.0, "float32": 157.3}  # /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
RING_STEPS = 16  # steps per collation macro-batch (espnet_amd.distributed.RecordRing)
FRONTEND_BYTES_PER_UTT = 160000 * 4 + 1001 * 80 * 4  # wave in + log-mel out (f32)
# the reference's own Speech2Text on CPU (the path cpu_baseline stands in for)
REFERENCE_MEASURED = {  # 8 vCPU Xeon @ 2.1 GHz survey container, torch CPU fp32, one 10 s utterance
    "greedy": {"value": 2.2, "unit": "audio-s/s", "seconds_per_utt": 4.51,
               "what": "reference Speech2Text(device=cpu, float32, ctc_weight=1.0, beam_size=1), Conformer-small: its "
                       "G1 route is a width-1 CTC prefix search of 249 steps; encode() + ctc.argmax alone (what the "
                       "port times) took 64-133 ms = 75-156 audio-s/s (BASELINE.md §2)"},
    "beam": {"value": 0.73, "unit": "audio-s/s", "seconds_per_utt": 13.8,
             "what": "reference Speech2Text(device=cpu, float32, beam_size=10, ctc_weight=0.3), Conformer-large + "
                     "6-layer decoder, 249 steps (BASELINE.md §2)"},
    "stream": {"value": 1.76, "unit": "audio-s/s", "seconds_per_utt": 5.69,
               "what": "reference Speech2TextStreaming(device=cpu), contextual_block_conformer 12x256d, 16 chunks of "
                       "640 ms, CTC-only beam 1; encoder forward_infer alone 0.44 s = 23 audio-s/s (BASELINE.md §2)"},
}

CONFIGS = {
    "small": dict(d=256, heads=4, ff=1024, win_length=400),
    "large": dict(d=512, heads=8, ff=2048, win_length=None),
    # E-Branchformer of egs2/librispeech/asr1/conf/tuning/train_asr_e_branchformer.yaml
    "ebf": dict(d=512, heads=8, ff=1024, win_length=None, ebf=dict(cg=3072, blocks=17, merge=31)),
}


def model_config(name, dtype):
    c = CONFIGS[name]
    fconf = dict(n_fft=512, hop_length=160)
    if c["win_length"]:
        fconf["win_length"] = c["win_length"]
    if "ebf" in c:
        e = c["ebf"]
        enc = dict(encoder="e_branchformer",
                   encoder_conf=dict(output_size=c["d"], attention_heads=c["heads"], linear_units=c["ff"],
                                     num_blocks=e["blocks"], input_layer="conv2d", rel_pos_type="latest",
                                     pos_enc_layer_type="rel_pos", attention_layer_type="rel_selfattn",
                                     cgmlp_linear_units=e["cg"], cgmlp_conv_kernel=31, use_linear_after_conv=False,
                                     gate_activation="identity", use_ffn=True, macaron_ffn=True,
                                     ffn_activation_type="swish", merge_conv_kernel=e["merge"]))
        return dict(token_list=["<blank>", "<unk>"] + [f"t{i}" for i in range(VOCAB - 3)] + ["<sos/eos>"],
                    frontend="default", frontend_conf=fconf, normalize="utterance_mvn", normalize_conf={},
                    decoder="transformer",
                    decoder_conf=dict(attention_heads=c["heads"], linear_units=2048, num_blocks=6),
                    model_conf=dict(ctc_weight=0.3), compute_dtype=dtype, **enc)
    return dict(
        token_list=["<blank>", "<unk>"] + [f"t{i}" for i in range(VOCAB - 3)] + ["<sos/eos>"],
        frontend="default", frontend_conf=fconf, normalize="utterance_mvn", normalize_conf={},
        encoder="conformer",
        encoder_conf=dict(output_size=c["d"], attention_heads=c["heads"], linear_units=c["ff"],
                          num_blocks=12, input_layer="conv2d", normalize_before=True,
                          macaron_style=True, rel_pos_type="latest", pos_enc_layer_type="rel_pos",
                          selfattention_layer_type="rel_selfattn", activation_type="swish",
                          use_cnn_module=True, cnn_module_kernel=31),
        decoder="transformer",
        decoder_conf=dict(attention_heads=c["heads"], linear_units=2048, num_blocks=6),
        model_conf=dict(ctc_weight=0.3), compute_dtype=dtype)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def synth_batch(first_utt, batch):
    wav = torch.empty(batch, N_SAMPLES)
    for i in range(batch):
        g = torch.Generator().manual_seed(1000 + first_utt + i)
        wav[i] = torch.randn(N_SAMPLES, generator=g) * 0.1
    return wav


def cpu_baseline(model, budget_s=12.0):
    """Oracle port (oracle/conformer.py = CPU-fp32 restatement of the reference path) timed on
    the host cores, batch 1 (the reference's own inference batch, asr_inference.py:760-765)."""
    from oracle import conformer as oc

    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    enc = model.encoder
    fe = model.frontend
    # oversubscribing a many-core host makes torch-CPU slower, not faster: cap at 32 threads of
    # the cores this process may run on, and report exactly the thread count used.
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(32, avail))
    torch.set_num_threads(cores)
    wl = fe.win_length
    encode_fn = oc.encode
    if type(enc).__name__ == "EBranchformerEncoder":
        from oracle import ebranchformer as oe

        encode_fn = oe.encode
    times = []
    t_start = time.perf_counter()
    i = 0
    with torch.no_grad():
        while True:
            wav = synth_batch(9000 + i, 1)
            t0 = time.perf_counter()
            e, ol = encode_fn(sd, wav, torch.tensor([N_SAMPLES]), enc.heads, enc.num_blocks, 512, wl, 160)
            oc.greedy_ctc(sd, e, ol, blank=0, sos_eos=VOCAB - 1)
            dt = time.perf_counter() - t0
            if i > 0:
                times.append(dt)
            i += 1
            if (time.perf_counter() - t_start > budget_s and len(times) >= 3) or len(times) >= 200:
                break
    med = sorted(times)[len(times) // 2]
    return {"value": round(AUDIO_SEC / med, 2), "unit": "audio-s/s", "cores": cores, "kind": "port",
            "cpu_model": _cpu_model(), "reference_measured": REFERENCE_MEASURED["greedy"],
            "sample": f"oracle CPU-fp32 port, {len(times)} utterances of 10 s, batch 1, median "
                      f"{med*1e3:.1f} ms/utt (frontend + encoder + greedy CTC G1), 1 warm-up"}


def cpu_baseline_beam(model, beam, ctc_weight, budget_s=15.0):
    """Oracle port of the reference Speech2Text beam search (oracle/beam_search.py; K/V-cached,
    i.e. FASTER than the reference's own CPU path, which measured 13.3 s/utt here) on host cores."""
    from oracle import beam_search as ob
    from oracle import conformer as oc

    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    enc, dec, fe = model.encoder, model.decoder, model.frontend
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(32, avail))
    torch.set_num_threads(cores)
    times = []
    i = -1  # (utterance -1 is the warm-up: thread pool, allocator, first-touch of the weights; not timed)
    t_start = None
    with torch.no_grad():
        while True:
            wav = synth_batch(9000 + i, 1)
            t0 = time.perf_counter()
            e, ol = oc.encode(sd, wav, torch.tensor([N_SAMPLES]), enc.heads, enc.num_blocks, 512,
                              fe.win_length, 160)
            ob.beam_search(sd, e[0, : int(ol[0])], dec.heads, dec.num_blocks, beam, ctc_weight,
                           sos=VOCAB - 1, eos=VOCAB - 1)
            if i >= 0:
                times.append(time.perf_counter() - t0)
            else:
                t_start = time.perf_counter()
            i += 1
            if len(times) >= 3 and (time.perf_counter() - t_start > budget_s or len(times) >= 20):
                break
    med = sorted(times)[len(times) // 2]
    return {"value": round(AUDIO_SEC / med, 3), "unit": "audio-s/s", "cores": cores, "kind": "port",
            "cpu_model": _cpu_model(), "reference_measured": REFERENCE_MEASURED["beam"],
            "sample": f"oracle CPU-fp32 port (K/V-cached restatement of Speech2Text beam search), "
                      f"{len(times)} utterances of 10 s, batch 1, median {med:.2f} s/utt "
                      f"(frontend + encoder + beam {beam} search, 249 steps), one untimed warm-up utterance"}


def beam_vs_oracle(model, enc_row, hyps, beam, ctc_weight, noise_seeds=4, noise_sigma=2e-3):
    """Parity of the TIMED search mode on one utterance of the bench batch, with the oracle as the checker (part of the
    cpu_baseline leg: host cores, never the thing measured): every hypothesis the device returned re-scored
    teacher-forced under the oracle's f32 scorers over the encoder rows the device search consumed
    (oracle.beam_search.rescore_batch), the oracle's own f32 search over the same rows, and - `noise_seeds` > 0 - the
    oracle search's OWN path noise at the device's per-entry error level (oracle.beam_search.path_noise_losses: the
    yardstick tests/test_gpu_fullsize.py::test_beam10_b16_rows_bf16_vs_oracle holds `best_score_loss` against)."""
    from oracle import beam_search as ob

    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    dec = model.decoder
    e = enc_row.float().cpu()
    ys = [h.yseq.tolist() for h in hyps]
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = ob.rescore_batch(sd, e, ys, dec.heads, dec.num_blocks, ctc_weight, VOCAB - 1)
        orc = ob.beam_search(sd, e, dec.heads, dec.num_blocks, beam, ctc_weight, sos=VOCAB - 1, eos=VOCAB - 1)
        losses = None
        if noise_seeds > 0:
            losses, _ = ob.path_noise_losses(sd, e, dec.heads, dec.num_blocks, beam, ctc_weight, VOCAB - 1, noise_sigma,
                                             list(range(noise_seeds)), clean=orc)
    err = {k: max(abs(float(h.scores[k]) - r[k]) / max(1, r["n_scored"]) for h, r in zip(hyps, ref))
           for k in ("decoder", "ctc")}
    mine = {tuple(y) for y in ys}
    kb = max(range(len(ref)), key=lambda k: ref[k]["score"])
    best = ref[kb]["score"]
    out = {"utterance": 0, "hypotheses_rescored": len(ys),
           "max_abs_err_per_token": {k: round(v, 6) for k, v in err.items()},
           "device_best_oracle_score": round(best, 4), "oracle_best_score": round(orc[0]["score"], 4),
           "best_score_loss": round(orc[0]["score"] - best, 4),
           "token_edit_distance": ob.token_edit_distance(ys[kb][1:-1], orc[0]["yseq"][1:-1]),
           "oracle_best_tokens": len(orc[0]["yseq"]) - 2,
           "oracle_nbest_span": round(orc[0]["score"] - orc[-1]["score"], 4),
           "oracle_hypotheses_in_device_nbest": sum(tuple(o["yseq"]) in mine for o in orc),
           "oracle_hypotheses": len(orc)}
    if losses is not None:
        out["oracle_path_noise"] = {"sigma": noise_sigma, "seeds": noise_seeds, "best_score_losses": [round(x, 3) for x in losses],
                                    "what": "the ORACLE's own search with N(0, sigma^2) on every log-probability, its best "
                                            "re-scored without noise: what a perturbation of the device's size costs the "
                                            "search itself (negative = the perturbed search ended better than the clean one)"}
    out["checker_seconds"] = round(time.perf_counter() - t0, 1)
    out["what"] = ("timed dtype against the oracle's f32 scorers on utterance 0 of the batch: per scored token and "
                   "scorer |device - oracle| along the device's own token paths; best_score_loss = oracle best minus the "
                   "oracle's score of the device's best (what reduced-precision pruning lost; random-init posteriors "
                   "are flat: the oracle's own n-best span and path noise are beside it); token_edit_distance between "
                   "the device's best and the oracle's best hypothesis; exact n-best on peaked posteriors is asserted "
                   "by tests/test_gpu_search.py::test_search_bf16_peaked_returns_reference_nbest_exactly")
    return out


def run_stream(dtype, steps, warmup, chunk=10240, stream_beam=1, cpu_base=True):
    """BASELINE.json configs[4]: streaming contextual-block Conformer (aishell recipe shape: 12 x
    256d, 4 heads, ff 2048, conv k 15, block 40 / hop 16 / look-ahead 16), one audio stream fed in
    chunks of `chunk` samples (10 240 = 640 ms, the reference's sim_chunk_length; 640 = the 40 ms-per-call
    stress case) through Speech2TextStreaming (HIP frontend -> hipGraph-captured encoder step ->
    incremental greedy CTC).  A step = one 10 s utterance."""
    import yaml

    from espnet_amd.bin.asr_inference_streaming import Speech2TextStreaming

    enc_conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12,
                    input_layer="conv2d", normalize_before=True, activation_type="swish",
                    macaron_style=True, use_cnn_module=True, cnn_module_kernel=15, block_size=40,
                    hop_size=16, look_ahead=16, init_average=True, ctx_pos_enc=True)
    cfg = dict(token_list=["<blank>", "<unk>"] + [f"t{i}" for i in range(VOCAB - 3)] + ["<sos/eos>"],
               frontend="default", frontend_conf=dict(n_fft=512, hop_length=160, win_length=400),
               normalize="utterance_mvn", normalize_conf={}, encoder="contextual_block_conformer",
               encoder_conf=enc_conf, decoder="transformer",
               decoder_conf=dict(attention_heads=4, linear_units=2048, num_blocks=6),
               model_conf=dict(ctc_weight=0.3))
    torch.manual_seed(0)
    with tempfile.TemporaryDirectory() as td:
        (Path(td) / "config.yaml").write_text(yaml.safe_dump(cfg))
        s2t = Speech2TextStreaming(str(Path(td) / "config.yaml"), None, device="cuda", dtype=dtype,
                                   beam_size=stream_beam, ctc_weight=0.3)
    wav = synth_batch(0, 1)[0]
    chunks = [wav[p : p + chunk] for p in range(0, N_SAMPLES, chunk)]

    def step():
        lat = []
        for k, c in enumerate(chunks):
            t0 = time.perf_counter()
            out = s2t(c, is_final=(k == len(chunks) - 1))  # ends with a host read of the new tokens
            lat.append(time.perf_counter() - t0)
        return out, lat

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lats = []
    for _ in range(steps):
        out, lat = step()
        lats += lat[2:-1]  # steady-state calls (the first two only buffer, the last is final)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lats.sort()
    res = {"value": round(AUDIO_SEC * steps / elapsed, 1), "unit": "audio-s/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(elapsed / steps * 1e3, 3), "dtype": "bf16" if dtype == "bfloat16" else "f32",
           "config": {"workload": "BASELINE.json configs[4]: streaming contextual_block_conformer (12x256d, "
                                  "block 40 / hop 16 / look-ahead 16), ONE stream, hipGraph-captured encoder step, "
                                  + ("incremental greedy CTC" if stream_beam <= 1 else
                                     "block-synchronous online beam search"),
                      "chunk_ms": chunk / 16.0, "calls_per_utt": len(chunks),
                      "call_latency_ms_median": round(lats[len(lats) // 2] * 1e3, 3),
                      "call_latency_ms_p95": round(lats[int(len(lats) * 0.95)] * 1e3, 3),
                      "realtime_factor_of_one_stream": round(elapsed / steps / AUDIO_SEC, 5),
                      "hipgraph_replays": s2t._runner.n_replays if s2t._runner else 0,
                      **({"search": f"BatchBeamSearchOnline beam {stream_beam}, ctc_weight 0.3",
                          "search_steps_per_utt": s2t.beam_search.n_steps // (steps + warmup)}
                         if stream_beam > 1 else {}),
                      "tokens_last_utt": len(out[0][2]) if out else 0}}
    if cpu_base:
        res["cpu_baseline"] = cpu_baseline_stream(s2t.asr_model, enc_conf)
    return res


def run_stream_batch(dtype, n_streams, steps, warmup, chunk=10240, groups=1, graph=False):
    """configs[4] as a SERVER runs it: n_streams live connections fed 640 ms chunks in lock step, one launch sequence
    per tick for all of them (Speech2TextStreaming.batch_call: batched HIP frontend -> forward_infer_batch -> greedy
    CTC, one device -> host read per tick).  A step = n_streams utterances of 10 s."""
    import yaml

    from espnet_amd.bin.asr_inference_streaming import Speech2TextStreaming

    enc_conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12,
                    input_layer="conv2d", normalize_before=True, activation_type="swish",
                    macaron_style=True, use_cnn_module=True, cnn_module_kernel=15, block_size=40,
                    hop_size=16, look_ahead=16, init_average=True, ctx_pos_enc=True)
    cfg = dict(token_list=["<blank>", "<unk>"] + [f"t{i}" for i in range(VOCAB - 3)] + ["<sos/eos>"],
               frontend="default", frontend_conf=dict(n_fft=512, hop_length=160, win_length=400),
               normalize="utterance_mvn", normalize_conf={}, encoder="contextual_block_conformer",
               encoder_conf=enc_conf, decoder="transformer",
               decoder_conf=dict(attention_heads=4, linear_units=2048, num_blocks=6),
               model_conf=dict(ctc_weight=0.3))
    torch.manual_seed(0)
    with tempfile.TemporaryDirectory() as td:
        (Path(td) / "config.yaml").write_text(yaml.safe_dump(cfg))
        s2t = Speech2TextStreaming(str(Path(td) / "config.yaml"), None, device="cuda", dtype=dtype, beam_size=1,
                                   ctc_weight=0.3, use_hipgraph=graph)  # (graph: the steady-state tick of a group as one hipGraph, BatchTickGraph - measured no faster
                                   # than the eager launches, 1.03 - 1.04 against 1.015 ms per tick of 32 streams: profiles/r06y_stream_tick_graph.txt)
    wav = synth_batch(0, n_streams)  # (S, N) host
    bounds = [(p, min(N_SAMPLES, p + chunk)) for p in range(0, N_SAMPLES, chunk)]
    # a tick's audio arrives in pinned host memory (what a server's receive buffers are): from pageable memory the
    # 1.3 MB copies of 32 streams stall ~90 ms every few ticks inside the HIP runtime (profiles/r03q_stream_batch_ticks.txt)
    ticks = [wav[:, lo:hi].contiguous().pin_memory() for lo, hi in bounds]

    # groups > 1: that many independent sets of n_streams lock-step streams, each on a HIP stream of its own, a tick of each in
    # flight (Speech2TextStreaming.batch_call_async): while the host reads group g's ids and prepares its next tick, the other
    # groups' launches keep the device busy.  A "tick latency" is then submit -> ids on the host of one group's tick.
    G = groups
    sts = StepPipeline._pick(torch.device("cuda", torch.cuda.current_device()), G)[0] if G > 1 else [None]

    def step():
        lat, pend, t_sub, out = [], [None] * G, [0.0] * G, None
        for k in range(len(bounds)):
            for g in range(G):
                if pend[g] is not None:
                    out = pend[g].result()  # ends with a host read of the new ids
                    lat.append(time.perf_counter() - t_sub[g])
                t_sub[g] = time.perf_counter()
                pend[g] = s2t.batch_call_async(ticks[k], is_final=(k == len(bounds) - 1), group=g, stream=sts[g])
        for g in range(G):
            out = pend[g].result()
            lat.append(time.perf_counter() - t_sub[g])
        return out, lat

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lats = []
    for _ in range(steps):
        out, lat = step()
        lats += lat[2 * G : -G]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lats.sort()
    n_streams_all = n_streams * G
    return {"value": round(n_streams_all * AUDIO_SEC * steps / elapsed, 1), "unit": "audio-s/s", "streams": n_streams_all,
            "groups_in_flight": G, "streams_per_tick": n_streams,
            "hipgraph_replays": sum(tg.n_replays for tg in s2t._tick_graphs.values()),
            "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 3),
            "tick_latency_ms_median": round(lats[len(lats) // 2] * 1e3, 3),
            "tick_latency_ms_p95": round(lats[int(len(lats) * 0.95)] * 1e3, 3),
            "chunk_ms": chunk / 16.0, "ticks_per_utt": len(bounds),
            "realtime_multiple": int(n_streams * (chunk / 16.0) / (lats[len(lats) // 2] * 1e3)),  # audio ms per wall ms of a tick
            "tokens_stream0": len(out[0]) if out else 0,
            "what": f"{n_streams} lock-step streams per tick, 640 ms chunks, one launch sequence per tick for all of them "
                    f"({'the steady-state tick as one hipGraph' if graph else 'eager launches'}, waveform chunks from pinned host memory, greedy CTC ids read back once per tick)"
                    + (f"; {G} such groups, a tick of each in flight on its own HIP stream" if G > 1 else "")}


def main_stream(args):
    assert int(os.environ.get("WORLD_SIZE", "1")) == 1, "--workload stream is single-stream"
    torch.cuda.set_device(0)
    res = run_stream(args.dtype, args.steps, args.warmup, args.stream_chunk, args.stream_beam,
                     cpu_base=not args.no_cpu_baseline)
    if args.stream_beam <= 1 and args.stream_chunk == 10240:
        res["batch32"] = run_stream_batch(args.dtype, 32, max(1, args.steps), 1)
        res["batch128"] = run_stream_batch(args.dtype, 128, max(1, min(args.steps, 4)), 1)
        res["batch32_hipgraph"] = run_stream_batch(args.dtype, 32, max(1, args.steps), 1, graph=True)
        res["batch32_two_groups"] = run_stream_batch(args.dtype, 32, max(1, args.steps), 1, groups=2)
        res["batch128_two_groups"] = run_stream_batch(args.dtype, 128, max(1, min(args.steps, 4)), 1, groups=2)
        res["batch64_three_groups"] = run_stream_batch(args.dtype, 64, max(1, min(args.steps, 4)), 1, groups=3)
    res = {"metric": "audio-seconds/sec (RTF^-1), Conformer-ASR, 10 s utterances", **res, "n_gpus": 1,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic"}
    print(json.dumps(res), flush=True)


def cpu_baseline_stream(model, enc_conf, budget_s=8.0):
    from oracle import conformer as oc
    from oracle.streaming import CBEncoderOracle

    sd = {k[len("encoder."):]: v.detach().float().cpu() for k, v in model.state_dict().items()
          if k.startswith("encoder.")}
    mel = model.frontend.logmel.melmat.detach().float().cpu()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(32, avail))
    torch.set_num_threads(cores)
    orc = CBEncoderOracle(sd, enc_conf["attention_heads"], enc_conf["num_blocks"], enc_conf["block_size"],
                          enc_conf["hop_size"], enc_conf["look_ahead"])
    times, t_start, i = [], time.perf_counter(), 0
    with torch.no_grad():
        while time.perf_counter() - t_start < budget_s or len(times) < 2:
            wav = synth_batch(9000 + i, 1)
            t0 = time.perf_counter()
            f, fl = oc.frontend_feats(wav, torch.tensor([N_SAMPLES]), mel, 512, 400, 160)
            f = oc.utterance_mvn(f, fl)[0]
            state, pos = None, 0
            while pos < f.size(0):
                nxt = min(f.size(0), pos + 64)
                _, state = orc.forward_infer(f[pos:nxt], state, nxt == f.size(0))
                pos = nxt
            times.append(time.perf_counter() - t0)
            i += 1
    med = sorted(times)[len(times) // 2]
    return {"value": round(AUDIO_SEC / med, 2), "unit": "audio-s/s", "cores": cores, "kind": "port",
            "cpu_model": _cpu_model(), "reference_measured": REFERENCE_MEASURED["stream"],
            "sample": f"oracle CPU-fp32 port of the streaming encoder (frontend + 64-frame chunks through "
                      f"forward_infer, no decoding), {len(times)} utterances of 10 s, median {med:.2f} s/utt"}


def pick_copy_stream(dev, host, bufs, candidates=6, beside=None, trial_steps=24):
    """A copy stream that really runs beside the compute stream(s).  HIP deals its streams onto a handful of hardware queues, and
    a copy stream that shares a queue with a stream the steps (or the record ring's flush) run on is executed IN that queue's
    order: the copy of a later batch then starts when the steps submitted before it have finished, and the whole copy (0.42 ms
    for 20.5 MB) lands in the step.  Which queue a new stream gets is not in the API; profiles/r06g_h2d_probe.txt shows the same
    feeder at 1.04 and at 1.39 ms per step in one process, depending on the streams created before it.  So every candidate
    runs a short trial of the feeder's OWN event pattern (acquire -> ~0.5 ms of kernels on the step's stream -> release ->
    prefetch, steps dealt to the streams of `beside` as the timed loop deals them) and the stream with the shortest trial
    stays.  (Round 6, first form: a copy timed beside free-running kernels - it did not see the bad pairings of the pipelined
    loop, where the copy stream must avoid three other queues.)"""
    beside = list(beside) if beside else [torch.cuda.current_stream()]
    work = [torch.empty(48 << 20, dtype=torch.float32, device=dev) for _ in beside]

    def trial(st):
        fd = HostFeeder.__new__(HostFeeder)
        fd._init(host, bufs, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(trial_steps):
            x = beside[k % len(beside)]
            with torch.cuda.stream(x):
                fd.acquire()
                for _ in range(4):
                    work[k % len(beside)].mul_(1.0)
                fd.release()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / trial_steps

    best, report = None, []
    for _ in range(candidates):
        st = torch.cuda.Stream(device=dev)
        trial(st)
        t = min(trial(st) for _ in range(2))
        report.append(round(t * 1e3, 3))
        if best is None or t < best[0]:
            best = (t, st)
    del work
    return best[1], {"trial_ms_per_step_per_candidate": report}


class StepPipeline:
    """`depth` batches in flight on as many HIP streams (round 6).  Every kernel of the greedy step takes the whole chip, so
    two steps never run side by side - but between two DEPENDENT launches of one stream the chip idles for the boundary
    (1.5 - 1.9 us, ~27 per step: drain, signal, dispatch), and the other stream's next launch fills it: 0.905 against 0.983 ms
    per batch with two streams, tools/two_stream_probe.py / profiles/r06k_two_stream_probe.txt.  Utterance batches are
    independent (that is the data-parallel story of the whole path); the encoder keeps one workspace per stream.
    The streams are chosen so that they really sit on different hardware queues (two single-thread spin kernels must run
    side by side): see pick_copy_stream for why that is not a given."""

    def __init__(self, dev, depth):
        self.dev, self.depth, self.i = dev, depth, 0
        self.streams, self.probe = self._pick(dev, depth)
        self.done = [None] * depth
        self.flush_ev = {}

    @staticmethod
    def _pick(dev, depth, candidates=8, spin=400000):
        def wall(sts):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for st in sts:
                with torch.cuda.stream(st):
                    torch.cuda._sleep(spin)
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        cand = [torch.cuda.Stream(device=dev) for _ in range(max(candidates, depth))]
        main = torch.cuda.current_stream()  # (the record ring is flushed there, behind the step streams: keep its queue apart too)
        one = min(wall([main]) for _ in range(3))
        chosen = []
        for st in cand:
            if len(chosen) == depth:
                break
            if min(wall([main] + chosen + [st]) for _ in range(2)) < 1.35 * one:  # runs beside every stream chosen so far
                chosen.append(st)
        note = {"streams_side_by_side": len(chosen), "asked": depth}
        while len(chosen) < depth:  # (no such stream found: still correct, just no overlap)
            chosen.append(cand[len(chosen)])
        return chosen, note

    def fork(self, half=None):
        """Before the first step of a macro-batch: every stream waits for the flush that last read the ring half it is about to
        write (recorded by `flushed(half)` a whole macro-batch ago - no stall), or, without one, for whatever the caller's stream
        has enqueued so far."""
        ev = self.flush_ev.get(half) if half is not None else None
        if ev is None:
            ev = torch.cuda.Event()
            ev.record()
        for st in self.streams:
            st.wait_event(ev)

    def flushed(self, half):
        """The caller's stream has just enqueued the flush (all-gather + D2H) of ring half `half`."""
        ev = self.flush_ev.get(half)
        if ev is None:
            ev = self.flush_ev[half] = torch.cuda.Event()
        ev.record()

    def run(self, fn):
        k = self.i % self.depth
        with torch.cuda.stream(self.streams[k]):
            fn()
            if self.done[k] is None:
                self.done[k] = torch.cuda.Event()
            self.done[k].record()
        self.i += 1

    def join(self):
        """The caller's stream waits for every step enqueued so far."""
        cur = torch.cuda.current_stream()
        for ev in self.done:
            if ev is not None:
                cur.wait_event(ev)


class HostFeeder:
    """Waveforms arriving in pinned host memory (the boundary's real input): with `nbuf` device buffers the copy of batch
    k + nbuf - 1 is issued on a copy stream when batch k is released, i.e. it runs under the compute of the nbuf - 1 batches in
    front of it (nbuf = batches in flight + 1)."""

    def __init__(self, wav_host, dev, nbuf=2, beside=None):
        host = wav_host.pin_memory()
        bufs = [torch.empty_like(wav_host, device=dev) for _ in range(nbuf)]
        st, self.stream_probe = pick_copy_stream(dev, host, bufs, beside=beside)
        self._init(host, bufs, st)

    def _init(self, host, bufs, copy_stream):
        self.host, self.bufs, self.n, self.copy_stream = host, bufs, len(bufs), copy_stream
        self.copied = [torch.cuda.Event() for _ in bufs]
        self.consumed = [None] * self.n
        self._consumed_ev = [torch.cuda.Event() for _ in bufs]  # (re-recorded every step: no event creation in the loop)
        self.k = 0
        for slot in range(self.n - 1):
            self._prefetch(slot)

    def _prefetch(self, slot):
        with torch.cuda.stream(self.copy_stream):
            if self.consumed[slot] is not None:
                self.copy_stream.wait_event(self.consumed[slot])
            self.bufs[slot].copy_(self.host, non_blocking=True)
            self.copied[slot].record(self.copy_stream)

    def acquire(self):
        slot = self.k % self.n
        torch.cuda.current_stream().wait_event(self.copied[slot])
        return self.bufs[slot]

    def release(self):
        slot = self.k % self.n
        ev = self._consumed_ev[slot]
        ev.record()
        self.consumed[slot] = ev
        self.k += 1
        self._prefetch((self.k + self.n - 2) % self.n)


def _edit_distance(a, b):
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def timed_loop(step, steps, warmup, barrier, finish=None):
    with torch.no_grad():
        for _ in range(warmup):
            step()
        if finish:
            finish()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        if finish:
            finish()
        barrier()
        return time.perf_counter() - t0


def decoder_step_bytes(model, B, W, T, NC, steps, es):
    """Algorithmic HBM bytes of ONE label step of the joint search (DESIGN.md §search roofline): every decoder
    weight once, the per-utterance memory K / V^T once, the self-attention K/V of the prefixes (average length
    steps/2), the logits written and read, the CTC log-prob columns of the NC candidates and the forward
    variables read + written."""
    dec = model.decoder
    d, ff, L, V = dec.d, dec.linear_units, dec.num_blocks, dec.vocab_size
    n = B * W
    w = L * (3 * d * d + d * d + d * d + d * d + 2 * d * ff) * es + d * V * es  # src K/V projections run at init
    mem = B * L * T * 2 * d * es
    cache = n * L * (steps / 2.0) * 2 * d * es
    logits = n * V * 4 * 2
    ctc = n * NC * T * 4 + n * T * 8 * 2
    return w + mem + cache + logits + ctc


def run_beam(args, dev, B, beam, steps, warmup, cpu_base, want_traffic=True, want_oracle=False):
    """configs[2] / configs[3]'s per-GPU batch: Conformer-large + 6-layer decoder, joint CTC/attention beam search."""
    from espnet_amd import distributed as D
    from espnet_amd.nets.batch_beam_search import build_beam_search
    from espnet_amd.tasks.asr import ASRTask

    torch.manual_seed(0)
    model = ASRTask.build_model(model_config("large", args.dtype)).to(dev).eval()
    bs = build_beam_search(model, beam_size=beam, ctc_weight=args.ctc_weight, penalty=0.0,
                           token_list=model.token_list)
    if os.environ.get("BENCH_ENC_IN_FLIGHT"):  # developer probe: what the encoder call is told about batches in flight (EM_ENC_IN_FLIGHT)
        model.encoder.batches_in_flight = int(os.environ["BENCH_ENC_IN_FLIGHT"])
    if os.environ.get("BENCH_SEARCH_EAGER"):  # developer probe: label steps as eager launches from em_search_steps, no hipGraph
        bs.use_hipgraph = False
    if os.environ.get("BENCH_STEP_CHUNK"):  # developer probe: label steps enqueued between two polls of the `done` flags
        bs.step_chunk = int(os.environ["BENCH_STEP_CHUNK"])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    wav = synth_batch(rank * B, B).to(dev)
    lens = [N_SAMPLES] * B
    T = model.encoder.output_frames(1 + N_SAMPLES // 160)
    t_search = [0.0, 0]
    last = {}

    def step(instrument=False):
        st = model.encode_device(wav, lens)
        if instrument:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        nbest = bs.search_batch(st.enc_act, st.olens)
        if instrument:
            t_search[0] += time.perf_counter() - t0
            t_search[1] += bs.last_steps
            last["nbest0"], last["enc0"] = nbest[0], st.enc_act[0, : int(st.olens[0])].clone()
        toks = [[t for t in h[0].yseq[1:-1].tolist()] if h else [] for h in nbest]
        sc = [float(h[0].score) if h else 0.0 for h in nbest]
        return toks, sc

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # N > 1: DYNAMIC dispatch - the job is world * steps batches, ranks pull batch indices from a shared counter
    # (espnet_amd.distributed.decode_dynamic), records carry their global utterance index and are collated once at the
    # end of the timed region: a slow GPU or early-ending beams cost the job their own share only.  N = 1: the same
    # loop, the counter is local.
    run_id = run_beam.calls = getattr(run_beam, "calls", 0) + 1
    depth = max(1, int(getattr(args, "in_flight", 1)))
    lanes = None
    if depth > 1:
        # Round 6: `depth` batches in flight, one joint search per HIP stream (espnet_amd.nets.batch_beam_search.SearchLanes):
        # a label step is ~47 small launches on a fraction of the chip - a second search runs in the gaps of the first, and the
        # host-side readout of a finished search overlaps the other lane's steps.
        from espnet_amd.nets.batch_beam_search import SearchLanes

        # (lane streams that really sit on different hardware queues; BENCH_LANE_STREAMS=plain: whatever torch hands out, round 6's first form)
        lane_streams = None if os.environ.get("BENCH_LANE_STREAMS") == "plain" else StepPipeline._pick(dev, depth)[0]
        # (BENCH_LANE_THREADS=1 / args.lane_threads: a host thread per lane - the lanes' launches in parallel, see SearchLanes)
        lane_threads = bool(getattr(args, "lane_threads", False)) or os.environ.get("BENCH_LANE_THREADS") == "1"
        lanes = SearchLanes([bs] + [bs.clone() for _ in range(depth - 1)], dev, streams=lane_streams, threaded=lane_threads)

        def lane_start(k, u):
            with torch.cuda.stream(lanes.stream(k)):
                st = model.encode_device(wav, lens)
            lanes.start(k, st.enc_act, st.olens, tag=(u, st))

        def lane_poll(k):
            r = lanes.poll(k)
            if r is None:
                return None
            (u, _st), nbest = r
            toks = [[t for t in h[0].yseq[1:-1].tolist()] if h else [] for h in nbest]
            sc = [float(h[0].score) if h else 0.0 for h in nbest]
            return list(range(u * B, u * B + B)), toks, sc
    with torch.no_grad():
        for _ in range(warmup):
            step()
        if lanes is not None:  # every lane captures its hipGraph once, one lane at a time
            for k in range(depth):
                lane_start(k, 0)
                while lane_poll(k) is None:
                    pass
        barrier()
        counter = D.SharedCounter(D.work_store() if world > 1 else None, f"bench_beam_{run_id}")
        t0 = time.perf_counter()
        if lanes is None:
            hyps, mine = D.decode_dynamic(lambda u: (list(range(u * B, u * B + B)), *step()), world * steps,
                                          world * steps * B, T + 2, dev, counter=counter)
        else:
            hyps, mine = D.decode_dynamic_lanes(lane_start, lane_poll, depth, world * steps, world * steps * B, T + 2, dev,
                                                counter=counter)
        barrier()
        elapsed = time.perf_counter() - t0
        if lanes is not None:
            lanes.close()
        step(instrument=True)
    n_tok = sum(len(t) for t, _ in hyps[-B:])
    es = 2 if args.dtype == "bfloat16" else 4
    S = bs.pre_beam_size if bs.do_pre_beam else VOCAB
    n_steps = t_search[1]
    per_step = decoder_step_bytes(model, B, beam, T, S + 1, n_steps, es)
    res = {"model": model, "elapsed": elapsed, "tokens": n_tok, "steps_this_rank": len(mine),
           "search": {"ms_per_search_step": round(t_search[0] / max(1, n_steps) * 1e3, 4),
                      "search_steps_per_utt_batch": n_steps, "steps_per_s": round(n_steps / t_search[0], 1),
                      "rows": B * beam,
                      "batches_in_flight": depth, "host_threads": depth if (lanes is not None and lane_threads) else 1,
                      "what": "ms_per_search_step: ONE search alone on the chip (latency of a label step); `value` of this "
                              "object's parent: `batches_in_flight` searches on as many HIP streams",
                      "roofline": {"bound": "hbm", "achieved": round(per_step * n_steps / t_search[0] / 1e9, 1),
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(per_step * n_steps / t_search[0] / 1e9 / HBM_PEAK_GBS, 5),
                                   "traffic": None,
                                   "algorithmic_mb_per_search_step": round(per_step / 1e6, 2),
                                   "what": "decoder weights + memory K/V + self-attention cache + logits + CTC "
                                           "columns and forward variables per label step (bench.py "
                                           "decoder_step_bytes), over the wall time of search_batch"}}}
    inner = os.environ.get("ESPNET_AMD_BENCH_INNER") == "1"
    if (want_traffic and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not inner and not args.no_traffic
            and not getattr(args, "quick", False)):
        try:
            traffic, note = collect_search_traffic(args, B, beam)
        except Exception as e:  # noqa: BLE001 - the bench line must still be printed
            traffic, note = None, f"{type(e).__name__}: {e}"
        res["search"]["roofline"]["traffic"] = None if traffic is None else round(traffic)
        res["search"]["roofline"]["traffic_source"] = note
    if cpu_base:
        res["cpu_baseline"] = cpu_baseline_beam(model, beam, args.ctc_weight)
    if (cpu_base or want_oracle) and world == 1:
        try:
            key = "bf16_vs_oracle" if args.dtype == "bfloat16" else "f32_vs_oracle"
            res[key] = beam_vs_oracle(model, last["enc0"], last["nbest0"], beam, args.ctc_weight,
                                      noise_seeds=4 if args.dtype == "bfloat16" else 0)
        except Exception as e:  # noqa: BLE001 - a checker must not cost the line
            res["bf16_vs_oracle"] = {"error": f"{type(e).__name__}: {e}"}
    return res


def collect_traffic(kernel_substr, args):
    """HBM bytes per launch of the dominant kernel from two `rocprofv3 --pmc` passes (FETCH_SIZE, then WRITE_SIZE:
    they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") of this script's own main loop.
    gfx950 correction of that guide's HBM section: FETCH_SIZE counts 64 B per 128-B request -> doubled."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get(
            "LD_PRELOAD", ""):
        return None, "already running under a profiler: no nested rocprofv3 passes"
    out = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "-d", td, "-o", "t", "--output-format", "csv", "--",
                   sys.executable, str(REPO / "bench.py"), "--quick", "--no-roofline", "--no-cpu-baseline", "--in-flight", "1",
                   "--steps", "3", "--warmup", "2", "--dtype", args.dtype, "--batch", str(args.batch),
                   "--model", args.model]
            env = dict(os.environ, TMPDIR="/tmp", ESPNET_AMD_BENCH_INNER="1")
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {ctr}: timeout"
            files = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr}: rc {r.returncode}, {len(files)} csv"
            tot, n = 0.0, 0
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == ctr and kernel_substr in row["Kernel_Name"]:
                        tot += float(row["Counter_Value"])
                        n += 1
            if n == 0:
                return None, f"no {kernel_substr} dispatch in the {ctr} pass"
            out[ctr] = (tot / n, n)
    fetch_kb, write_kb = out["FETCH_SIZE"][0], out["WRITE_SIZE"][0]
    return (2.0 * fetch_kb + write_kb) * 1024.0, (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over "
                                                  f"{out['FETCH_SIZE'][1]} launches: FETCH_SIZE {fetch_kb:.1f} KB x 2 "
                                                  f"(gfx950 correction) + WRITE_SIZE {write_kb:.1f} KB per launch")


SEARCH_STEP_KERNELS = ("ln_gemm_kernel", "mid_gemm_kernel", "dec_self_attn_kernel", "dec_src_attn_kernel",
                       "dec_embed_kernel", "logsoftmax_prebeam_kernel", "tail_kernel", "candidate_kernel",
                       "select_kernel", "ctc_state_kernel", "update_kernel", "layernorm4_kernel", "Li6ELi0ELi64ELi128E")


def collect_search_traffic(args, B, beam):
    """HBM bytes per LABEL STEP of the beam search from two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE) of this
    script's own beam leg: the counters of every dispatch of the label step's kernels (SEARCH_STEP_KERNELS: decoder step,
    pre-beam, tail; the final LayerNorm and the vocabulary GEMM share their kernel templates with a few launches of the
    encoder / search initialisation, < 1 % of the sum) summed and divided by the number of label steps (= dispatches of
    the step's last kernel).  Same gfx950 correction as collect_traffic."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get(
            "LD_PRELOAD", ""):
        return None, "already running under a profiler: no nested rocprofv3 passes"
    out = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "-d", td, "-o", "t", "--output-format", "csv", "--",
                   sys.executable, str(REPO / "bench.py"), "--workload", "beam", "--no-cpu-baseline", "--in-flight", "1", "--steps", "1",
                   "--warmup", "0", "--dtype", args.dtype, "--batch", str(B), "--beam", str(beam),
                   "--ctc-weight", str(args.ctc_weight)]
            env = dict(os.environ, TMPDIR="/tmp", ESPNET_AMD_BENCH_INNER="1")
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=400)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {ctr}: timeout"
            files = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr}: rc {r.returncode}, {len(files)} csv"
            tot, n_steps = 0.0, 0
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != ctr:
                        continue
                    name = row["Kernel_Name"]
                    if any(k in name for k in SEARCH_STEP_KERNELS):
                        tot += float(row["Counter_Value"])
                    if "tail_kernel" in name or "update_kernel" in name:
                        n_steps += 1
            if n_steps == 0:
                return None, f"no label step in the {ctr} pass"
            out[ctr] = (tot / n_steps, n_steps)
    fetch_kb, write_kb = out["FETCH_SIZE"][0], out["WRITE_SIZE"][0]
    return (2.0 * fetch_kb + write_kb) * 1024.0, (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over "
                                                  f"{out['FETCH_SIZE'][1]} label steps: FETCH_SIZE {fetch_kb:.1f} KB x 2 "
                                                  f"(gfx950 correction) + WRITE_SIZE {write_kb:.1f} KB per label step")


def PROF_NAMES():
    from espnet_amd import lib as L

    return {L.EM_PROF_GEMM: "gemm_kernel<T,EPI,AMODE> (all instantiations)",
            L.EM_PROF_BLOCK: "block_kernel<MODE> (fused Conformer block, csrc/block.hip)",
            L.EM_PROF_ATTN: "relpos_attn2_kernel (csrc/attention2.hip)",
            L.EM_PROF_ROWS: "ffn_rows_kernel<LNMODE,PRE,MAIN> (row-block feed-forward / projection / GLU launches of the 512-wide "
                            "model, csrc/ffn_rows.hip)"}


def PROF_MATCH():
    from espnet_amd import lib as L

    return {L.EM_PROF_GEMM: "gemm_kernel", L.EM_PROF_BLOCK: "block_kernel", L.EM_PROF_ATTN: "relpos_attn2_kernel",
            L.EM_PROF_ROWS: "ffn_rows_kernel"}


def profile_families(step_fn, nprof):
    """HIP events recorded by the library around every launch of the MFMA kernel families (on the launch stream:
    em_profile_*), over `nprof` passes of `step_fn`.  Returns {family tag: [ms, algorithmic flops, launches]}."""
    from espnet_amd import lib as L

    lib = L.load()
    cap = 32768
    prof = lib.em_profile_create(cap)
    ms = (C.c_float * cap)()
    fl = (C.c_double * cap)()
    tg = (C.c_int32 * cap)()
    cnt = C.c_int32(0)
    fam = {}
    with torch.no_grad():
        for _ in range(nprof):
            lib.em_profile_attach(prof)
            step_fn()
            lib.em_profile_attach(None)
            L.check(lib.em_profile_read2(prof, ms, fl, tg, cap, C.byref(cnt)), "em_profile_read2")
            for i in range(cnt.value):
                f = fam.setdefault(tg[i], [0.0, 0.0, 0])
                f[0] += ms[i]
                f[1] += fl[i]
                f[2] += 1
    lib.em_profile_destroy(prof)
    return fam


def event_bracket_us(dev, dtype):
    """What the HIP-event bracket adds to a launch's measured duration: one encoder-sized projection GEMM
    (M = 7 968, N = K = 256) launched 200 times back to back between two events (its cost inside an unbracketed
    pipeline, boundary included) against the mean of its bracketed durations.  The family figures of `roofline` are
    reported net of it (the bracketed sum exceeded the wall step)."""
    from espnet_amd import lib as L

    lib = L.load()
    act = torch.bfloat16 if dtype == "bfloat16" else torch.float32
    em = L.EM_BF16 if dtype == "bfloat16" else L.EM_F32
    M, N, K = 7968, 256, 256
    A = torch.randn(M, K, device=dev).to(act)
    W = torch.randn(N, K, device=dev).to(act)
    Cm = torch.empty(M, N, dtype=act, device=dev)
    args = L.EmGemmArgs(A=A.data_ptr(), W=W.data_ptr(), C=Cm.data_ptr(), bias=None, M=M, N=N, K=K, lda=K, ldc=N, scale=1.0)

    def launch():
        L.check(lib.em_gemm(em, L.EM_EPI_STORE, L.EM_A_PLAIN, C.byref(args), L.current_stream_ptr()), "em_gemm")

    n = 200
    for _ in range(20):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        launch()
    e1.record()
    torch.cuda.synchronize()
    plain_us = e0.elapsed_time(e1) / n * 1e3
    fam = profile_families(lambda: [launch() for _ in range(n)], 1)
    brk_us = fam[L.EM_PROF_GEMM][0] / fam[L.EM_PROF_GEMM][2] * 1e3
    return max(0.0, brk_us - plain_us)


def rocprof_family_ms(args, model, batch):
    """Per-step kernel time of the MFMA families by `rocprofv3 --kernel-trace --stats` over this script's own main loop
    (a nested run like collect_traffic; VERDICT r04 7b: the HIP-event brackets of back-to-back launches sum to a little
    more than the wall step, the profiler's table is the un-bracketed account).  Steps are counted by the frontend kernel's
    launches.  Returns ({family substring: ms per step}, note)."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get(
            "LD_PRELOAD", ""):
        return None, "already running under a profiler: no nested rocprofv3 passes"
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        cmd = [exe, "--kernel-trace", "--stats", "-d", td, "-o", "s", "--output-format", "csv", "--",
               sys.executable, str(REPO / "bench.py"), "--quick", "--no-roofline", "--no-cpu-baseline", "--no-traffic",
               "--in-flight", "1",  # (one batch in flight: kernels of two streams that overlap in time inflate each other's durations)
               "--steps", "20", "--warmup", "3", "--dtype", args.dtype, "--batch", str(batch), "--model", model]
        env = dict(os.environ, TMPDIR="/tmp", ESPNET_AMD_BENCH_INNER="1")
        if getattr(args, "enc_in_flight", None):  # (the SAME launch sequence as the timed loop: its row-block choice depends on it)
            env["BENCH_ENC_IN_FLIGHT"] = str(args.enc_in_flight)
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=200)
        except subprocess.TimeoutExpired:
            return None, "rocprofv3 --kernel-trace --stats: timeout"
        files = glob.glob(os.path.join(td, "**", "*kernel_stats.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None, f"rocprofv3 --kernel-trace --stats: rc {r.returncode}, {len(files)} csv"
        rows = list(csv.DictReader(open(files[0])))
    steps = sum(int(x["Calls"]) for x in rows if "frontend_logmel" in x["Name"])
    if steps == 0:
        return None, "no frontend launch in the kernel table"
    out = {}
    for sub in set(PROF_MATCH().values()) | {"sub2_kernel"}:
        ns = sum(float(x["TotalDurationNs"]) for x in rows if sub in x["Name"])
        if ns > 0:
            out[sub] = round(ns / steps * 1e-6, 4)
    return out, (f"rocprofv3 --kernel-trace --stats over {steps} steps of the same loop (nested run; under the tracer the loop "
                 f"itself runs 1-3 % slower than the timed one, so this sum can still sit that much above ms_per_step)")


def block_family_algorithmic_bytes(B, T, ff, d=256, L=12, vocab_units=79):
    """Algorithmic HBM bytes of ONE greedy step's block_kernel launches (256-wide fused path): every operand a
    launch must read or write once - the f32 residual rows in and out, q (row-major) and K / V^T (Tpad keys per head) bf16,
    the GLU rows bf16, the position fragments, the launch's weights once - nothing counted twice for the halo or per
    workgroup.  Returns (bytes per step, launches per step)."""
    M = B * T
    Tpad = (T + 255) // 256 * 256
    x = M * d * 4
    qkv = 3 * B * Tpad * d * 2
    glu = M * d * 2
    w_ffn = 2 * d * ff * 2
    w_a = w_ffn + 3 * d * d * 2                      # macaron FFN + q | k | v
    w_c = d * d * 2 + 2 * d * d * 2                  # linear_out + pointwise_conv1
    w_d = d * d * 2 + w_ffn + 31 * d * 4             # pointwise_conv2 + FFN + depthwise taps
    npg = 2 * ((T + 31) // 32) + 4 * ((T + 63) // 64) + 2
    pos = 4 * npg * 2048
    a = 2 * x + qkv + w_a
    attc = qkv + pos + 2 * x + glu + w_c
    da = glu + 2 * x + qkv + w_d + w_a
    fin = glu + x + M * d * 4 + M * d * 2 + M * 4 + w_d + vocab_units * 64 * d * 2
    return a + L * attc + (L - 1) * da + fin, 2 * L + 1


def roofline_object(fam, nprof, bracket_us, peak, wall_s_per_step, traffic, traffic_note, rocprof=None, rocprof_note=None,
                    algorithmic_bytes_per_launch=None):
    names = PROF_NAMES()
    net = {t: max(f[0] - f[2] * bracket_us * 1e-3, 1e-9) for t, f in fam.items()}  # ms, net of the event brackets
    tot_ms = sum(net.values())
    tot_fl = sum(f[1] for f in fam.values())
    launches = sum(f[2] for f in fam.values())
    dom = max(fam, key=lambda t: fam[t][0])
    d_ms, (d_gross, d_fl, d_n) = net[dom], fam[dom]
    achieved = d_fl / (d_ms * 1e-3) / 1e12
    obj = {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": None if traffic is None else round(traffic),
        "traffic_source": traffic_note,
        "algorithmic_bytes_per_launch": None if algorithmic_bytes_per_launch is None else round(algorithmic_bytes_per_launch),
        "traffic_over_algorithmic": (None if traffic is None or not algorithmic_bytes_per_launch
                                     else round(traffic / algorithmic_bytes_per_launch, 3)),
        "kernel": names[dom], "launches_per_step": d_n // nprof,
        "avg_launch_us": round(d_ms * 1e3 / d_n, 2),
        "avg_launch_us_with_event_bracket": round(d_gross * 1e3 / d_n, 2),
        "event_bracket_us": round(bracket_us, 2),
        "timing": "HIP events around every launch on the launch stream, net of the bracket's own cost "
                  "(event_bracket_us, calibrated in the same process on a back-to-back GEMM); the gross figure is "
                  "beside it",
        "algorithmic_gflop_per_launch": round(d_fl / d_n / 1e9, 3),
        "algorithmic_gflop_per_step": round(d_fl / nprof / 1e9, 2),
        "kernel_ms_per_step": round(d_ms / nprof, 3),
        "whole_step": {"algorithmic_gflop_per_step": round(tot_fl / nprof / 1e9, 2),
                       "frac_of_mfma_peak_over_wall_time": round(tot_fl / nprof / wall_s_per_step / 1e12 / peak, 4)},
        "all_mfma_kernels": {"achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                             "frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / peak, 4),
                             "launches_per_step": launches // nprof, "ms_per_step": round(tot_ms / nprof, 3),
                             "algorithmic_gflop_per_step": round(tot_fl / nprof / 1e9, 2)},
        "families": {names[t].split(" ")[0]: {"ms_per_step": round(net[t] / nprof, 3),
                                              "tflops": round(f[1] / (net[t] * 1e-3) / 1e12, 1),
                                              "launches_per_step": f[2] // nprof} for t, f in fam.items()},
    }
    if rocprof:
        # the un-bracketed account of the same families: rocprofv3's kernel table of a nested run of the same
        # loop.  `all_mfma_kernels.ms_per_step` becomes THIS sum; the HIP-event sum (net of the calibrated bracket) stays
        # beside it - brackets around back-to-back launches add up to a little more than the wall step.
        match = PROF_MATCH()
        tot_r = 0.0
        for t, f in fam.items():
            ms = rocprof.get(match[t], 0.0) + (rocprof.get("sub2_kernel", 0.0) if match[t] == "gemm_kernel" else 0.0)
            obj["families"][names[t].split(" ")[0]]["rocprofv3_ms_per_step"] = round(ms, 3)
            tot_r += ms
        a = obj["all_mfma_kernels"]
        a["ms_per_step_hip_events"] = a["ms_per_step"]
        a["ms_per_step"] = round(tot_r, 3)
        a["achieved_rocprofv3"] = round(tot_fl / nprof / (tot_r * 1e-3) / 1e12, 2) if tot_r > 0 else None
        obj["rocprofv3_kernel_ms_per_step"] = round(rocprof.get(match[dom], 0.0), 3)
        obj["rocprofv3_frac"] = round(d_fl / nprof / (rocprof.get(match[dom], 1e9) * 1e-3) / 1e12 / peak, 4)
        obj["rocprofv3_source"] = rocprof_note
    elif rocprof_note:
        obj["rocprofv3_source"] = rocprof_note
    return obj


# what the probes of box_state() read on the boxes of this round that ran the step at its nominal speed (profiles/r06*)
BOX_NOMINAL = {"gemm_4096_tflops": 600.0, "copy_gbs": 2500.0, "block_family_avg_launch_us": 36.0}


def box_state(dev, roofline):
    """Is this box in the pool's slow state?  (HISTORY.md C: the same build at 1.0 and at 1.4 - 1.6 ms per step; the weight-
    streaming kernels run ~1.5x slower there, ALU-bound kernels unchanged.)  Three cheap probes, recorded with the line so that
    a slow lease is labelled instead of being read as a regression: a 4096^3 bf16 GEMM of the library (MFMA + L2), a 1 GiB
    device-to-device copy (HBM), and the fused block kernels' own mean launch time from the roofline leg."""
    from espnet_amd import lib as L

    lib = L.load()
    n = 4096
    A = torch.randn(n, n, device=dev).to(torch.bfloat16)
    W = torch.randn(n, n, device=dev).to(torch.bfloat16)
    Cm = torch.empty(n, n, dtype=torch.bfloat16, device=dev)
    ga = L.EmGemmArgs(A=A.data_ptr(), W=W.data_ptr(), C=Cm.data_ptr(), bias=None, M=n, N=n, K=n, lda=n, ldc=n, scale=1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def gemm():
        L.check(lib.em_gemm(L.EM_BF16, L.EM_EPI_STORE, L.EM_A_PLAIN, C.byref(ga), L.current_stream_ptr()), "em_gemm")

    for _ in range(3):
        gemm()
    e0.record()
    for _ in range(20):
        gemm()
    e1.record()
    torch.cuda.synchronize()
    tfl = 2.0 * n ** 3 * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0.record()
    for _ in range(5):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    gbs = 2.0 * src.numel() * 4 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst, A, W, Cm
    blk = roofline.get("avg_launch_us") if "block_kernel" in roofline.get("kernel", "") else None
    slow = tfl < 0.75 * BOX_NOMINAL["gemm_4096_tflops"] or gbs < 0.75 * BOX_NOMINAL["copy_gbs"] or (
        blk is not None and blk > 1.25 * BOX_NOMINAL["block_family_avg_launch_us"])
    return {"state": "slow" if slow else "nominal", "gemm_4096_tflops": round(tfl, 1), "copy_gbs": round(gbs, 1),
            "block_family_avg_launch_us": blk, "nominal": BOX_NOMINAL,
            "what": "probes of the pool's slow state (a probe 25 % off its nominal value labels the run `slow`: read the "
                    "headline as a property of the lease, not of the code)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000,
                    help="timed steps (default: ~3 s of the greedy workload, long enough for coarse GPU-busy sampling)")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU per step")
    ap.add_argument("--model", default=None, choices=sorted(CONFIGS))
    ap.add_argument("--workload", default="greedy", choices=["greedy", "beam", "stream"],
                    help="greedy = BASELINE.json configs[1] (the bench line); beam = configs[2]: "
                         "Conformer-large, joint CTC/attention beam 10, batch 16; stream = configs[4]")
    ap.add_argument("--beam", type=int, default=10)
    ap.add_argument("--ctc-weight", type=float, default=0.3)
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float32"])
    ap.add_argument("--stream-beam", type=int, default=1,
                    help="--workload stream: beam size; > 1 decodes with the block-synchronous online search "
                         "(BatchBeamSearchOnline) instead of incremental greedy CTC")
    ap.add_argument("--stream-chunk", type=int, default=10240, help="--workload stream: samples per call")
    ap.add_argument("--h2d", action="store_true",
                    help="main loop with the waveforms arriving in pinned host memory (H2D overlapped with "
                         "compute); the default run reports this as the `pcie_inclusive` sub-object instead")
    ap.add_argument("--in-flight", type=int, default=None,
                    help="batches in flight on as many HIP streams: greedy workload default 2 (1 = one stream, steps back to "
                         "back), beam workload default 4 joint searches (a host thread each)")
    ap.add_argument("--no-lane-threads", action="store_true",
                    help="beam workload: ONE host thread drives every search in flight (round 6's first form; saturates at two lanes) "
                         "instead of a host thread per lane (SearchLanes(threaded=True))")
    ap.add_argument("--quick", action="store_true", help="main line + roofline + cpu_baseline only")
    ap.add_argument("--dist-debug-one-gpu", action="store_true",
                    help="developer check of the multi-rank control flow on a ONE-GPU box: every rank uses "
                         "cuda:0 and the collectives go through gloo on host copies (not a measurement)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc passes (roofline.traffic null)")
    args = ap.parse_args()
    inner = os.environ.get("ESPNET_AMD_BENCH_INNER") == "1"
    if args.workload == "stream":
        if args.steps == 2000 and args.warmup == 50:
            args.steps, args.warmup = 20, 2  # a step is a whole 10 s utterance fed chunk by chunk
        return main_stream(args)
    if args.model is None:
        args.model = "small" if args.workload == "greedy" else "large"
    if args.batch is None:
        args.batch = 32 if args.workload == "greedy" else 16
    if args.workload == "beam" and args.steps == 2000 and args.warmup == 50:
        args.steps, args.warmup = 12, 1  # a beam step is ~100x a greedy one
    if args.in_flight is None:
        args.in_flight = 4 if args.workload == "beam" else 2
    args.lane_threads = not args.no_lane_threads

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (the contract's own command line,
        # one rank per GPU over RCCL) instead of failing on the world-size check below
        import socket

        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(REPO / "bench.py"), *sys.argv[1:]]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if args.dist_debug_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_debug_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from espnet_amd import lib as L
    from espnet_amd.tasks.asr import ASRTask

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from espnet_amd import distributed as D

    def sink_factory(B, width):
        # device ring of RING_STEPS steps' records, written in place by the CTC collapse kernel; one all-gather (N > 1)
        # and one device->host copy per RING_STEPS steps, delivered while the next steps run
        return D.RecordRing(rank, world, B, width, RING_STEPS, dev, host_gloo=args.dist_debug_one_gpu)

    B = args.batch
    extras = {}
    if args.workload == "beam":
        r = run_beam(args, dev, B, args.beam, args.steps, args.warmup,
                     cpu_base=(rank == 0 and world == 1 and not args.no_cpu_baseline))
        model, elapsed, n_tok = r.pop("model"), r.pop("elapsed"), r.pop("tokens")
        extras = r
        step_plain = None
    else:
        torch.manual_seed(0)
        model = ASRTask.build_model(model_config(args.model, args.dtype)).to(dev).eval()
        wav_host = synth_batch(rank * B, B)
        wav = wav_host.to(dev)
        lens = [N_SAMPLES] * B
        T = model.encoder.output_frames(1 + N_SAMPLES // 160)
        sink = sink_factory(B, T)
        pipe = StepPipeline(dev, args.in_flight) if args.in_flight > 1 else None
        # (the library takes the 512-wide models' row-block launches from a smaller share of the chip when it is told that
        # other batches are in flight: EM_ENC_IN_FLIGHT)
        model.encoder.batches_in_flight = int(os.environ.get("BENCH_ENC_IN_FLIGHT") or (pipe.depth if pipe is not None else 1))
        feeder = (HostFeeder(wav_host, dev, nbuf=(pipe.depth if pipe is not None else 1) + 1,
                             beside=pipe.streams if pipe is not None else None) if args.h2d else None)

        def step_plain(src=None, out=None):
            st = model.encode_device(wav if src is None else src, lens)
            return model.greedy_ctc_device(st, out=out)

        def body(tok_v, len_v):
            if feeder is not None:
                step_plain(feeder.acquire(), out=(tok_v, len_v))
                feeder.release()
            else:
                step_plain(out=(tok_v, len_v))

        def step():
            tok_v, len_v, _ = sink.slot()
            if pipe is not None:
                # (a macro-batch of the record ring = RING_STEPS steps dealt round-robin to the streams; the ring half is
                # flushed - all-gather + D2H on the main stream - behind all of them, and rewritten only behind its flush)
                half = (sink.k // RING_STEPS) & 1
                if sink.k % RING_STEPS == 0:
                    pipe.fork(half)
                pipe.run(lambda: body(tok_v, len_v))
                if (sink.k + 1) % RING_STEPS == 0:
                    pipe.join()
                    sink.commit()
                    pipe.flushed(half)
                    return
            else:
                body(tok_v, len_v)
            sink.commit()

        def finish():
            if pipe is not None:
                pipe.join()
            sink.drain()
            if pipe is not None:
                pipe.flush_ev.clear()  # (drain delivered everything to the host: the next macro-batch forks from this stream)

        elapsed = timed_loop(step, args.steps, args.warmup, barrier, finish)
        n_tok = int(sink.last[1].sum()) if sink.last is not None else 0
        assert rank != 0 or sink.delivered == args.steps + args.warmup, (sink.delivered, args.steps, args.warmup)
    el = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.dist_debug_one_gpu else dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    out = None
    cfgm = CONFIGS[args.model]
    if rank == 0:
        value = world * B * AUDIO_SEC * args.steps / elapsed
        if args.workload == "greedy":
            wl = ((f"BASELINE.json configs[1]: Conformer-{args.model} (12x{cfgm['d']}d, {cfgm['heads']} heads), "
                   if "ebf" not in cfgm else
                   "SURVEY 8(f) rank 4: E-Branchformer (17x512d, 8 heads, cgMLP 3072, merge k31), ") +
                  f"HIP STFT/log-mel + HIP encoder + greedy CTC (G1) + hypotheses collated to the host, "
                  f"{B} x 10 s utterances per GPU per step, V={VOCAB}")
        else:
            wl = (f"BASELINE.json configs[2]: Conformer-{args.model} (12x{cfgm['d']}d, {cfgm['heads']} heads) + 6-layer "
                  f"attention decoder, joint CTC/attention beam search beam={args.beam} ctc_weight={args.ctc_weight}, "
                  f"{B} x 10 s utterances per GPU per step, V={VOCAB}")
        out = {
            "metric": "audio-seconds/sec (RTF^-1), Conformer-ASR, 10 s utterances",
            "value": round(value, 1), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bfloat16" else "f32", "data": "synthetic",
            "config": {"workload": wl, "batch_per_gpu": B, "global_batch": world * B,
                       "audio_seconds_per_utt": AUDIO_SEC, "parallelism": f"utterance-dp{world}",
                       "collation": ("espnet_amd.distributed.RecordRing: records written in place into a device ring, one "
                                     f"all-gather (N > 1) + one async D2H per {RING_STEPS} steps, all delivered inside "
                                     "the timed region" if args.workload == "greedy" else
                                     "espnet_amd.distributed.decode_dynamic: batches pulled from a shared counter, "
                                     "indexed records collated once at the end of the timed region"),
                       "inputs": ("pinned host -> device copy inside the timed step (double buffered)" if args.h2d
                                  else "resident in HBM when the timed region starts (the bench contract of this "
                                       "repository; SURVEY 8(d) counts the H2D copy of the waveforms: that rate is "
                                       "the `pcie_inclusive` sub-object of this line)"),
                       "tokens_last_step": n_tok},
        }
        if args.workload == "greedy":
            out["config"]["batches_in_flight"] = (
                {"n": pipe.depth, **pipe.probe,
                 "what": "steps dealt round-robin to this many HIP streams: every kernel takes the whole chip, so steps do not run "
                         "side by side, but one stream's launch fills the boundary between two dependent launches of the "
                         "other (~27 per step); ms_per_step = timed region / steps, i.e. per batch; `one_stream` below is the "
                         "same loop with --in-flight 1"} if pipe is not None else {"n": 1})
        out.update(extras)
    # ---- roofline of the MFMA kernel families: HIP events around every launch (on the launch stream)
    if rank == 0 and not args.no_roofline and step_plain is not None:
        nprof = max(1, min(args.steps, 5))
        fam = profile_families(step_plain, nprof)
        bracket_us = event_bracket_us(dev, args.dtype)
        dom = max(fam, key=lambda t: fam[t][0])
        traffic, traffic_note = (None, "skipped")
        if world == 1 and not inner and not args.no_traffic and not args.quick:
            try:
                traffic, traffic_note = collect_traffic(PROF_MATCH()[dom], args)
            except Exception as e:  # measurement helper: never lose the bench line to it
                traffic, traffic_note = None, f"{type(e).__name__}: {e}"
        rp, rp_note = None, None
        if world == 1 and not inner and not args.no_traffic and not args.quick:
            try:
                rp, rp_note = rocprof_family_ms(args, args.model, args.batch)
            except Exception as e:  # noqa: BLE001
                rp, rp_note = None, f"{type(e).__name__}: {e}"
        alg = None
        if args.model == "small" and PROF_MATCH()[dom] == "block_kernel":
            ab, nl = block_family_algorithmic_bytes(B, T, CONFIGS["small"]["ff"])
            alg = ab / nl
        out["roofline"] = roofline_object(fam, nprof, bracket_us, MFMA_PEAK_TFLOPS[args.dtype], elapsed / args.steps,
                                          traffic, traffic_note, rp, rp_note, algorithmic_bytes_per_launch=alg)
        out["box_state"] = box_state(dev, out["roofline"])
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "greedy":
        out["cpu_baseline"] = cpu_baseline(model)

    # ---- the other rows of the measurement contract, as sub-objects of the one line (N = 1 only) -------------
    if rank == 0 and world == 1 and args.workload == "greedy" and not args.quick and not inner:
        only = [x for x in os.environ.get("BENCH_ONLY_LEGS", "").split(",") if x]  # developer probe: run these sub-legs only

        def guarded(name, fn):
            if only and name not in only:
                return
            try:
                out[name] = fn()
            except Exception as e:  # a sub-measurement must never cost the headline line
                out[name] = {"error": f"{type(e).__name__}: {e}"}

        def frontend_leg():
            flens = model.frontend.feature_lengths(lens)
            flens_dev = torch.tensor(flens, dtype=torch.int32, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            with torch.no_grad():
                for _ in range(5):
                    model.frontend.forward_device(wav, flens_dev, None)
                e0.record()
                for _ in range(n):
                    model.frontend.forward_device(wav, flens_dev, None)
                e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / n * 1e-3
            gbs = B * FRONTEND_BYTES_PER_UTT / t / 1e9
            return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "us_per_batch": round(t * 1e6, 1),
                    "algorithmic_bytes_per_utt": FRONTEND_BYTES_PER_UTT,
                    "kernel": "frontend_logmel (STFT + power + log-mel, csrc/frontend.hip), HIP events on the launch "
                              "stream over 50 launches"}

        def pipelined_loop(sk, pp, body, k, warm):
            """`k` timed steps of `body(tok_v, len_v)` collated through ring `sk`, dealt to the streams of `pp` (None: this stream)."""
            def st():
                tok_v, len_v, _ = sk.slot()
                if pp is None:
                    body(tok_v, len_v)
                else:
                    half = (sk.k // RING_STEPS) & 1
                    if sk.k % RING_STEPS == 0:
                        pp.fork(half)
                    pp.run(lambda: body(tok_v, len_v))
                    if (sk.k + 1) % RING_STEPS == 0:
                        pp.join()
                        sk.commit()
                        pp.flushed(half)
                        return
                sk.commit()

            def fin():
                if pp is not None:
                    pp.join()
                sk.drain()
                if pp is not None:
                    pp.flush_ev.clear()

            return timed_loop(st, k, warm, barrier, fin)

        def one_stream_leg():
            k = min(args.steps, 600)
            t = pipelined_loop(sink_factory(B, T), None, lambda a, b: step_plain(out=(a, b)), k, 20)
            return {"value": round(B * AUDIO_SEC * k / t, 1), "unit": "audio-s/s", "ms_per_step": round(t / k * 1e3, 3), "steps": k,
                    "what": "the loop of `value` with ONE batch in flight (steps back to back on one stream: the form of rounds 1-5)"}

        def pcie_leg():
            depth = pipe.depth if pipe is not None else 1
            pp = StepPipeline(dev, depth) if depth > 1 else None
            # (the copy stream must run beside EVERY stream the steps run on: probed against them, not against this stream)
            fd = HostFeeder(wav_host, dev, nbuf=depth + 1, beside=pp.streams if pp is not None else None)

            def body(tok_v, len_v):
                step_plain(fd.acquire(), out=(tok_v, len_v))
                fd.release()

            k = min(args.steps, 300)
            t = pipelined_loop(sink_factory(B, T), pp, body, k, 10)
            return {"value": round(B * AUDIO_SEC * k / t, 1), "unit": "audio-s/s", "ms_per_step": round(t / k * 1e3, 3),
                    "vs_resident": round((B * AUDIO_SEC * k / t) / value, 4),
                    "copy_stream_probe": fd.stream_probe,
                    "steps": k, "what": "the same step with the waveforms arriving in pinned host memory: H2D of "
                                        f"{B * N_SAMPLES * 4 / 1e6:.1f} MB per step on a copy stream into one of (batches in flight + 1) buffers, "
                                        "overlapped with the previous steps' compute, batches in flight as in `value`; D2H of the "
                                        "hypotheses as in `value`"}

        ids_bf16 = {}

        def parity_leg():
            with torch.no_grad():
                ids_b, tok_b, tl_b = (t.cpu() for t in step_plain())
                model.set_compute_dtype("float32")
                st_f = model.encode_device(wav, lens)
                ids_f, tok_f, tl_f = (t.cpu() for t in model.greedy_ctc_device(st_f))
                # top-2 log-prob margin of the f32 result per frame (checker arithmetic in torch, not the product path)
                lo = model.ctc.ctc_lo
                lp = torch.log_softmax(st_f.enc_out.float() @ lo.weight.float().t() + lo.bias.float(), dim=-1)
                top2 = lp.topk(2, dim=-1).values
                margin = (top2[..., 0] - top2[..., 1]).cpu()
                del lp, top2
            flipped = ids_b != ids_f
            mism = float(flipped.float().mean())
            worst = float(margin[flipped].max()) if bool(flipped.any()) else 0.0
            dist_, nref = 0, 0
            for b in range(B):
                ref = tok_f[b, : int(tl_f[b])].tolist()
                dist_ += _edit_distance(tok_b[b, : int(tl_b[b])].tolist(), ref)
                nref += len(ref)
            ids_bf16["done"] = True
            return {"frame_id_mismatch_rate": round(mism, 5), "token_edit_distance": dist_, "reference_tokens": nref,
                    "token_error_rate": round(dist_ / max(1, nref), 5),
                    "max_reference_margin_at_mismatch": round(worst, 6),
                    "frames_with_margin_below_that": round(float((margin <= worst).float().mean()), 5),
                    "what": "timed mode (bf16 MFMA, fused blocks) against the f32 parity mode of the same weights on "
                            "the bench batch: per-frame CTC arg-max ids and G1 tokens; max_reference_margin_at_mismatch = "
                            "the largest f32 top-2 log-prob margin among the frames whose id flipped (random-init "
                            "posteriors are nearly flat: only such near-ties may flip; tests/test_gpu_fullsize.py and "
                            "tests/test_gpu_e2e.py assert it against the oracle / the reference fixtures, and require exact "
                            "tokens on the peaked-posterior fixture)"}

        def f32_leg():
            if not ids_bf16:
                model.set_compute_dtype("float32")
            sk = sink_factory(B, T)

            def st():
                tok_v, len_v, _ = sk.slot()
                step_plain(out=(tok_v, len_v))
                sk.commit()

            k = min(args.steps, 30)
            t = timed_loop(st, k, 3, barrier, sk.drain)
            model.set_compute_dtype(args.dtype)
            return {"value": round(B * AUDIO_SEC * k / t, 1), "unit": "audio-s/s", "ms_per_step": round(t / k * 1e3, 3),
                    "steps": k, "dtype": "f32", "what": "the exact-f32 parity mode (the mode the element-wise oracle "
                                                        "tests run in) on the same batch"}

        def encoder_leg(name, Bl, survey_gflop_per_utt, what, want_pmc=False):
            """An encoder alone (greedy CTC so that the step is the encoder) with the roofline of its MFMA kernel families:
            Conformer-large (the model of configs[2] / [3]) at B = 64, the per-GPU batch of configs[3] - SURVEY 8(d): 68.56
            algorithmic GFLOP per utterance; E-Branchformer at B = 32 - flops from the launches' own
            accounting (no survey figure)."""
            def fn():
                torch.manual_seed(0)
                m = ASRTask.build_model(model_config(name, args.dtype)).to(dev).eval()
                w = synth_batch(0, Bl).to(dev)
                ls = [N_SAMPLES] * Bl
                sk = sink_factory(Bl, T)

                def sp(out=None):
                    return m.greedy_ctc_device(m.encode_device(w, ls), out=out)

                k = 32
                # (batches in flight as in `value`; the E-Branchformer leg at B = 32 - 125 row-block workgroups per launch, half the
                # chip - with three: 4.62 against 4.89 ms per batch with two, profiles/r06ae_rows_fill_ab.txt)
                pp = StepPipeline(dev, max(pipe.depth, 3) if name == "ebf" else pipe.depth) if pipe is not None else None
                m.encoder.batches_in_flight = pp.depth if pp is not None else 1
                t = pipelined_loop(sk, pp, lambda a_, b_: sp(out=(a_, b_)), k, 4)
                fam = profile_families(sp, 3)
                brk = event_bracket_us(dev, args.dtype)
                peak = MFMA_PEAK_TFLOPS[args.dtype]
                del m
                torch.cuda.empty_cache()
                traffic, note = None, "not collected for this leg"
                if want_pmc and world == 1 and not inner and not args.no_traffic:
                    # HBM bytes per launch of the leg's dominant family (the row-block launches had none)
                    dom = max(fam, key=lambda tg: fam[tg][0])
                    a2 = argparse.Namespace(**vars(args))
                    a2.model, a2.batch = name, Bl
                    try:
                        traffic, note = collect_traffic(PROF_MATCH()[dom], a2)
                    except Exception as e:  # noqa: BLE001 - a measurement helper must not cost the line
                        traffic, note = None, f"{type(e).__name__}: {e}"
                rp, rp_note = None, None
                if want_pmc and world == 1 and not inner and not args.no_traffic:
                    try:
                        a4 = argparse.Namespace(**vars(args))
                        a4.enc_in_flight = pp.depth if pp is not None else 1
                        rp, rp_note = rocprof_family_ms(a4, name, Bl)
                    except Exception as e:  # noqa: BLE001
                        rp, rp_note = None, f"{type(e).__name__}: {e}"
                r = roofline_object(fam, 3, brk, peak, t / k, traffic, note, rp, rp_note)
                res = {"value": round(Bl * AUDIO_SEC * k / t, 1), "unit": "audio-s/s", "ms_per_step": round(t / k * 1e3, 3),
                       "steps": k, "warmup": 4, "dtype": "bf16" if args.dtype == "bfloat16" else "f32",
                       "config": {"workload": f"{what} encoder + greedy CTC, {Bl} x 10 s utterances per step, V={VOCAB}",
                                  "batches_in_flight": pp.depth if pp is not None else 1}}
                if survey_gflop_per_utt:
                    g = survey_gflop_per_utt * Bl
                    res["whole_step_frac_of_mfma_peak"] = round(g * 1e9 / (t / k) / 1e12 / peak, 4)
                    res["algorithmic_gflop_per_step_survey"] = round(g, 1)
                else:  # (the MFMA launches' own flop accounting over the wall time of the step)
                    res["whole_step_frac_of_mfma_peak"] = r["whole_step"]["frac_of_mfma_peak_over_wall_time"]
                res["roofline"] = r
                return res
            return fn

        def beam_leg(Bb, steps, cpu):
            def fn():
                a2 = argparse.Namespace(**vars(args))
                a2.model = "large"
                # joint searches in flight, a host thread per lane: four at configs[3]'s per-GPU batch (6 / 8: no more), eight at
                # configs[2]'s 160 rows, whose launches leave most of the chip idle (4 / 6 / 8 lanes: 4 033 / 4 083 / 4 705
                # audio-s/s, profiles/r06al_beam_lanes_sweep.txt)
                a2.in_flight, a2.lane_threads = (8 if Bb <= 16 else 4), True
                r = run_beam(a2, dev, Bb, 10, steps, 1, cpu_base=cpu and not args.no_cpu_baseline,
                             want_traffic=cpu,  # counters for the configs[2] leg only
                             want_oracle=not args.no_cpu_baseline)  # ... the oracle check of utterance 0 for both
                r.pop("model")
                el_, tok_ = r.pop("elapsed"), r.pop("tokens")
                torch.cuda.empty_cache()
                if cpu and not args.no_cpu_baseline:
                    # the exact-f32 parity mode of the same search on the same batch: the mode in which every oracle
                    # hypothesis is in the device n-best
                    try:
                        a3 = argparse.Namespace(**vars(a2))
                        a3.dtype, a3.in_flight = "float32", 1
                        r3 = run_beam(a3, dev, Bb, 10, 1, 1, cpu_base=False, want_traffic=False, want_oracle=True)
                        r3.pop("model")
                        e3 = r3.pop("elapsed")
                        r["f32_mode"] = {"value": round(Bb * AUDIO_SEC / e3, 1), "unit": "audio-s/s", "ms_per_step": round(e3 * 1e3, 2),
                                         "ms_per_search_step": r3["search"]["ms_per_search_step"],
                                         "f32_vs_oracle": r3.get("f32_vs_oracle")}
                        torch.cuda.empty_cache()
                    except Exception as e:  # noqa: BLE001
                        r["f32_mode"] = {"error": f"{type(e).__name__}: {e}"}
                return {"value": round(Bb * AUDIO_SEC * steps / el_, 1), "unit": "audio-s/s",
                        "ms_per_step": round(el_ / steps * 1e3, 2), "steps": steps, "warmup": 1,
                        "dtype": "bf16" if args.dtype == "bfloat16" else "f32",
                        "config": {"workload": f"Conformer-large (12x512d, 8 heads) + 6-layer attention decoder, joint "
                                               f"CTC/attention beam search beam=10 ctc_weight={args.ctc_weight}, "
                                               f"{Bb} x 10 s utterances per step, V={VOCAB}", "tokens_last_step": tok_},
                        **r}
            return fn

        def stream_leg():
            r = run_stream(args.dtype, 3, 1, 10240, 1, cpu_base=not args.no_cpu_baseline)
            s = run_stream(args.dtype, 2, 1, 640, 1, cpu_base=False)
            r["stress_40ms_calls"] = {"value": s["value"], "ms_per_step": s["ms_per_step"], **{
                k: s["config"][k] for k in ("chunk_ms", "calls_per_utt", "call_latency_ms_median",
                                            "call_latency_ms_p95", "realtime_factor_of_one_stream")}}
            r["batch32"] = run_stream_batch(args.dtype, 32, 4, 1)
            # a tick's row-block launches are two 32-row workgroups per stream - 64 of 256 CUs at 32 streams - and nearly flat
            # until the chip is full: the same tick with four times the streams.  (Four steps, not two:
            # one stalled tick in a two-step run read 34 800 where three runs around it read 52 800 - 53 200, profiles/r05x, r05y)
            r["batch128"] = run_stream_batch(args.dtype, 128, 4, 1)
            # ... and with a tick of a SECOND group of streams in flight on another HIP stream (batch_call_async): the host's
            # turn-around between a tick's read and the next tick's first launch, and the ~25 small launches either side of the
            # layers, under the other group's launches
            r["batch32_hipgraph"] = run_stream_batch(args.dtype, 32, 4, 1, graph=True)  # (the tick as one captured unit: no faster)
            r["batch32_two_groups"] = run_stream_batch(args.dtype, 32, 4, 1, groups=2)
            r["batch128_two_groups"] = run_stream_batch(args.dtype, 128, 4, 1, groups=2)
            r["batch64_three_groups"] = run_stream_batch(args.dtype, 64, 4, 1, groups=3)
            return r

        # Order (round 6): the two beam legs first.  Inside this one process the late legs measured slower than the same
        # leg alone (`BENCH_ONLY_LEGS=beam_cfg3_per_gpu`: 0.515 ms per label step and 7 250 - 7 550 audio-s/s; at the end of the
        # full line 0.58 and 5 600 - 6 400, three runs) while nothing but the process' history differed (H2D feeders, pinned
        # buffers, the encoder legs' models and PMC sub-processes in front of them).  The cause is not identified: the number
        # of streams the process has created is NOT it (tools/null_stream_probe.py: 4.6 against 4.7 us per launch after all of
        # torch's pool exists, profiles/r06ak_null_stream_probe.txt).  A label step is ~40 launches of ~10 us and feels every
        # microsecond of launch cost; the other legs measure the same in either order (profiles/r06ai vs r06ak).
        # `python bench.py --workload beam --batch 16 | 64` is each leg in a process of its own.
        guarded("beam", beam_leg(16, 24, True))
        guarded("beam_cfg3_per_gpu", beam_leg(64, 12, False))
        guarded("frontend", frontend_leg)
        guarded("pcie_inclusive", pcie_leg)
        if pipe is not None:
            guarded("one_stream", one_stream_leg)
        guarded("bf16_vs_f32", parity_leg)
        guarded("f32_mode", f32_leg)
        del model
        torch.cuda.empty_cache()
        guarded("encoder_large_b64", encoder_leg("large", 64, 68.56, "Conformer-large (12x512d, 8 heads, ff 2048)", want_pmc=True))
        guarded("encoder_ebranchformer_b32", encoder_leg("ebf", 32, None, "E-Branchformer (17x512d, 8 heads, cgMLP 3072, merge k31)", want_pmc=True))
        guarded("stream", stream_leg)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
