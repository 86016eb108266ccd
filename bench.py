#!/usr/bin/env python3
"""bench.py — audio-seconds/sec (RTF^-1) of the ASR-inference hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic utterances already resident in
HBM: HIP STFT/log-mel frontend -> HIP Conformer encoder -> greedy CTC (G1) decode, all ranks'
hypotheses collated with one RCCL all-gather per step (N > 1).  Workload = BASELINE.json
configs[1]: Conformer-small (12 x 256d, 4 heads, ff 1024), batch 32 x 10 s @ 16 kHz per GPU
(weak scaling: per-GPU batch fixed).  Random-init weights (torch.manual_seed(0)), synthetic
N(0, 0.1^2) waveforms.

One JSON line on rank 0 carries the throughput, the roofline of the dominant kernel family (the
MFMA GEMM template, measured live with HIP events around every GEMM launch of extra steps run
right after the timed region), and the CPU baseline (the oracle port on this box's host cores).
"""
import argparse
import ctypes as C
import json
import os
import sys
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
            "cpu_model": _cpu_model(),
            "sample": f"oracle CPU-fp32 port, {len(times)} utterances of 10 s, batch 1, median "
                      f"{med*1e3:.1f} ms/utt (frontend + encoder + greedy CTC G1), 1 warm-up"}


def cpu_baseline_beam(model, beam, ctc_weight, budget_s=25.0):
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
    t_start = time.perf_counter()
    i = 0
    with torch.no_grad():
        while True:
            wav = synth_batch(9000 + i, 1)
            t0 = time.perf_counter()
            e, ol = oc.encode(sd, wav, torch.tensor([N_SAMPLES]), enc.heads, enc.num_blocks, 512,
                              fe.win_length, 160)
            ob.beam_search(sd, e[0, : int(ol[0])], dec.heads, dec.num_blocks, beam, ctc_weight,
                           sos=VOCAB - 1, eos=VOCAB - 1)
            times.append(time.perf_counter() - t0)
            i += 1
            if time.perf_counter() - t_start > budget_s or len(times) >= 20:
                break
    med = sorted(times)[len(times) // 2]
    return {"value": round(AUDIO_SEC / med, 3), "unit": "audio-s/s", "cores": cores, "kind": "port",
            "cpu_model": _cpu_model(),
            "sample": f"oracle CPU-fp32 port (K/V-cached restatement of Speech2Text beam search), "
                      f"{len(times)} utterances of 10 s, batch 1, median {med:.2f} s/utt "
                      f"(frontend + encoder + beam {beam} search, 249 steps), no warm-up"}


def main_stream(args):
    """BASELINE.json configs[4]: streaming contextual-block Conformer (aishell recipe shape: 12 x
    256d, 4 heads, ff 2048, conv k 15, block 40 / hop 16 / look-ahead 16), one audio stream fed in
    640 ms chunks through Speech2TextStreaming (HIP frontend -> hipGraph-captured encoder step ->
    incremental greedy CTC).  A step = one 10 s utterance = 16 chunks."""
    import tempfile

    import yaml

    from espnet_amd.bin.asr_inference_streaming import Speech2TextStreaming

    assert int(os.environ.get("WORLD_SIZE", "1")) == 1, "--workload stream is single-stream"
    torch.cuda.set_device(0)
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
        s2t = Speech2TextStreaming(str(Path(td) / "config.yaml"), None, device="cuda", dtype=args.dtype,
                                   beam_size=args.stream_beam, ctc_weight=0.3)
    chunk = 10240
    wav = synth_batch(0, 1)[0]
    chunks = [wav[p : p + chunk] for p in range(0, N_SAMPLES, chunk)]

    def step():
        lat = []
        for k, c in enumerate(chunks):
            t0 = time.perf_counter()
            out = s2t(c, is_final=(k == len(chunks) - 1))  # ends with a host read of the new tokens
            lat.append(time.perf_counter() - t0)
        return out, lat

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lats = []
    for _ in range(args.steps):
        out, lat = step()
        lats += lat[2:-1]  # steady-state chunks (the first two only buffer, the last is final)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lats.sort()
    res = {"metric": "audio-seconds/sec (RTF^-1), Conformer-ASR, 10 s utterances", "value":
           round(AUDIO_SEC * args.steps / elapsed, 1), "unit": "audio-s/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16" if args.dtype == "bfloat16" else "f32", "data": "synthetic",
           "config": {"workload": "BASELINE.json configs[4]: streaming contextual_block_conformer (12x256d, "
                                  "block 40 / hop 16 / look-ahead 16), ONE stream, 640 ms chunks, hipGraph-"
                                  "captured encoder step, " + ("incremental greedy CTC" if args.stream_beam <= 1 else
                                                               "block-synchronous online beam search"),
                      "chunk_ms": 640, "chunks_per_utt": len(chunks),
                      "chunk_latency_ms_median": round(lats[len(lats) // 2] * 1e3, 3),
                      "chunk_latency_ms_p95": round(lats[int(len(lats) * 0.95)] * 1e3, 3),
                      "hipgraph_replays": s2t._runner.n_replays if s2t._runner else 0,
                      **({"search": f"BatchBeamSearchOnline beam {args.stream_beam}, ctc_weight 0.3",
                          "search_steps_per_utt": s2t.beam_search.n_steps // (args.steps + args.warmup)}
                         if args.stream_beam > 1 else {}),
                      "tokens_last_utt": len(out[0][2]) if out else 0}}
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_stream(s2t.asr_model, enc_conf)
    print(json.dumps(res), flush=True)


def cpu_baseline_stream(model, enc_conf, budget_s=15.0):
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
            "cpu_model": _cpu_model(),
            "sample": f"oracle CPU-fp32 port of the streaming encoder (frontend + 64-frame chunks through "
                      f"forward_infer, no decoding), {len(times)} utterances of 10 s, median {med:.2f} s/utt"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU per step")
    ap.add_argument("--model", default=None, choices=sorted(CONFIGS))
    ap.add_argument("--workload", default="greedy", choices=["greedy", "beam", "stream"],
                    help="greedy = BASELINE.json configs[1] (the bench line); beam = configs[2]: "
                         "Conformer-large, joint CTC/attention beam 10, batch 16")
    ap.add_argument("--beam", type=int, default=10)
    ap.add_argument("--ctc-weight", type=float, default=0.3)
    ap.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float32"])
    ap.add_argument("--streams", type=int, default=1,
                    help="split the per-GPU batch over this many HIP streams (independent utterances)")
    ap.add_argument("--stream-beam", type=int, default=1,
                    help="--workload stream: beam size; > 1 decodes with the block-synchronous online search "
                         "(BatchBeamSearchOnline) instead of incremental greedy CTC")
    ap.add_argument("--h2d", action="store_true",
                    help="also copy the waveforms host->device inside every timed step (pinned host memory): the "
                         "PCIe-inclusive rate quoted in DESIGN.md; never the headline `value`")
    ap.add_argument("--graph", action="store_true",
                    help="greedy workload: capture the whole pass (all --streams branches) in one hipGraph")
    ap.add_argument("--dist-debug-one-gpu", action="store_true",
                    help="developer check of the multi-rank control flow on a ONE-GPU box: every rank uses "
                         "cuda:0 and the collectives go through gloo on host copies (not a measurement)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()
    if args.workload == "stream":
        return main_stream(args)
    if args.model is None:
        args.model = "small" if args.workload == "greedy" else "large"
    if args.batch is None:
        args.batch = 32 if args.workload == "greedy" else 16

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

    torch.manual_seed(0)
    model = ASRTask.build_model(model_config(args.model, args.dtype)).to(dev).eval()
    B = args.batch
    wav_host = synth_batch(rank * B, B).pin_memory() if args.h2d else None
    wav = synth_batch(rank * B, B).to(dev)
    lens = [N_SAMPLES] * B
    T = model.encoder.output_frames(1 + N_SAMPLES // 160)
    # hypotheses are collated with ONE fixed-shape all-gather per step: (B, T + 1) int32 records,
    # token ids padded with -1 and the token count in the last column
    gathered = torch.empty(world * B, T + 1, dtype=torch.int32, device=dev) if world > 1 else None

    def collate(tokens, tlens):
        rec = torch.cat([tokens, tlens.view(-1, 1)], dim=1).contiguous()
        if args.dist_debug_one_gpu:
            parts = [torch.empty(rec.shape, dtype=rec.dtype) for _ in range(world)]
            dist.all_gather(parts, rec.cpu())
            gathered.copy_(torch.cat(parts, 0))
            return
        dist.all_gather_into_tensor(gathered, rec)

    beam_search = None
    if args.workload == "beam":
        from espnet_amd.nets.batch_beam_search import build_beam_search

        beam_search = build_beam_search(model, beam_size=args.beam, ctc_weight=args.ctc_weight,
                                        penalty=0.0, token_list=model.token_list)

    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else []

    def step_multistream():
        # utterances are independent: each slice of the batch runs the whole path on its own
        # stream, so the small latency-bound kernels of different slices overlap on the chip
        main = torch.cuda.current_stream()
        outs = []
        per = (B + len(streams) - 1) // len(streams)
        for k, s in enumerate(streams):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                sl = slice(k * per, min(B, (k + 1) * per))
                st = model.encode_device(wav[sl], lens[sl])
                outs.append(model.greedy_ctc_device(st)[1:])
        for s in streams:
            main.wait_stream(s)
        return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])

    graph = {}

    def step(do_collate=True):
        # do_collate=False: rank 0's event-instrumented passes after the timed region run alone, so
        # they must not enter a collective the other ranks never join
        if graph:
            graph["g"].replay()
            if world > 1 and do_collate:
                collate(*graph["out"])
            return graph["out"]
        if streams and beam_search is None:
            tokens, tlens = step_multistream()
            if world > 1 and do_collate:
                collate(tokens, tlens)
            return tokens, tlens
        if wav_host is not None:
            wav.copy_(wav_host, non_blocking=True)
        st = model.encode_device(wav, lens)
        if beam_search is None:
            _, tokens, tlens = model.greedy_ctc_device(st)
        else:  # n-best lists are rebuilt on the host; the best one is padded back for collation
            nbest = beam_search.search_batch(st.enc_act, st.olens)
            tokens = torch.full((B, T), -1, dtype=torch.int32)
            tl = []
            for b, hyps in enumerate(nbest):
                ids = [t for t in hyps[0].yseq[1:-1].tolist() if t != 0][:T] if hyps else []
                tokens[b, : len(ids)] = torch.tensor(ids, dtype=torch.int32)
                tl.append(len(ids))
            tokens = tokens.to(dev)
            tlens = torch.tensor(tl, dtype=torch.int32, device=dev)
        if world > 1 and do_collate:  # collate hypotheses: one RCCL all-gather of fixed-shape ids + lengths
            collate(tokens, tlens)
        return tokens, tlens

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        if args.graph and beam_search is None:
            # the pass has no host-dependent control flow (lengths live on the device), so the whole
            # frontend -> encoder -> greedy CTC chain, with its fork/join over --streams, is one graph
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                if streams:
                    out_g = step_multistream()
                else:
                    out_g = model.greedy_ctc_device(model.encode_device(wav, lens))[1:]
            graph.update(g=g, out=out_g)
            for _ in range(2):
                step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tokens, tlens = step()
        barrier()
        elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.dist_debug_one_gpu else dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    n_tok = int(tlens.sum().item())

    out = None
    if rank == 0:
        value = world * B * AUDIO_SEC * args.steps / elapsed
        out = {
            "metric": "audio-seconds/sec (RTF^-1), Conformer-ASR, 10 s utterances",
            "value": round(value, 1), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bfloat16" else "f32", "data": "synthetic",
            "config": {"workload": ((f"BASELINE.json configs[1]: Conformer-{args.model} "
                                     f"(12x{CONFIGS[args.model]['d']}d, {CONFIGS[args.model]['heads']} heads), "
                                     if "ebf" not in CONFIGS[args.model] else
                                     "SURVEY 8(f) rank 4: E-Branchformer (17x512d, 8 heads, cgMLP 3072, merge k31), ") +
                                    f"HIP STFT/log-mel + HIP encoder + greedy CTC (G1), "
                                    f"{B} x 10 s utterances per GPU per step, V={VOCAB}")
                       if beam_search is None else
                       (f"BASELINE.json configs[2]: Conformer-{args.model} "
                        f"(12x{CONFIGS[args.model]['d']}d, {CONFIGS[args.model]['heads']} heads) + 6-layer "
                        f"attention decoder, joint CTC/attention beam search beam={args.beam} "
                        f"ctc_weight={args.ctc_weight}, {B} x 10 s utterances per GPU per step, V={VOCAB}"),
                       "batch_per_gpu": B, "global_batch": world * B, "audio_seconds_per_utt": AUDIO_SEC,
                       "parallelism": f"utterance-dp{world}", "greedy_tokens_last_step_rank0": n_tok,
                       **({"inputs": "host (pinned) -> device copy inside the timed step"} if args.h2d else {})},
        }
    # ---- roofline of the MFMA kernel families: HIP events around every launch (on the launch stream)
    if rank == 0 and not args.no_roofline:
        lib = L.load()
        cap = 32768
        prof = lib.em_profile_create(cap)
        ms = (C.c_float * cap)()
        fl = (C.c_double * cap)()
        tg = (C.c_int32 * cap)()
        cnt = C.c_int32(0)
        fam = {}  # tag -> [ms, flops, launches]
        nprof = max(1, min(args.steps, 5 if beam_search is None else 1))
        graph.clear()  # event bracketing needs live launches
        with torch.no_grad():
            for _ in range(nprof):
                lib.em_profile_attach(prof)
                step(do_collate=False)
                lib.em_profile_attach(None)
                L.check(lib.em_profile_read2(prof, ms, fl, tg, cap, C.byref(cnt)), "em_profile_read2")
                for i in range(cnt.value):
                    f = fam.setdefault(tg[i], [0.0, 0.0, 0])
                    f[0] += ms[i]
                    f[1] += fl[i]
                    f[2] += 1
        lib.em_profile_destroy(prof)
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        names = {L.EM_PROF_GEMM: "gemm_kernel<T,EPI,AMODE> (all instantiations)",
                 L.EM_PROF_BLOCK: "block_kernel<MODE> (fused Conformer block, csrc/block.hip)",
                 L.EM_PROF_ATTN: "relpos_attn2_kernel (csrc/attention2.hip)"}
        tot_ms = sum(f[0] for f in fam.values())
        tot_fl = sum(f[1] for f in fam.values())
        launches = sum(f[2] for f in fam.values())
        dom = max(fam, key=lambda t: fam[t][0])
        d_ms, d_fl, d_n = fam[dom]
        achieved = d_fl / (d_ms * 1e-3) / 1e12
        out["roofline"] = {
            "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": None,
            "kernel": names[dom], "launches_per_step": d_n // nprof,
            "avg_launch_us": round(d_ms * 1e3 / d_n, 2),
            "algorithmic_gflop_per_step": round(d_fl / nprof / 1e9, 2),
            "kernel_ms_per_step": round(d_ms / nprof, 3),
            "all_mfma_kernels": {"achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                                 "frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / peak, 4),
                                 "launches_per_step": launches // nprof, "ms_per_step": round(tot_ms / nprof, 3),
                                 "algorithmic_gflop_per_step": round(tot_fl / nprof / 1e9, 2)},
            "families": {names[t].split(" ")[0]: {"ms_per_step": round(f[0] / nprof, 3),
                                                  "tflops": round(f[1] / (f[0] * 1e-3) / 1e12, 1),
                                                  "launches_per_step": f[2] // nprof} for t, f in fam.items()},
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = (cpu_baseline(model) if beam_search is None
                               else cpu_baseline_beam(model, args.beam, args.ctc_weight))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
