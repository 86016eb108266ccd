"""Joint searches in flight, one HOST THREAD per search (round 6 probe).  With one host thread two searches in flight saturate at
~60 ms per batch of 16 whatever the number of lanes (profiles/r06z_beam_lanes_ab.txt): 249 label steps x 47 launches x 2 lanes
at ~2.5 us of hipLaunchKernel each IS the 0.24 ms per step - the host's launch rate, not the device.  ctypes releases the GIL
inside em_search_steps, so a thread per lane launches in parallel.  Prints audio-s/s for 1 .. 4 threads (each: encoder +
search_batch per batch on its own stream, configs[2]: Conformer-large, beam 10, B = 16).
`python tools/experiments/beam_threads_probe.py [B]`"""
import sys
import threading
import time
from pathlib import Path
from types import SimpleNamespace

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from espnet_amd.nets.batch_beam_search import build_beam_search
    from espnet_amd.tasks.asr import ASRTask

    torch.manual_seed(0)
    model = ASRTask.build_model(bench.model_config("large", "bfloat16")).to(dev).eval()
    bs0 = build_beam_search(model, beam_size=10, ctc_weight=0.3, penalty=0.0, token_list=model.token_list)
    wav = bench.synth_batch(0, B).to(dev)
    lens = [bench.N_SAMPLES] * B
    for n_thr in (1, 2, 3, 4, 6):
        searches = [bs0] + [bs0.clone() for _ in range(n_thr - 1)]
        streams = bench.StepPipeline._pick(dev, n_thr)[0]
        n_batches = 2 * n_thr
        results = [None] * n_thr

        def work(k, n):
            with torch.no_grad(), torch.cuda.stream(streams[k]):
                for _ in range(n):
                    st = model.encode_device(wav, lens)
                    results[k] = searches[k].search_batch(st.enc_act, st.olens)
                torch.cuda.current_stream().synchronize()

        for k in range(n_thr):  # warm-up, one lane at a time (hipGraph capture, workspaces)
            work(k, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        thr = [threading.Thread(target=work, args=(k, n_batches // n_thr)) for k in range(n_thr)]
        for t in thr:
            t.start()
        for t in thr:
            t.join()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        same = all(results[k][0][0].yseq.tolist() == results[0][0][0].yseq.tolist() for k in range(n_thr))
        print(f"{n_thr} threads: {n_batches * B * 10.0 / el:8.1f} audio-s/s, {el / n_batches * 1e3:6.1f} ms per batch, lanes agree: {same}", flush=True)


if __name__ == "__main__":
    main()
