"""CPU (gloo, world_size 2): the utterance-sharding + hypothesis-collation path of
espnet_amd/distributed.py.  The decode itself needs an MI355X, so a deterministic stand-in
produces each utterance's tokens; what is under test is the N > 1 plumbing (slab bounds, fixed
shape padding, all-gather order)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from espnet_amd.distributed import decode_sharded, shard_bounds


def fake_decode(lo, hi):
    toks = [[(7 * u + j) % 5000 for j in range(3 + u % 5)] for u in range(lo, hi)]
    scores = [-1.5 * u for u in range(lo, hi)]
    return toks, scores


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = decode_sharded(fake_decode, n_items, max_len=8, device="cpu")
        q.put((rank, out))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 8, 1])
def test_two_ranks_collate_in_utterance_order(n_items):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_t, want_s = fake_decode(0, n_items)
    for r in range(world):  # every rank holds the full, ordered result
        assert [t for t, _ in got[r]] == want_t
        assert [s for _, s in got[r]] == pytest.approx(want_s)


def test_shard_bounds_partition():
    for n in range(0, 40):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_path():
    out = decode_sharded(fake_decode, 5, max_len=8, device="cpu")
    t, s = fake_decode(0, 5)
    assert [a for a, _ in out] == t


# ---- bench.py's per-step record layout: greedy tokens (B, T) i32 padded with -1, token counts (B,), scores (B,),
#      collated by gather_records + unpack_records(as_arrays=True) -- the functions HypothesisSink in bench.py calls
def _bench_records(rank, B, T):
    g = torch.Generator().manual_seed(100 + rank)
    lens = torch.randint(0, T + 1, (B,), generator=g, dtype=torch.int32)
    ids = torch.randint(1, 4999, (B, T), generator=g, dtype=torch.int32)
    ids[torch.arange(T).unsqueeze(0) >= lens.unsqueeze(1)] = -1
    return ids, lens, torch.full((B,), float(rank), dtype=torch.float32)


def _bench_worker(rank, world, port, B, T, q):
    from espnet_amd.distributed import gather_records, unpack_records

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = gather_records(*_bench_records(rank, B, T))
        ids, lens, sc = unpack_records(*g, world * B, world, as_arrays=True)
        hyps = unpack_records(*g, world * B, world)
        q.put((rank, ids.tolist(), lens.tolist(), sc.tolist(), hyps))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_bench_record_layout_two_ranks():
    world, B, T = 2, 4, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, B, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [_bench_records(r, B, T) for r in range(world)]
    w_ids = torch.cat([w[0] for w in want]).tolist()
    w_lens = torch.cat([w[1] for w in want]).tolist()
    w_sc = torch.cat([w[2] for w in want]).tolist()
    for r in range(world):  # rank-major == global utterance order when every rank holds a full slab
        ids, lens, sc, hyps = got[r]
        assert ids == w_ids and lens == w_lens and sc == w_sc
        assert len(hyps) == world * B
        for u, (toks, s) in enumerate(hyps):
            assert toks == w_ids[u][: w_lens[u]] and s == w_sc[u]


def test_record_rows_uneven_slabs():
    from espnet_amd.distributed import record_rows

    # 7 utterances over 2 ranks: slabs of 4 (rank 0 holds 4, rank 1 holds 3 + one padding record)
    assert record_rows(7, 2, 4).tolist() == [0, 1, 2, 3, 4, 5, 6]
    # 5 over 4 ranks: slab 2; ranks hold 2, 1, 1, 1
    assert record_rows(5, 4, 2).tolist() == [0, 1, 2, 4, 6]


# ---- `inference(--ngpu N)`: windows claimed from a shared counter -> shard files -> indexed-record collation ->
#      merged files in key order (dynamic dispatch, round 4)
def _fake_window_decoder(keys, window, delay_s=0.0):
    """Stand-in for the single-GPU `inference(window_claim=...)` of one rank: walks the windows of the key list in order,
    decodes the ones `claim` grants it and writes the files the real loop would write."""
    import time

    def decode_claimed(claim, shard_dir):
        mine = {}
        for w in range((len(keys) + window - 1) // window):
            if not claim(w):
                continue
            time.sleep(delay_s)
            for k in keys[w * window : (w + 1) * window]:
                u = int(k[3:])
                mine[k] = ([(11 * u + j) % 4999 + 1 for j in range(2 + u % 4)], -0.5 * u)
        d = shard_dir / "1best_recog"
        d.mkdir(parents=True, exist_ok=True)
        with (d / "token_int").open("w") as f1, (d / "score").open("w") as f2, (d / "text").open("w") as f3:
            for k, (t, s) in mine.items():
                f1.write(f"{k} {' '.join(map(str, t))}\n")
                f2.write(f"{k} {s}\n")
                f3.write(f"{k} text of {k}\n")
        return mine, {"utterances": len(mine)}

    return decode_claimed


def _ngpu_worker(rank, world, port, keys, out_dir, window, delays, q):
    from espnet_amd.bin.asr_inference import sharded_decode_rank

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        hyps, st = sharded_decode_rank(_fake_window_decoder(keys, window, delays[rank]), keys, out_dir, 1, 16, "cpu", window)
        q.put((rank, hyps, st))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _run_ngpu(tmp_path, keys, window, delays):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ngpu_worker, args=(r, world, port, keys, str(tmp_path), window, delays, q))
             for r in range(world)]
    for p in procs:
        p.start()
    got = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_ngpu_dynamic_inference_merges_in_key_order(tmp_path):
    from pathlib import Path

    keys = [f"utt{u}" for u in (5, 3, 9, 0, 7)]  # input order is not sorted: the merged files must keep it
    got = _run_ngpu(tmp_path, keys, 2, [0.0, 0.0])
    want = {k: ([(11 * int(k[3:]) + j) % 4999 + 1 for j in range(2 + int(k[3:]) % 4)], -0.5 * int(k[3:])) for k in keys}
    for r in range(2):  # every rank holds the full result in key order
        assert [t for t, _ in got[r][0]] == [want[k][0] for k in keys]
        assert [s for _, s in got[r][0]] == pytest.approx([want[k][1] for k in keys])
    w0, w1 = got[0][1]["windows"], got[1][1]["windows"]
    assert sorted(w0 + w1) == [0, 1, 2] and not set(w0) & set(w1)  # every window decoded exactly once
    assert got[0][1]["utterances"] + got[1][1]["utterances"] == len(keys)
    rows = (Path(tmp_path) / "1best_recog" / "token_int").read_text().splitlines()
    assert [ln.split()[0] for ln in rows] == keys
    assert [[int(t) for t in ln.split()[1:]] for ln in rows] == [want[k][0] for k in keys]
    assert (Path(tmp_path) / "1best_recog" / "text").read_text().splitlines()[2] == "utt9 text of utt9"


def test_ngpu_dynamic_dispatch_takes_the_slow_rank_out_of_the_critical_path(tmp_path):
    """one rank four times slower (a slow-state GPU).  With static slabs the job would take
    20 windows x 80 ms = 1.6 s on the slow rank; with the shared counter the fast rank takes ~4/5 of the windows and
    the job finishes in about total / (sum of rates): asserted at 1.35x that ideal, and every window exactly once."""
    import time

    n_win, window = 40, 2
    keys = [f"utt{u}" for u in range(n_win * window)]
    fast, slow = 0.02, 0.08
    t0 = time.perf_counter()
    got = _run_ngpu(tmp_path, keys, window, [fast, slow])
    wall = time.perf_counter() - t0
    w0, w1 = got[0][1]["windows"], got[1][1]["windows"]
    assert sorted(w0 + w1) == list(range(n_win))
    assert len(w0) >= 2.5 * len(w1), (len(w0), len(w1))       # ~4 : 1
    busy = max(len(w0) * fast, len(w1) * slow)                  # the ranks' own decode time (spawn / rendezvous aside)
    ideal = n_win / (1 / fast + 1 / slow)                       # total work / sum of rates = 0.64 s
    static = (n_win // 2) * slow                                # what contiguous slabs cost: 1.6 s
    assert busy <= 1.35 * ideal < static, (busy, ideal, static, wall)


# ---- RecordRing: bench.py's per-step records, collated once per M steps
def _ring_worker(rank, world, port, B, W, M, steps, q):
    from espnet_amd.distributed import RecordRing

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ring = RecordRing(rank, world, B, W, M, "cpu")
        seen = []
        for k in range(steps):
            ids, cnt, sc = ring.slot()
            ids.fill_(-1)
            n = (k + rank) % (W + 1)
            ids[:, :n] = 100 * rank + k
            cnt.fill_(n)
            sc.fill_(float(k) + 0.5 * rank)
            ring.commit()
            if ring.last is not None:
                seen.append((ring.delivered, ring.last[0][:, 0].tolist(), ring.last[1].tolist(), ring.last[2].tolist()))
        ring.drain()
        last = None if ring.last is None else (ring.last[0][:, 0].tolist(), ring.last[1].tolist(), ring.last[2].tolist())
        q.put((rank, ring.delivered, last, len(seen)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("steps", [37, 32, 5])
def test_record_ring_two_ranks(steps):
    world, B, W, M = 2, 3, 6, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_worker, args=(r, world, port, B, W, M, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r][0] == steps  # every step delivered, none twice
    # rank 0 owns the results: the last delivered step is the job's last step, rank-major
    k = steps - 1
    first_col, counts, scores = got[0][1]
    want_counts = [(k + r) % (W + 1) for r in range(world) for _ in range(B)]
    assert counts == want_counts
    assert scores == [float(k) + 0.5 * r for r in range(world) for _ in range(B)]
    assert first_col == [(100 * r + k) if (k + r) % (W + 1) > 0 else -1 for r in range(world) for _ in range(B)]


def test_gather_variable_records_single_process():
    from espnet_amd.distributed import pack_indexed_records, unpack_indexed_records

    rec = pack_indexed_records([2, 0, 1], [[5, 6], [], [7]], [-1.0, 0.0, -2.5], 4, "cpu")
    assert unpack_indexed_records(rec, 3) == [([], 0.0), ([7], -2.5), ([5, 6], -1.0)]
    with pytest.raises(AssertionError, match="decoded by no rank"):
        unpack_indexed_records(rec[:2], 3)
    with pytest.raises(AssertionError, match="decoded twice"):
        unpack_indexed_records(torch.cat([rec, rec[:1]]), 3)


# ---- failure of one rank: no hang (a rank that failed before the collective left its peers in the all-gather)
def _failing_decode(lo, hi):
    if lo > 0:  # rank 1's slab
        raise ValueError("hypothesis 0 has 99 tokens, the collective's record holds 8")
    return fake_decode(lo, hi)


def _failing_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        decode_sharded(_failing_decode, 6, max_len=8, device="cpu")
        q.put((rank, "no exception"))
    except Exception as e:
        q.put((rank, f"{type(e).__name__}: {e}"))
    dist.destroy_process_group()


def test_one_failing_rank_raises_on_every_rank():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1].startswith("ValueError")           # the rank that failed re-raises its own error
    assert got[0].startswith("RuntimeError") and "another rank failed" in got[0]  # its peer is told, not left waiting


def _report_then_hang(rank, q):
    import time

    if rank == 1:
        q.put((rank, RuntimeError("rank 1: ValueError: bad audio")))
        raise SystemExit(1)
    time.sleep(600)  # a peer stuck in a collective nobody will complete


def _die_silently(rank, q):
    import time

    if rank == 1:
        os._exit(9)
    time.sleep(600)


@pytest.mark.parametrize("target", [_report_then_hang, _die_silently])
def test_collect_ranks_terminates_survivors(target):
    from espnet_amd.bin.asr_inference import _collect_ranks

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, q)) for r in range(2)]
    for p in procs:
        p.start()
    with pytest.raises(RuntimeError, match="bad audio|exited with 9"):
        _collect_ranks(procs, q, poll_s=0.2)
    assert all(not p.is_alive() for p in procs)


def test_bench_gpus_n_without_launcher_becomes_the_launcher(monkeypatch):
    """`python bench.py --gpus 8` with no WORLD_SIZE in the environment must re-launch itself as the contract's
    torchrun command line (one rank per GPU, 127.0.0.1 rendezvous) instead of dying on the world-size assertion."""
    import importlib.util
    import subprocess
    import sys
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("bench_under_test", Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 0
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _diverging_worker(rank, world, port, q):
    """Rank 1 makes one call of the collective entry point more than rank 0 before the call both make: its call sequence
    is one ahead, and without the meeting on the key it would silently work on a counter of its own."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from espnet_amd import distributed as D

    st = D.work_store()
    try:
        k1 = D.call_key("decode_x", st, world)               # both ranks: agree
        if rank == 1:
            D._CALL_SEQ["decode_x"] += 1                     # (the skipped / repeated call, without its wait)
        try:
            D.call_key("decode_x", st, world, timeout_s=2.0)
            q.put((rank, k1, "no error"))
        except RuntimeError as e:
            q.put((rank, k1, str(e)))
        D.release_key(st, k1, rank)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_diverged_call_sequences_raise_instead_of_duplicating_work():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_diverging_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: (k, msg) for r, k, msg in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0] == "decode_x_1"            # the call both made met on one key
    for r in range(world):                                   # the diverged one is an error on every rank that waits for it
        assert "diverged" in got[r][1], got


def _lanes_worker(rank, world, port, n_units, q):
    """decode_dynamic_lanes over gloo: three lanes per rank, a unit needs a few polls before it is done (rank 1 twice as many)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from espnet_amd import distributed as D

    B = 2
    lanes = {}
    try:
        def start(k, u):
            assert k not in lanes
            lanes[k] = [u, (2 + u % 3) * (1 + rank)]

        def poll(k):
            import time

            time.sleep(0.004 * (1 + rank))  # (a poll waits for the lane's stream: rank 1's take twice as long)
            lanes[k][1] -= 1
            if lanes[k][1] > 0:
                return None
            u = lanes.pop(k)[0]
            toks, scores = fake_decode(u * B, u * B + B)
            return list(range(u * B, u * B + B)), toks, scores

        dist.barrier()  # (both ranks start claiming together)
        hyps, mine = D.decode_dynamic_lanes(start, poll, 3, n_units, n_units * B, 8, "cpu")
        q.put((rank, hyps, mine))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_dynamic_dispatch_with_lanes_two_ranks():
    world, n_units, B = 2, 24, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lanes_worker, args=(r, world, port, n_units, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: (h, m) for r, h, m in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_t, want_s = fake_decode(0, n_units * B)
    for r in range(world):  # every rank holds the full result in global utterance order
        assert [t for t, _ in got[r][0]] == want_t
        assert [s for _, s in got[r][0]] == pytest.approx(want_s)
    assert sorted(got[0][1] + got[1][1]) == list(range(n_units))  # every unit decoded exactly once
    assert len(got[0][1]) > len(got[1][1]) > 0  # the rank whose units take a quarter of the time took more of them
