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
