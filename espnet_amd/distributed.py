"""Utterance-parallel inference across the GPUs of one node: one process per GPU, utterances
sharded in contiguous slabs, weights replicated, NO data-path collective; hypotheses are collated
with one fixed-shape all-gather (RCCL over xGMI on the GPU box: backend "nccl"; "gloo" in the CPU
tests).

The reference scales inference the same way, with independent processes over split key files
merged afterwards (egs2/TEMPLATE/asr1/asr.sh:1589-1619, 1636-1648; `ngpu > 1` is rejected inside
one process, espnet2/bin/asr_inference.py:760-765).  SURVEY.md §8(e).
"""
import functools
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slab [lo, hi) of rank `rank`; slab sizes differ by at most one (the first
    n_items % world ranks get the extra item), like `split_scps` in asr.sh."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_hypotheses(token_ids: Sequence[Sequence[int]], scores: Sequence[float], max_len: int,
                    slab: int, device) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Fixed-shape records for the collective: ids (slab, max_len) i32 padded with -1, lengths
    (slab,) i32 (-1 marks an empty padding record), scores (slab,) f32."""
    ids = torch.full((slab, max_len), -1, dtype=torch.int32)
    lens = torch.full((slab,), -1, dtype=torch.int32)
    sc = torch.zeros(slab, dtype=torch.float32)
    for k, (t, s) in enumerate(zip(token_ids, scores)):
        t = list(t)
        if len(t) > max_len:  # a silently shortened hypothesis would be a wrong result on every rank
            raise ValueError(f"hypothesis {k} has {len(t)} tokens, the collective's record holds {max_len}")
        ids[k, : len(t)] = torch.tensor(t, dtype=torch.int32)
        lens[k] = len(t)
        sc[k] = float(s)
    return ids.to(device), lens.to(device), sc.to(device)


def gather_records(ids: torch.Tensor, lens: torch.Tensor, scores: torch.Tensor, group=None):
    """The collective itself: ONE all-gather of the ranks' fixed-shape slabs (RCCL over xGMI on the GPU box; the
    ≈ 32 KB per rank are latency-bound, so token ids, counts and scores travel as one int32 record
    `[ids (L) | count | score bits]` per hypothesis instead of three collectives).  Returns the (world * slab, ...)
    tensors on the callers' device, rank-major, as views of the gathered record table; nothing is copied to the
    host and nothing synchronises."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return ids, lens, scores
    slab, L = ids.shape
    rec = torch.cat([ids.to(torch.int32), lens.to(torch.int32).view(slab, 1),
                     scores.to(torch.float32).contiguous().view(torch.int32).view(slab, 1)], dim=1).contiguous()
    g = torch.empty((world * slab, L + 2), dtype=torch.int32, device=ids.device)
    dist.all_gather_into_tensor(g, rec, group=group)
    return g[:, :L], g[:, L], g[:, L + 1].contiguous().view(torch.float32)


@functools.lru_cache(maxsize=64)
def _record_rows_np(n_items: int, world: int, slab: int):
    rows = []
    for r in range(world):
        lo, hi = shard_bounds(n_items, r, world)
        rows.extend(range(r * slab, r * slab + hi - lo))
    return np.asarray(rows, dtype=np.int64)


def record_rows(n_items: int, world: int, slab: int) -> torch.Tensor:
    """Row of the gathered (world * slab) record table that holds utterance u, for u < n_items (global utterance
    order; rank r's slab starts at r * slab and holds shard_bounds(n_items, r, world))."""
    return torch.from_numpy(_record_rows_np(n_items, world, slab).copy())


def unpack_records(g_ids, g_lens, g_sc, n_items: int, world: int, as_arrays: bool = False):
    """Host side of the collation: gathered HOST tensors (or numpy arrays) -> the n_items hypotheses in global
    utterance order.  as_arrays=True keeps them as (ids (n_items, max_len) padded with -1, lens, scores) numpy
    arrays -- no per-utterance Python, which is what a rank that only forwards results (or a benchmark loop)
    wants.  Plain numpy on purpose: this runs once per step next to a GPU that is being fed by the same thread."""
    a_ids, a_lens, a_sc = (t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t) for t in (g_ids, g_lens, g_sc))
    slab = a_ids.shape[0] // world
    rows = _record_rows_np(n_items, world, slab)
    ids, lens, sc = a_ids[rows], a_lens[rows], a_sc[rows]
    if n_items and int(lens.min()) < 0:
        raise AssertionError("padding record inside a rank's slab")
    if as_arrays:
        return ids, lens, sc
    return [(ids[k, : int(lens[k])].tolist(), float(sc[k])) for k in range(n_items)]


def gather_hypotheses(ids: torch.Tensor, lens: torch.Tensor, scores: torch.Tensor, n_items: int,
                      group=None) -> List[Tuple[List[int], float]]:
    """One all-gather per tensor of the ranks' fixed-shape slabs; returns the n_items hypotheses in
    global utterance order on EVERY rank (rank 0 is the one that writes results)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    g_ids, g_lens, g_sc = gather_records(ids, lens, scores, group)
    return unpack_records(g_ids.cpu(), g_lens.cpu(), g_sc.cpu(), n_items, world)


def decode_sharded(decode_fn, n_items: int, max_len: int, device, group=None):
    """decode_fn(lo, hi) -> (list of token-id lists, list of scores) for utterances [lo, hi).
    Every rank decodes its slab and the hypotheses are collated in utterance order."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_bounds(n_items, rank, world)
    slab = (n_items + world - 1) // world
    # A rank that fails before the collective (bad audio, out of memory, a hypothesis longer than the record) must
    # not leave its peers waiting inside the all-gather: every rank first agrees on an error flag (one tiny
    # all-reduce), and one rank's failure becomes an exception on every rank.
    err = None
    ids = lens = sc = None
    try:
        toks, scores = decode_fn(lo, hi) if hi > lo else ([], [])
        ids, lens, sc = pack_hypotheses(toks, scores, max_len, slab, device)
    except Exception as e:  # re-raised below, after the peers have been told
        err = e
    if world > 1:
        flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        failed = int(flag.item()) != 0
    else:
        failed = err is not None
    if err is not None:
        raise err
    if failed:
        raise RuntimeError(f"rank {rank}: another rank failed before the hypothesis collation; nothing was gathered")
    return gather_hypotheses(ids, lens, sc, n_items, group)
