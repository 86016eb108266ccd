"""Utterance-parallel inference across the GPUs of one node: one process per GPU, weights replicated,
NO data-path collective; hypotheses are collated with fixed-shape all-gathers (RCCL over xGMI on the
GPU box: backend "nccl"; "gloo" in the CPU tests).

Two ways of handing out the work (round 4):
  * static contiguous slabs (`shard_bounds`, `decode_sharded`) - every rank repeats the same number of steps;
  * DYNAMIC dispatch (`SharedCounter`, `WindowClaimer`, `decode_dynamic`): ranks pull fixed-size units from one
    shared counter (a key of the process group's rendezvous store, `store.add` is atomic), so a slow GPU - the
    pool shows boxes that run the weight-streaming kernels at 0.74x - or a batch of early-ending beams takes
    fewer units instead of holding every step of the job back.  Records then carry their global utterance index
    and are collated ONCE per macro-batch (`gather_variable_records`), not per step.
`RecordRing` is the per-step path of bench.py: a device ring of M steps' fixed-shape records that the decode
kernels write in place, one all-gather + one device->host copy per M steps.

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


# ---------------------------------------------------------------------- dynamic dispatch (round 4)
def work_store(prefix: str = "espnet_amd_work"):
    """A key-value store shared by the ranks of the default process group (the rendezvous store the group was
    initialised with: TCPStore on MASTER_ADDR:MASTER_PORT under torchrun / `_multi_gpu_worker`), behind a prefix so
    that its keys cannot collide with torch.distributed's own."""
    from torch.distributed import distributed_c10d as c10d

    if not dist.is_initialized():
        return None
    return dist.PrefixStore(prefix, c10d._get_default_store())


class SharedCounter:
    """Atomic fetch-and-add over the shared store: `next(n)` returns the first of n consecutive unit indices that now
    belong to the caller.  Without a process group it is a plain local counter (one rank takes everything)."""

    def __init__(self, store, key: str = "unit"):
        self.store, self.key, self._local = store, key, 0

    def next(self, n: int = 1) -> int:
        if self.store is None:
            self._local += n
            return self._local - n
        return int(self.store.add(self.key, n)) - n


_CALL_SEQ = {}


def call_key(name: str, store=None, world: int = 1, timeout_s: float = 300.0) -> str:
    """A counter key for ONE call of a collective entry point: `name` + how many times this process has asked for it, so a
    second decode inside the same process group starts from a fresh counter instead of the previous call's final value
    (ADVICE r04).  The keys agree without communication ONLY if every rank makes the same calls in the same order; a rank
    that skipped or repeated a call (an exception before its decode, a retry on one rank) would silently work on a counter
    of its own and decode every window again.  So with a store the ranks MEET on the key: each adds itself to
    `<key>/arrive`; more arrivals than ranks, or fewer within `timeout_s`, is an error on the ranks that see it, not a
    duplicated transcript.  `release_key` deletes the key's entries when the call is over."""
    _CALL_SEQ[name] = _CALL_SEQ.get(name, 0) + 1
    key = f"{name}_{_CALL_SEQ[name]}"
    if store is not None and world > 1:
        import time

        n = int(store.add(key + "/arrive", 1))
        if n > world:
            raise RuntimeError(f"{key}: {n} arrivals for {world} ranks - the ranks' call sequences have diverged")
        t0 = time.monotonic()
        while n < world:
            if time.monotonic() - t0 > timeout_s:
                raise RuntimeError(f"{key}: only {n} of {world} ranks arrived within {timeout_s:.0f} s - the ranks' call "
                                   f"sequences have diverged (or a rank died before this call)")
            time.sleep(0.002)
            n = int(store.add(key + "/arrive", 0))
        if n > world:
            raise RuntimeError(f"{key}: {n} arrivals for {world} ranks - the ranks' call sequences have diverged")
    return key


def release_key(store, key: str, rank: int = 0):
    """Forget a finished call's counter (rank 0, after the call's closing collective: nobody reads it any more)."""
    if store is None or rank != 0:
        return
    for k in (key, key + "/arrive"):
        try:
            store.delete_key(k)
        except Exception:  # noqa: BLE001  (a store without delete_key: the keys stay, as before)
            pass


class WindowClaimer:
    """`claim(w)` for units visited in increasing order w = 0, 1, 2, ... (every rank walks the same list): True for the
    units this rank owns.  A rank asks the shared counter for its next unit only when it reaches the previous one, so
    how many units a rank ends up with follows how fast it consumes them."""

    def __init__(self, counter: SharedCounter):
        self.counter, self.mine, self.claimed = counter, None, []

    def __call__(self, w: int) -> bool:
        if self.mine is None or self.mine < w:
            self.mine = self.counter.next(1)
        if self.mine == w:
            self.claimed.append(w)
            return True
        return False


def gather_variable_records(rec: torch.Tensor, group=None) -> torch.Tensor:
    """Ranks hold DIFFERENT numbers of fixed-width int32 records (n_r, W): returns all of them (sum n_r, W), rank-major,
    on every rank.  Two collectives per call - the counts, then one all-gather of slabs padded to the largest count -
    which is why it is called once per macro-batch and not per step."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return rec
    n = torch.tensor([rec.size(0)], dtype=torch.int32, device=rec.device)
    counts = torch.empty(world, dtype=torch.int32, device=rec.device)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts = [int(c) for c in counts.tolist()]
    cap, W = max(max(counts), 1), rec.size(1)
    slab = torch.full((cap, W), -1, dtype=torch.int32, device=rec.device)
    slab[: rec.size(0)] = rec
    g = torch.empty((world * cap, W), dtype=torch.int32, device=rec.device)
    dist.all_gather_into_tensor(g, slab, group=group)
    return torch.cat([g[r * cap : r * cap + counts[r]] for r in range(world)], dim=0)


def pack_indexed_records(utt_index: Sequence[int], token_ids: Sequence[Sequence[int]], scores: Sequence[float],
                         max_len: int, device) -> torch.Tensor:
    """(n, max_len + 3) int32 records `[utterance index | ids padded with -1 | count | score bits]`."""
    n = len(utt_index)
    rec = np.full((n, max_len + 3), -1, dtype=np.int32)
    for k, (u, t, s) in enumerate(zip(utt_index, token_ids, scores)):
        t = list(t)
        if len(t) > max_len:
            raise ValueError(f"hypothesis of utterance {u} has {len(t)} tokens, the record holds {max_len}")
        rec[k, 0] = u
        rec[k, 1 : 1 + len(t)] = t
        rec[k, max_len + 1] = len(t)
        rec[k, max_len + 2] = np.float32(s).view(np.int32)
    return torch.from_numpy(rec).to(device)


def unpack_indexed_records(rec, n_items: int) -> List[Tuple[List[int], float]]:
    """Gathered indexed records (any order) -> the n_items hypotheses in global utterance order; every utterance must
    appear exactly once (a unit decoded twice or not at all is an error of the dispatch, not something to paper over)."""
    a = rec.cpu().numpy() if isinstance(rec, torch.Tensor) else np.asarray(rec)
    L = a.shape[1] - 3
    out: List = [None] * n_items
    for row in a:
        u = int(row[0])
        if not 0 <= u < n_items or out[u] is not None:
            raise AssertionError(f"utterance index {u} out of range or decoded twice")
        out[u] = (row[1 : 1 + int(row[L + 1])].tolist(), float(row[L + 2 : L + 3].view(np.float32)[0]))
    missing = [u for u, o in enumerate(out) if o is None]
    if missing:
        raise AssertionError(f"{len(missing)} utterances were decoded by no rank (first: {missing[0]})")
    return out


def decode_dynamic(decode_unit, n_units: int, unit_items, max_len: int, device, group=None, counter=None):
    """Dynamic counterpart of `decode_sharded`: `decode_unit(u)` decodes work unit u (a window of utterances) and
    returns (global utterance indices, token-id lists, scores); ranks claim units from the shared counter until it
    passes n_units; one `gather_variable_records` at the end.  `unit_items` = total number of utterances.  Failure
    protocol as in `decode_sharded`: one rank's exception becomes an exception on every rank, nobody is left inside a
    collective.  Returns (hypotheses in global utterance order, units this rank decoded)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    own_key = None
    if counter is None:
        st = work_store() if world > 1 else None
        own_key = call_key("decode_dynamic", st, world)
        counter = SharedCounter(st, own_key)
    err, recs, mine = None, [], []
    try:
        while True:
            u = counter.next(1)
            if u >= n_units:
                break
            idx, toks, scores = decode_unit(u)
            recs.append(pack_indexed_records(idx, toks, scores, max_len, device))
            mine.append(u)
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
    rec = torch.cat(recs, dim=0) if recs else torch.empty((0, max_len + 3), dtype=torch.int32, device=device)
    out = unpack_indexed_records(gather_variable_records(rec, group), unit_items), mine
    if own_key is not None:
        release_key(counter.store, own_key, rank)  # (behind the closing collective: every rank is past its last claim)
    return out


def decode_dynamic_lanes(start, poll, n_lanes: int, n_units: int, unit_items, max_len: int, device, group=None,
                         counter=None):
    """`decode_dynamic` with several work units IN FLIGHT per rank (round 6: one joint search per HIP stream,
    espnet_amd.nets.batch_beam_search.SearchLanes).  `start(lane, u)` enqueues unit u on lane `lane` and returns at once;
    `poll(lane)` returns None while the lane is busy, else (global utterance indices, token-id lists, scores) of the unit it
    finished.  Units are claimed from the shared counter one at a time, whenever a lane falls free; collation and failure
    protocol as in `decode_dynamic`.  Returns (hypotheses in global utterance order, units this rank decoded)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    own_key = None
    if counter is None:
        st = work_store() if world > 1 else None
        own_key = call_key("decode_dynamic", st, world)
        counter = SharedCounter(st, own_key)
    err, recs, mine = None, [], []
    try:
        unit_of, exhausted = [None] * n_lanes, False
        while True:
            for k in range(n_lanes):  # fill the free lanes
                if unit_of[k] is None and not exhausted:
                    u = counter.next(1)
                    if u >= n_units:
                        exhausted = True
                    else:
                        start(k, u)
                        unit_of[k] = u
            busy = [k for k in range(n_lanes) if unit_of[k] is not None]
            if not busy:
                break
            for k in busy:
                r = poll(k)
                if r is not None:
                    idx, toks, scores = r
                    recs.append(pack_indexed_records(idx, toks, scores, max_len, device))
                    mine.append(unit_of[k])
                    unit_of[k] = None
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
    rec = torch.cat(recs, dim=0) if recs else torch.empty((0, max_len + 3), dtype=torch.int32, device=device)
    out = unpack_indexed_records(gather_variable_records(rec, group), unit_items), mine
    if own_key is not None:
        release_key(counter.store, own_key, rank)
    return out


class RecordRing:
    """Per-step collation for a loop that decodes one fixed-shape batch per step (bench.py): a device ring of M steps'
    records `[ids (B x width) | counts (B) | score bits (B)]` that the decode kernels write IN PLACE (`slot()` hands out
    the views of the current step), and per M steps ONE all-gather (N > 1) + ONE device->host copy into pinned memory,
    delivered to the host while the next macro-batch is being computed (two ring halves).  Per-step collectives made
    every step as slow as its slowest rank and cost three small blits per step; this costs neither."""

    def __init__(self, rank: int, world: int, B: int, width: int, M: int, device, group=None, host_gloo: bool = False):
        self.rank, self.world, self.B, self.width, self.M, self.group, self.host_gloo = rank, world, B, width, M, group, host_gloo
        self.rec = B * (width + 2)
        self.ring = torch.full((2, M, self.rec), -1, dtype=torch.int32, device=device)
        self.ring_f = self.ring.view(torch.float32)
        self.ring_f[:, :, B * width + B :] = 0.0  # scores: 0 unless the step writes them
        self.pinned = [torch.empty((world, M, self.rec), dtype=torch.int32).pin_memory() if torch.cuda.is_available()
                       else torch.empty((world, M, self.rec), dtype=torch.int32) for _ in range(2)]
        self.events = [torch.cuda.Event(), torch.cuda.Event()] if torch.cuda.is_available() else [None, None]
        self.k = 0                 # steps committed
        self.pending = None        # (half, steps in it) in flight to the host
        self.delivered = 0         # steps delivered to the host
        self.last = None           # (ids (world*B, width), counts, scores) numpy arrays of the last delivered step

    def slot(self):
        """Views of the current step's record: ids (B, width) i32, counts (B,) i32, scores (B,) f32."""
        half, j = (self.k // self.M) & 1, self.k % self.M
        B, w = self.B, self.width
        return (self.ring[half, j, : B * w].view(B, w), self.ring[half, j, B * w : B * w + B],
                self.ring_f[half, j, B * w + B :])

    def commit(self):
        self.k += 1
        if self.k % self.M == 0:
            self._flush(((self.k - 1) // self.M) & 1, self.M)

    def _flush(self, half: int, steps: int):
        src = self.ring[half]
        if self.world > 1:
            # (the gathered table is (world * M, rec): all_gather_into_tensor concatenates along dim 0)
            if self.host_gloo:  # developer check of the control flow on a one-GPU box
                src_h = src.cpu()
                g = torch.empty((self.world * src_h.size(0), src_h.size(1)), dtype=torch.int32)
                dist.all_gather_into_tensor(g, src_h, group=self.group)
            else:
                g = torch.empty((self.world * src.size(0), src.size(1)), dtype=torch.int32, device=src.device)
                dist.all_gather_into_tensor(g, src, group=self.group)
            g = g.view(self.world, src.size(0), src.size(1))
        else:
            g = src.unsqueeze(0)
        if self.rank == 0:
            self.pinned[half].copy_(g, non_blocking=True)
            if self.events[half] is not None:
                self.events[half].record()
        prev, self.pending = self.pending, (half, steps)
        if prev is not None:
            self._deliver(*prev)

    def _deliver(self, half: int, steps: int):
        if self.rank == 0:
            if self.events[half] is not None:
                self.events[half].synchronize()
            a = self.pinned[half].numpy()[:, steps - 1]  # the macro-batch's last step, all ranks: (world, rec)
            B, w = self.B, self.width
            # (copies: at world == 1 the slices are views of the pinned half, which the ring overwrites two macro-batches on)
            self.last = (a[:, : B * w].reshape(self.world * B, w).copy(), a[:, B * w : B * w + B].reshape(-1).copy(),
                         a[:, B * w + B :].copy().view(np.float32).reshape(-1))
        self.delivered += steps

    def drain(self):
        """Flush a partial macro-batch and deliver everything (inside the caller's timed region)."""
        if self.k % self.M:
            self._flush((self.k // self.M) & 1, self.k % self.M)
            self.k += self.M - self.k % self.M  # the next step starts a fresh half
        if self.pending is not None:
            self._deliver(*self.pending)
            self.pending = None

    def step_records(self, half_array, j):
        """(ids, counts, scores) of step j of a delivered macro-batch array (world, M, rec): rank-major = global
        utterance order of that step."""
        a = half_array[:, j]
        B, w = self.B, self.width
        return (a[:, : B * w].reshape(self.world * B, w), a[:, B * w : B * w + B].reshape(-1),
                a[:, B * w + B :].copy().view(np.float32).reshape(-1))
