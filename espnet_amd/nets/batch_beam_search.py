"""BatchBeamSearch on the MI355X: label-synchronous joint CTC/attention beam search, batched over
utterances x beam, fully device resident (csrc/search.hip + csrc/decoder.hip).

Mirrors espnet2/legacy/nets/beam_search.py:36-126 (constructor: scorers / weights / beam_size /
vocab_size / sos / eos / pre_beam_ratio / pre_beam_score_key / normalize_length; zero-weight
scorers are dropped) and :385-498 (`forward(x, maxlenratio, minlenratio) -> n-best Hypothesis
list`), with BatchBeamSearch's step semantics (legacy/nets/batch_beam_search.py:253-423).
`search_batch` is the utterance-batched entry the MI355X path adds; `forward` keeps the
reference's single-utterance signature.

Supported scorers: "decoder" (TransformerDecoder), "ctc" (CTCPrefixScorer), "length_bonus" and
"lm" (espnet_amd.lm.transformer_lm.TransformerLM).  An n-gram scorer is SURVEY.md §8(f) "next".
"""
import ctypes as C
import logging
import os
from typing import Dict, List, Optional

import torch

from espnet_amd import lib as L
from espnet_amd.nets.beam_search import Hypothesis
from espnet_amd.nets.scorers.ctc import CTCPrefixScorer
from espnet_amd.nets.scorers.length_bonus import LengthBonus

logger = logging.getLogger(__name__)

_I32 = {"xlens", "maxlens", "minlens", "tok", "parent", "anc_a", "anc_b", "alive", "cand_tok",
        "sel_idx", "end_count", "end_pos", "end_slot", "end_forced", "done", "step"}
_ACT = {"xn", "qkv", "qs", "ctx", "hbuf", "self_k", "self_v", "mem_kv", "mem_vT", "mem_kf", "mem_vf", "lm_e", "lm_xn",
        "lm_qkv", "lm_ctx", "lm_h", "lm_k", "lm_v", "rnn_hs", "rnn_hin"}
_ZERO = {"mem_vT", "rnn_hs", "rnn_cs", "rnn_hin"}  # pad columns / tails that must read as zero


class BeamSearch:
    def __init__(self, scorers: Dict[str, object], weights: Dict[str, float], beam_size: int,
                 vocab_size: int, sos: int, eos: int, token_list: List[str] = None,
                 pre_beam_ratio: float = 1.5, pre_beam_score_key: str = None,
                 return_hs: bool = False, hyp_primer: List[int] = None,
                 normalize_length: bool = False):
        if return_hs or hyp_primer is not None:
            raise NotImplementedError("return_hs / hyp_primer are outside the MI355X hot path")
        self.weights = weights
        self.scorers, self.full_scorers, self.part_scorers = {}, {}, {}
        for k, v in scorers.items():  # beam_search.py:80-96
            w = weights.get(k, 0)
            if w == 0 or v is None:
                continue
            if k not in ("decoder", "ctc", "length_bonus", "lm"):
                raise NotImplementedError(f"scorer {k!r}: SURVEY.md §8(f) 'next' (n-gram)")
            self.scorers[k] = v
            (self.part_scorers if isinstance(v, CTCPrefixScorer) else self.full_scorers)[k] = v
        self.sos, self.eos = sos, eos
        self.token_list = token_list
        self.pre_beam_size = int(pre_beam_ratio * beam_size)
        self.beam_size = beam_size
        self.n_vocab = vocab_size
        if (pre_beam_score_key is not None and pre_beam_score_key != "full"
                and pre_beam_score_key not in self.full_scorers):
            raise KeyError(f"{pre_beam_score_key} is not found in {self.full_scorers}")
        if pre_beam_score_key not in (None, "full"):
            raise NotImplementedError("pre_beam_score_key other than 'full'")
        self.pre_beam_score_key = pre_beam_score_key
        self.do_pre_beam = (self.pre_beam_score_key is not None
                            and self.pre_beam_size < self.n_vocab and len(self.part_scorers) > 0)
        self.normalize_length = normalize_length
        if not any(k in self.scorers for k in ("decoder", "ctc", "lm")):
            raise ValueError("beam search needs the decoder, the ctc and/or the lm scorer")
        self._bufs = {}
        self._graphs = {}
        self.max_live_shapes = 3  # buffer sets (and their captured graphs) kept for recurring batch shapes
        self.step_chunk = 16  # steps enqueued between two polls of the `done` flags
        # use_hipgraph: replay a captured hipGraph of `step_chunk` search steps instead of em_search_steps' eager launches.  On by
        # default through round 5; round 6 measured the eager launches (issued from C, ~2.5 us each) no slower - one search alone
        # 0.343 against 0.352 ms per label step, four searches in flight 46.1 against 47.5 - 49.3 ms per batch of 16
        # (profiles/r06z_beam_eager_ab.txt) - and a capture cannot run while another host thread launches (SearchLanes'
        # threaded lanes: hipErrorStreamCaptureInvalidated).  Off by default; ESPNET_AMD_SEARCH_GRAPH=1 turns it on.
        self.use_hipgraph = False
        # (Round 2 also cut the batch into sub-batch searches on concurrent HIP streams: 0.646 -> 0.740 (2 lanes) ->
        # 1.332 ms (4) per label step, profiles/r02_experiments_not_kept.txt - the launches of the lanes do not
        # overlap.  Removed in round 3; the search is one generator over the whole batch.)
        if os.environ.get("ESPNET_AMD_SEARCH_GRAPH") == "1":
            self.use_hipgraph = True


class BatchBeamSearch(BeamSearch):
    def clone(self):
        """Another search object over the SAME scorers and weights with its own buffer sets and hipGraphs: what a second
        search in flight needs (`SearchLanes`)."""
        import copy

        c = copy.copy(self)
        c._bufs, c._graphs = {}, {}
        return c

    # ------------------------------------------------------------------ buffers
    def _alloc(self, dev, act, B, W, V, T, Tpad, NC, Lmax, cap, d, ff, nl, lm=None, online=False):
        key = (str(dev), act, B, W, V, T, Tpad, NC, Lmax, cap, d, ff, nl,
               None if lm is None else lm.search_key(), online)
        if key in self._bufs:
            self._bufs[key] = self._bufs.pop(key)  # most recently used last
            return self._bufs[key]
        while len(self._bufs) >= self.max_live_shapes:  # evict the least recently used shape and the
            old = self._bufs.pop(next(iter(self._bufs)))
            for gk in [gk for gk in self._graphs if gk[0] == id(old)]:  # graphs captured over its buffers
                del self._graphs[gk]
        n = B * W
        use_dec, use_ctc = "decoder" in self.scorers, "ctc" in self.scorers
        shapes = dict(
            xlens=(B,), maxlens=(B,), minlens=(B,), tok=(Lmax, n), parent=(Lmax, n),
            anc_a=(n, Lmax), anc_b=(n, Lmax), alive=(n,), run_score=(n,), run_sdec=(n,),
            run_sctc=(n,), run_slen=(n,), s_prev=(n,),
            r_a=(n, T, 2) if use_ctc else (1,), r_b=(n, T, 2) if use_ctc else (1,),
            ctc_lpT=(V, B * T) if use_ctc else None,
            cand_tok=(n, NC), cand_full=(n, NC), cand_psi=(n, NC), cand_total=(n, NC),
            sel_idx=(n,), sel_total=(n,), step=(2,), end_count=(B,), end_pos=(B, cap), end_slot=(B, cap),
            end_forced=(B, cap), end_score=(B, cap), end_sdec=(B, cap), end_sctc=(B, cap),
            end_slen=(B, cap), best_all=(B,), best_by_len=(B, Lmax + 2), done=(B,))
        if use_dec:
            shapes.update(x=(n, d), xn=(n, d), qkv=(n, 3 * d), qs=(n, d), ctx=(n, d), hbuf=(n, ff),
                          dec_logp=(n, V), self_k=(nl, Lmax, n, d), self_v=(nl, Lmax, n, d),
                          mem_kv=(nl, B * T, 2 * d), mem_vT=(nl, B, d, Tpad))
            if (act == torch.bfloat16 and not online and d % 64 == 0 and n >= 96
                    and os.environ.get("ESPNET_AMD_NO_MEM_FRAG") != "1"):
                # round 6: the memory's K / V^T fragment-major as well (EmSearchBuffers.mem_kf / mem_vf): 1 KiB operand loads in the
                # source attention of a label step.  ESPNET_AMD_NO_MEM_FRAG=1: developer A/B switch (read at allocation).
                shapes.update(mem_kf=(nl, B, d, Tpad), mem_vf=(nl, B, d, Tpad))
        if lm is not None:
            shapes.update(lm.search_buffers(n, V, Lmax, B, cap))
        if online:  # em_search_online_* (batch_beam_search_online.py)
            shapes.update(online_best=(n, 8), online_psi=(n,), online_snap=(n, 8))
        t = {}
        for name, shp in shapes.items():
            if shp is None:
                continue
            dt = torch.int32 if name in _I32 else (act if name in _ACT else torch.float32)
            t[name] = (torch.zeros if name in _ZERO else torch.empty)(shp, dtype=dt, device=dev)
        self._bufs[key] = t
        return t

    # ------------------------------------------------------------------ batched search
    @torch.no_grad()
    def search_batch(self, enc_act: torch.Tensor, olens: List[int], maxlenratio: float = 0.0,
                     minlenratio: float = 0.0) -> List[List[Hypothesis]]:
        """enc_act (B, T, d) encoder output in the compute dtype ON THE GPU; olens host ints.
        Returns the ended hypotheses of every utterance, best first (beam_search.py:453-461).  An utterance
        that ends no hypothesis is searched again with minlenratio lowered by 0.1, as BeamSearch.forward does
        for its single utterance (beam_search.py:462-474), until it yields one or minlenratio < 0.1."""
        out = self._search_once(enc_act, olens, maxlenratio, minlenratio)
        steps = self.last_steps  # of THIS search: the back-off below recurses and must not overwrite it
        empty = [b for b, h in enumerate(out) if len(h) == 0]
        if empty and minlenratio >= 0.1:
            idx = torch.tensor(empty, device=enc_act.device)
            again = self.search_batch(enc_act.index_select(0, idx), [olens[b] for b in empty], maxlenratio,
                                      max(0.0, minlenratio - 0.1))
            for b, h in zip(empty, again):
                out[b] = h
        self.last_steps = steps
        return out

    def _search_once(self, enc_act: torch.Tensor, olens: List[int], maxlenratio: float,
                     minlenratio: float) -> List[List[Hypothesis]]:
        """Drives the search generator (`_search_run`): it enqueues its work without synchronising and yields the
        device `done` flags where the reference loop looks at the host (between step chunks); the answer sent back
        is "every utterance is done"."""
        L.require_gpu(enc_act, "enc_act")
        gen = self._search_run(enc_act, olens, maxlenratio, minlenratio)
        msg = None
        while True:
            try:
                req = gen.send(msg)
            except StopIteration as e:
                return e.value
            msg = bool(req.all().item())

    def _search_run(self, enc_act: torch.Tensor, olens: List[int], maxlenratio: float, minlenratio: float):
        """Generator: one search over a (sub-)batch on the CURRENT stream.  Yields the device `done` flags wherever
        the host decides whether to go on (send back True when every utterance is done); returns the n-best."""
        L.require_gpu(enc_act, "enc_act")
        lib = L.load()
        dev = enc_act.device
        B, T, d = enc_act.shape
        W, V = self.beam_size, self.n_vocab
        dec = self.scorers.get("decoder")
        ctc_sc = self.scorers.get("ctc")
        lm = self.scorers.get("lm")
        em_dtype = dec.em_dtype if dec is not None else (ctc_sc.ctc.em_dtype if ctc_sc is not None else lm.em_dtype)
        if lm is not None and lm.em_dtype != em_dtype:
            lm.compute_dtype = "bfloat16" if em_dtype == L.EM_BF16 else "float32"
            lm.invalidate()
        act = torch.bfloat16 if em_dtype == L.EM_BF16 else torch.float32
        enc_act = enc_act.to(act).contiguous()
        # length bounds per utterance (beam_search.py:414-429 with inp.shape[0] = olens[b])
        maxlens, minlens = [], []
        for tb in olens:
            if maxlenratio == 0:
                ml = tb
            elif maxlenratio < 0:
                ml = -1 * int(maxlenratio)
            else:
                ml = max(1, int(maxlenratio * tb))
            maxlens.append(ml)
            minlens.append(-1 * int(minlenratio) if minlenratio < 0 else int(minlenratio * tb))
        # Shape bucketing: buffers, strides and the captured hipGraph depend on (B, T, Lmax); the memory is
        # padded with zero frames to a multiple of 32 and the token capacity follows the bucket, so ragged
        # batches of a decode run (bin/asr_inference.py `inference`) keep hitting the same allocation and
        # graph.  Per-utterance xlens / maxlens bound every loop, so the padding changes no result.
        Tb = (T + 31) // 32 * 32
        if Tb != T:
            enc_act = torch.nn.functional.pad(enc_act, (0, 0, 0, Tb - T))
            T = Tb
        lcap = max(max(maxlens), T if maxlenratio == 0 else 0)
        Lmax = lcap + 2
        S = self.pre_beam_size if self.do_pre_beam else V
        NC = S + 1 if S < V else V
        cap = W * (lcap + 1)
        Tpad = T
        nl = dec.num_blocks if dec is not None else 0
        ff = dec.linear_units if dec is not None else 0
        bufs = self._alloc(dev, act, B, W, V, T, Tpad, NC, Lmax, cap, d, ff, nl, lm)
        bufs["xlens"].copy_(torch.tensor(olens, dtype=torch.int32))
        bufs["maxlens"].copy_(torch.tensor(maxlens, dtype=torch.int32))
        bufs["minlens"].copy_(torch.tensor(minlens, dtype=torch.int32))
        ctc_pk = ctc_sc.ctc._pack(dev) if ctc_sc is not None else None
        p = L.EmSearchParams(B=B, W=W, V=V, T=T, Tpad=Tpad, S=S, NC=NC, Lmax=Lmax, end_cap=cap,
                             sos=self.sos, eos=self.eos, blank=0,
                             use_end_detect=1 if maxlenratio == 0.0 else 0,
                             w_dec=float(self.weights.get("decoder", 0.0)) if dec is not None else 0.0,
                             w_ctc=float(self.weights.get("ctc", 0.0)) if ctc_sc is not None else 0.0,
                             w_len=float(self.weights.get("length_bonus", 0.0))
                             if "length_bonus" in self.scorers else 0.0,
                             w_lm=float(self.weights.get("lm", 0.0)) if lm is not None else 0.0)
        bs = L.EmSearchBuffers()
        for name in L.SEARCH_BUFFERS:
            setattr(bs, name, bufs[name].data_ptr() if name in bufs else None)
        if not self.use_hipgraph:
            bs.step = None
        lmw = lm.ensure_packed(dev, Lmax)["w"] if lm is not None else None
        if lmw is not None:
            bs.lm = C.addressof(lmw)
        dw = dec.ensure_packed(dev, Lmax)["w"] if dec is not None else None
        dwp = C.byref(dw) if dw is not None else None
        stream = L.current_stream_ptr()

        def init():
            L.check(lib.em_search_init(em_dtype, C.byref(p), dwp, C.byref(bs), L.ptr(enc_act), d,
                                       L.ptr(ctc_pk["w"]) if ctc_pk else None,
                                       L.ptr(ctc_pk["b"]) if ctc_pk else None,
                                       L.current_stream_ptr()), "em_search_init")

        def steps(i0, i1):
            L.check(lib.em_search_steps(em_dtype, C.byref(p), dwp, C.byref(bs), i0, i1,
                                        L.current_stream_ptr()), "em_search_steps")

        init()
        imax, K = max(maxlens), self.step_chunk
        if self.use_hipgraph:
            # One hipGraph = K search steps (~75 launches each).  Every step-dependent kernel reads the
            # step index from device memory, so the same graph is replayed ceil(imax / K) times; steps
            # past the end are no-ops.  The graph is tied to the buffer set and the parameter block.
            gkey = (id(bufs), bytes(p), em_dtype, id(dec._packed) if dec is not None else 0,
                    id(lm._packed) if lm is not None else 0)
            g = self._graphs.get(gkey)
            if g is None:
                steps(0, 1)  # warm-up outside capture (one-time attribute calls, lazy module loads)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    steps(0, K)
                self._graphs[gkey] = (g, p, bs, lmw)  # keep the argument blocks alive with the graph
                init()  # the warm-up step advanced the search state: start over
            else:
                g = g[0]
            i = 0
            while i < imax:
                g.replay()
                i += K
                if i < imax and (yield bufs["done"]):
                    break
            self.last_steps = min(i, imax)  # label steps enqueued by this search (bench.py's search roofline)
            return self._collect(bufs, B, W, maxlens)
        i = 0
        while i < imax:
            j = min(imax, i + K)
            steps(i, j)
            i = j
            if i < imax and (yield bufs["done"]):  # the only host sync of the search
                break
        self.last_steps = i
        return self._collect(bufs, B, W, maxlens)

    # ------------------------------------------------------------------ readout (host)
    def _collect(self, bufs, B, W, maxlens) -> List[List[Hypothesis]]:
        """One D2H copy of the ended lists + token tree, then a vectorised back-trace: all ended
        hypotheses walk their `parent` pointers together, one numpy gather per token position."""
        import numpy as np

        cpu = {k: bufs[k].cpu().numpy() for k in ("end_count", "end_pos", "end_slot", "end_forced",
                                                  "end_score", "end_sdec", "end_sctc", "end_slen",
                                                  "end_slm", "tok", "parent") if k in bufs}
        tok, parent = cpu["tok"], cpu["parent"]
        counts = cpu["end_count"]
        bi = np.repeat(np.arange(B), counts)
        ei = np.concatenate([np.arange(c) for c in counts]) if counts.sum() else np.zeros(0, dtype=np.int64)
        pos = cpu["end_pos"][bi, ei].astype(np.int64)
        cur = cpu["end_slot"][bi, ei].astype(np.int64)
        E = len(pos)
        ys = np.full((E, tok.shape[0] + 1), -1, dtype=np.int64)
        for j in range(int(pos.max()) if E else -1, -1, -1):
            m = pos >= j
            c = cur[m]
            ys[m, j] = tok[j, c]
            cur[m] = parent[j, c]
        forced = cpu["end_forced"][bi, ei].astype(bool)
        ys[np.arange(E)[forced], pos[forced] + 1] = self.eos
        lens = pos + 1 + forced
        keys = [k for k in ("decoder", "ctc", "length_bonus", "lm") if k in self.scorers]
        col = dict(decoder="end_sdec", ctc="end_sctc", length_bonus="end_slen", lm="end_slm")
        out, off = [], 0
        for b in range(B):
            hyps = []
            for e in range(int(counts[b])):
                k = off + e
                hyps.append(Hypothesis(yseq=torch.from_numpy(ys[k, : lens[k]].copy()),
                                       score=torch.tensor(cpu["end_score"][b, e]),
                                       scores={kk: torch.tensor(cpu[col[kk]][b, e]) for kk in keys}))
            off += int(counts[b])
            if self.normalize_length:  # beam_search.py:453-459
                hyps.sort(key=lambda h: float(h.score) / (len(h.yseq) - 1), reverse=True)
            else:
                hyps.sort(key=lambda h: float(h.score), reverse=True)
            if len(hyps) == 0:
                logger.warning("there is no N-best results, perform recognition again with smaller minlenratio.")
            out.append(hyps)
        return out

    # ------------------------------------------------------------------ reference signature
    def forward(self, x: torch.Tensor, maxlenratio: float = 0.0, minlenratio: float = 0.0,
                pre_x: torch.Tensor = None) -> List[Hypothesis]:
        """x (T, D) encoded speech of ONE utterance (beam_search.py:385-498)."""
        if pre_x is not None:
            raise NotImplementedError("sequential attention (pre_x)")
        L.require_gpu(x, "x")
        logger.info("decoder input length: " + str(x.shape[0]))
        nbest = self.search_batch(x.unsqueeze(0), [int(x.shape[0])], maxlenratio, minlenratio)[0]
        if len(nbest) == 0:  # search_batch has already backed minlenratio off (beam_search.py:462-474)
            return []
        best = nbest[0]
        for k, v in best.scores.items():
            logger.info(f"{float(v):6.2f} * {self.weights[k]:3} = {float(v) * self.weights[k]:6.2f} for {k}")
        logger.info(f"total log probability: {float(best.score):.2f}")
        logger.info(f"normalized log probability: {float(best.score) / len(best.yseq):.2f}")
        return nbest

    __call__ = forward


class SearchLanes:
    """Several joint searches IN FLIGHT, one per HIP stream (round 6).  A label step of the batched search is ~47 small
    launches whose grids cover a fraction of the chip (160 hypothesis rows at configs[2]) and whose time is launch latency and
    dependent round trips, not bandwidth: a second, independent search on another stream runs in the gaps of the first.
    Each lane owns a `BatchBeamSearch` (its own buffer set and hipGraphs - two searches must not share them) over the SAME
    scorers (weights are read-only), and is driven through the search generator (`_search_run`): the host enqueues a chunk of
    steps on a lane, moves on to the next lane, and comes back to look at the lane's `done` flags.  The host-side readout of a
    finished search (D2H + back-trace) overlaps the other lanes' steps as well.

        lanes = SearchLanes([build_beam_search(model, ...) for _ in range(2)], device)
        lanes.start(k, enc_act, olens, tag)      # on lane k's stream: enqueue the first chunk
        tag, nbest = lanes.poll(k)               # None while lane k is still searching
    `lanes.stream(k)` is the stream to run the lane's encoder call on."""

    def __init__(self, searches, device, streams=None, threaded=False):
        """`streams`: one HIP stream per lane (default: new ones).  HIP deals streams onto a handful of hardware queues and two
        streams on the same queue serialise - a caller that wants the lanes side by side probes for streams that are
        (bench.py `StepPipeline._pick`)."""
        import torch as _t

        self.searches = list(searches)
        self.streams = list(streams) if streams is not None else [_t.cuda.Stream(device=device) for _ in self.searches]
        assert len(self.streams) == len(self.searches)
        # threaded: a HOST THREAD per lane runs the lane's whole search (search_batch) on the lane's stream.  With one host
        # thread the lanes saturate at two: 249 label steps x 47 launches x ~2.5 us of hipLaunchKernel per search is the host's
        # launch rate, not the device (profiles/r06z_beam_lanes_ab.txt: 2 / 3 / 4 lanes 60.4 / 60.3 / 60.9 ms per batch).  ctypes
        # releases the GIL inside em_search_steps and torch inside its waits, so the lanes' launches proceed in parallel
        # (profiles/r06z_beam_threads_probe.txt: 2 / 3 / 6 threads 61.7 / 54.3 / 49.7 ms per batch of 16).
        self.threaded = bool(threaded)
        if self.threaded:
            for bs_ in self.searches:  # (a graph capture while another lane's thread launches is invalid: eager label steps)
                bs_.use_hipgraph = False
            import queue
            import threading

            self._jobs = [queue.SimpleQueue() for _ in self.searches]
            self._res = [None] * len(self.searches)
            self._done = threading.Event()
            self._threads = [threading.Thread(target=self._worker, args=(k,), daemon=True) for k in range(len(self.searches))]
            for t in self._threads:
                t.start()
        self.state = [None] * len(self.searches)
        self.ready = {}  # results finished while another lane was being waited for (`wait`)

    def _worker(self, k):
        while True:
            job = self._jobs[k].get()
            if job is None:
                return
            tag, args = job
            try:
                with torch.no_grad(), torch.cuda.stream(self.streams[k]):
                    out = self.searches[k].search_batch(*args)
                    torch.cuda.current_stream().synchronize()
                self._res[k] = (tag, out, None)
            except BaseException as e:  # handed to the thread that polls
                self._res[k] = (tag, None, e)
            self._done.set()

    def close(self):
        """Threaded lanes: end the worker threads (idle lanes only)."""
        if self.threaded:
            for q in self._jobs:
                q.put(None)
            for t in self._threads:
                t.join()
            self.threaded = False

    def __len__(self):
        return len(self.searches)

    def stream(self, k):
        return self.streams[k]

    def busy(self, k) -> bool:
        return self.state[k] is not None

    @torch.no_grad()
    def start(self, k, enc_act, olens, tag=None, maxlenratio: float = 0.0, minlenratio: float = 0.0):
        assert self.state[k] is None, "lane is busy"
        if self.threaded:  # (enc_act was produced on this lane's stream or is complete: the worker's launches follow it there)
            self.state[k] = dict(tag=tag)
            self._res[k] = None
            self._jobs[k].put((tag, (enc_act, olens, maxlenratio, minlenratio)))
            return
        with torch.cuda.stream(self.streams[k]):
            gen = self.searches[k]._search_run(enc_act, olens, maxlenratio, minlenratio)
            try:
                req = gen.send(None)
                self.state[k] = dict(gen=gen, req=req, out=None, tag=tag, args=(enc_act, olens, maxlenratio, minlenratio))
            except StopIteration as e:  # (a search shorter than one chunk of steps)
                self.state[k] = dict(gen=None, req=None, out=e.value, tag=tag, args=(enc_act, olens, maxlenratio, minlenratio))

    @torch.no_grad()
    def poll(self, k):
        """Look at lane k's `done` flags (waits for the chunk in flight on ITS stream only) and enqueue its next chunk;
        (tag, n-best lists) once the search has ended, else None."""
        st = self.state[k]
        assert st is not None, "lane is idle"
        if self.threaded:
            r = self._res[k]
            if r is None:  # (no spinning on the GIL the workers need: sleep until some lane reports, a millisecond at most)
                self._done.wait(0.001)
                self._done.clear()
                r = self._res[k]
                if r is None:
                    return None
            self._res[k], self.state[k] = None, None
            if r[2] is not None:
                raise r[2]
            return r[0], r[1]
        with torch.cuda.stream(self.streams[k]):
            if st["out"] is None:
                msg = bool(st["req"].all().item())
                try:
                    st["req"] = st["gen"].send(msg)
                    return None
                except StopIteration as e:
                    st["out"] = e.value
            out = st["out"]
            enc_act, olens, maxr, minr = st["args"]
            empty = [b for b, h in enumerate(out) if len(h) == 0]
            if empty and minr >= 0.1:  # the reference's back-off for an utterance that ended no hypothesis (search_batch)
                idx = torch.tensor(empty, device=enc_act.device)
                again = self.searches[k].search_batch(enc_act.index_select(0, idx), [olens[b] for b in empty], maxr,
                                                      max(0.0, minr - 0.1))
                for b, h in zip(empty, again):
                    out[b] = h
        self.state[k] = None
        return st["tag"], out

    def free_lane(self):
        """Index of an idle lane, or None."""
        for k, st in enumerate(self.state):
            if st is None and k not in self.ready:
                return k
        return None

    def wait(self, k):
        """(tag, n-best lists) of lane k's search.  While waiting, the OTHER busy lanes are polled as well - a lane that is not
        polled stops after the chunk of steps it has enqueued - and what they finish is kept for their own `wait`."""
        while k not in self.ready:
            for j in range(len(self.state)):
                if self.state[j] is not None and j not in self.ready:
                    r = self.poll(j)
                    if r is not None:
                        self.ready[j] = r
        return self.ready.pop(k)


def build_beam_search(asr_model, beam_size: int, ctc_weight: float, penalty: float,
                      lm_weight: float = 0.0, token_list=None, normalize_length: bool = False, lm=None):
    """Scorer / weight set-up of Speech2Text (espnet2/bin/asr_inference.py:168-176, 310-316,
    353-381)."""
    scorers = dict(decoder=asr_model.decoder,
                   ctc=CTCPrefixScorer(ctc=asr_model.ctc, eos=asr_model.eos) if asr_model.ctc is not None else None,
                   length_bonus=LengthBonus(len(token_list)))
    if lm is not None:
        scorers["lm"] = lm  # asr_inference.py:179-191 (scorers["lm"] = lm.lm)
    weights = dict(decoder=1.0 - ctc_weight, ctc=ctc_weight, lm=lm_weight, length_bonus=penalty)
    return BatchBeamSearch(beam_size=beam_size, weights=weights, scorers=scorers, sos=asr_model.sos,
                           eos=asr_model.eos, vocab_size=len(token_list), token_list=token_list,
                           pre_beam_score_key=None if ctc_weight == 1.0 else "full",
                           normalize_length=normalize_length)
