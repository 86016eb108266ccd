"""BatchBeamSearchOnline: block-synchronous streaming beam search on the MI355X.

Mirrors espnet2/legacy/nets/batch_beam_search_online.py:22-534 (Tsunoo et al., arXiv:2006.14941) for
the configuration `Speech2TextStreaming` builds (espnet2/bin/asr_inference_streaming.py:122-136:
block 40 / hop 16 / look-ahead 16 regardless of the encoder's own block sizes, blockwise branch,
`time_sync=False`, `incremental_decode=False`, no length limits).

Split of the work.  The reference's control flow is data dependent step by step (repetition detection,
hypotheses reaching <eos> inside a block, rewinding by one step, end detection on the ended list), so
the HOST mirrors it line by line — `forward` :155-376, `process_one_block` :394-493 including the
reference's quirks (the second `post_process` at `maxlen - 1`, the ended hypotheses of the final block
appended twice) — while every number is produced on the DEVICE through the C-ABI:

  em_search_init / em_search_online_extend   init_hyp + extend (:518-534): decoder memory K/V of the
                                              visible frames, CTC log-probs, forward variables of the
                                              running rows continued over the new frames (Eq. 14)
  em_search_online_core                      best = search(running_hyps, h)
  em_search_online_commit                    prev_hyps = running_hyps; running_hyps = post_process(best)
  em_search_online_rewind                    running_hyps = prev_hyps

One 32-byte record per beam row comes back per step (parent slot, token, total and per-scorer scores);
token sequences and the ended list live on the host, exactly like the reference's Hypothesis objects.
There is no CPU path for the scores: without the HIP library the calls raise.
"""
import ctypes as C
import os
from typing import List, Optional

import torch

from espnet_amd import lib as L
from espnet_amd.nets.batch_beam_search import BatchBeamSearch
from espnet_amd.nets.beam_search import Hypothesis
from espnet_amd.nets.e2e_asr_common import end_detect


class _Rows:
    """Host view of a BatchHypothesis: per device slot k the token list and the scores."""

    def __init__(self, slots, yseq, score, scores):
        self.slots, self.yseq, self.score, self.scores = slots, yseq, score, scores

    def __len__(self):
        return len(self.slots)

    def hyp(self, j, keys) -> Hypothesis:
        return Hypothesis(yseq=torch.tensor(self.yseq[j], dtype=torch.long), score=torch.tensor(self.score[j]),
                          scores={k: torch.tensor(self.scores[j][k]) for k in keys})


class BatchBeamSearchOnline(BatchBeamSearch):
    def __init__(self, *args, block_size=40, hop_size=16, look_ahead=16, disable_repetition_detection=False,
                 encoded_feat_length_limit=0, decoder_text_length_limit=0, incremental_decode=False,
                 time_sync=False, ctc=None, hold_n=0, transducer_conf=None, joint_network=None,
                 max_frames: int = 1536, **kwargs):
        super().__init__(*args, **kwargs)
        if (encoded_feat_length_limit or decoder_text_length_limit or incremental_decode or time_sync or hold_n
                or transducer_conf is not None or joint_network is not None or block_size == 0):
            raise NotImplementedError("BatchBeamSearchOnline: only the blockwise label-synchronous mode that "
                                      "Speech2TextStreaming configures is on the MI355X path")
        self.block_size, self.hop_size, self.look_ahead = block_size, hop_size, look_ahead
        self.disable_repetition_detection = disable_repetition_detection
        self.max_frames = max_frames  # frame capacity of one stream (source attention keeps T scores in LDS)
        # The ~50 launches of a label step can be replayed as one captured hipGraph: the step index lives in device memory
        # (written by the host before each replay - a block may rewind by one), the number of visible frames is baked
        # into the launches, so there is one graph per (buffer set, T); the block schedule is the same for every
        # utterance of a stream slot, so the keys repeat, and a graph is captured the third time its key is used.
        # Bit-identical to the eager sequence (tests/test_gpu_online_search.py::test_online_search_graph_replay_equals_eager)
        # and NOT faster: 131.8 ms per 10 s utterance against 125.3 eager (beam 10, 279 label steps, profiles/r03z) -
        # the step is bound by its ~50 dependent kernels on the GPU, and the eager launches of a step already run ahead
        # of them.  Off unless ESPNET_AMD_ONLINE_GRAPH=1.
        self.use_hipgraph = os.environ.get("ESPNET_AMD_ONLINE_GRAPH", "0") == "1"
        self.graph_after = 2  # eager uses of a (buffers, T) key before it is captured
        self._graph_uses = {}
        self.n_replays = 0  # label steps served by a graph replay (diagnostics / tests)
        self.events: List[str] = []  # log markers of the reference, kept for tests / diagnostics
        self.n_steps = 0  # label steps computed on the device since construction (diagnostics)
        self.reset()

    def reset(self):
        self.encbuffer: Optional[torch.Tensor] = None  # (max_frames, d) act dtype on the device
        self.n_enc = 0
        self.running: Optional[_Rows] = None
        self.prev_hyps: Optional[_Rows] = None
        self.ended_hyps: List[Hypothesis] = []
        self.processed_block = 0
        self.process_idx = 0
        self.prev_output = None
        self._visible = 0  # frames the device state has been extended to
        self._dev = None

    # ------------------------------------------------------------------ device plumbing
    def _keys(self):
        return [k for k in ("decoder", "ctc", "length_bonus", "lm") if k in self.scorers]

    def _setup(self, x: torch.Tensor):
        L.require_gpu(x, "x")  # no CPU path: the scores come from the device
        dev, d = x.device, x.size(-1)
        dec, ctc_sc, lm = self.scorers.get("decoder"), self.scorers.get("ctc"), self.scorers.get("lm")
        em_dtype = dec.em_dtype if dec is not None else (ctc_sc.ctc.em_dtype if ctc_sc is not None else lm.em_dtype)
        if lm is not None and lm.em_dtype != em_dtype:
            lm.compute_dtype = "bfloat16" if em_dtype == L.EM_BF16 else "float32"
            lm.invalidate()
        act = torch.bfloat16 if em_dtype == L.EM_BF16 else torch.float32
        W, V, Tcap = self.beam_size, self.n_vocab, self.max_frames
        Lmax = Tcap + 2
        S = self.pre_beam_size if self.do_pre_beam else V
        NC = S + 1 if S < V else V
        bufs = self._alloc(dev, act, 1, W, V, Tcap, (Tcap + 31) // 32 * 32, NC, Lmax, W, d,
                           dec.linear_units if dec is not None else 0, dec.num_blocks if dec is not None else 0,
                           lm, online=True)
        bufs["maxlens"].fill_(Lmax + 16)  # no forced <eos> on the device: the host owns the length logic
        bufs["minlens"].zero_()
        bs = L.EmSearchBuffers()
        for name in L.SEARCH_BUFFERS:
            setattr(bs, name, bufs[name].data_ptr() if name in bufs else None)
        bs.step = bufs["step"].data_ptr() if self.use_hipgraph else None  # graph mode: step index in device memory
        lmw = lm.ensure_packed(dev, Lmax)["w"] if lm is not None else None
        if lmw is not None:
            bs.lm = C.addressof(lmw)
        dw = dec.ensure_packed(dev, Lmax)["w"] if dec is not None else None
        self._dev = dict(dev=dev, d=d, em_dtype=em_dtype, act=act, bufs=bufs, bs=bs, lmw=lmw, dw=dw,
                         dwp=C.byref(dw) if dw is not None else None, S=S, NC=NC, Lmax=Lmax, Tcap=Tcap,
                         ctc_pk=ctc_sc.ctc._pack(dev) if ctc_sc is not None else None,
                         best_host=torch.empty(W, 8, dtype=torch.float32).pin_memory(),
                         step_host=torch.zeros(2, dtype=torch.int32).pin_memory(),
                         gkey=(id(bufs), em_dtype, id(dec._packed) if dec is not None else 0,
                               id(lm._packed) if lm is not None else 0))
        self.encbuffer = torch.empty(Tcap, d, dtype=act, device=dev)

    def _params(self, T: int) -> L.EmSearchParams:
        D = self._dev
        return L.EmSearchParams(
            B=1, W=self.beam_size, V=self.n_vocab, T=T, Tpad=(T + 31) // 32 * 32, S=D["S"], NC=D["NC"],
            Lmax=D["Lmax"], end_cap=self.beam_size, sos=self.sos, eos=self.eos, blank=0, use_end_detect=0,
            w_dec=float(self.weights.get("decoder", 0.0)) if "decoder" in self.scorers else 0.0,
            w_ctc=float(self.weights.get("ctc", 0.0)) if "ctc" in self.scorers else 0.0,
            w_len=float(self.weights.get("length_bonus", 0.0)) if "length_bonus" in self.scorers else 0.0,
            w_lm=float(self.weights.get("lm", 0.0)) if "lm" in self.scorers else 0.0, ldT=D["Tcap"])

    def _see(self, T: int):
        """init_hyp (first block) / extend (:518-534): the device state now covers encbuffer[:T]."""
        D, lib = self._dev, L.load()
        self._p = self._params(T)
        D["bufs"]["xlens"].fill_(T)
        D["bufs"]["mem_vT"].zero_()  # the padded tail of V^T must be zero under the new Tpad stride
        ctc_w = L.ptr(D["ctc_pk"]["w"]) if D["ctc_pk"] else None
        ctc_b = L.ptr(D["ctc_pk"]["b"]) if D["ctc_pk"] else None
        if self._visible == 0:
            L.check(lib.em_search_init(D["em_dtype"], C.byref(self._p), D["dwp"], C.byref(D["bs"]),
                                       L.ptr(self.encbuffer), D["d"], ctc_w, ctc_b, L.current_stream_ptr()),
                    "em_search_init")
            D["bufs"]["maxlens"].fill_(D["Lmax"] + 16)
        elif T != self._visible:
            L.check(lib.em_search_online_extend(D["em_dtype"], C.byref(self._p), D["dwp"], C.byref(D["bs"]),
                                                L.ptr(self.encbuffer), D["d"], ctc_w, ctc_b, self.process_idx,
                                                self._visible, L.current_stream_ptr()), "em_search_online_extend")
        self._visible = T

    def _search(self) -> _Rows:
        """best = self.search(self.running_hyps, h) (:400)."""
        D, lib = self._dev, L.load()
        self.n_steps += 1

        def core():
            L.check(lib.em_search_online_core(D["em_dtype"], C.byref(self._p), D["dwp"], C.byref(D["bs"]),
                                              self.process_idx, L.current_stream_ptr()), "em_search_online_core")

        if self.use_hipgraph:
            # (the previous step's host sync lies between this write and the copy that read the buffer last)
            D["step_host"][0] = self.process_idx
            D["bufs"]["step"].copy_(D["step_host"], non_blocking=True)
            gkey = D["gkey"] + (bytes(self._p),)
            ent = self._graphs.get(gkey)
            if ent is not None:
                ent[0].replay()
                self.n_replays += 1
            else:
                core()
                uses = self._graph_uses.get(gkey, 0) + 1
                self._graph_uses[gkey] = uses
                if uses > self.graph_after:
                    torch.cuda.current_stream().synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):  # (captured, not executed: the eager call above did this step)
                        core()
                    self._graphs[gkey] = (g, self._p, D["bs"], D["lmw"], D["dw"])  # argument blocks kept alive
                    del self._graph_uses[gkey]
        else:
            core()
        D["best_host"].copy_(D["bufs"]["online_best"], non_blocking=True)
        torch.cuda.current_stream().synchronize()  # the one host sync of a step
        rec = D["best_host"].tolist()
        keys = self._keys()
        col = dict(decoder=4, ctc=5, length_bonus=6, lm=7)
        by_slot = {s: j for j, s in enumerate(self.running.slots)}
        slots, yseq, score, scores = [], [], [], []
        for k, r in enumerate(rec):
            if r[0] == 0.0:
                continue
            slots.append(k)
            yseq.append(self.running.yseq[by_slot[int(r[1])]] + [int(r[2])])
            score.append(r[3])
            scores.append({kk: r[col[kk]] for kk in keys})
        return _Rows(slots, yseq, score, scores)

    def _commit(self):
        """Device side of `prev_hyps = running_hyps; running_hyps = post_process(best)` (:459-462)."""
        D = self._dev
        L.check(L.load().em_search_online_commit(D["em_dtype"], C.byref(self._p), C.byref(D["bs"]), self.process_idx,
                                                 L.current_stream_ptr()), "em_search_online_commit")

    def _rewind(self):
        """Device side of `running_hyps = prev_hyps` (:484-487)."""
        D = self._dev
        L.check(L.load().em_search_online_rewind(C.byref(self._p), C.byref(D["bs"]), L.current_stream_ptr()),
                "em_search_online_rewind")

    def _post_process(self, i, maxlen, minlen, best: _Rows, ended: List[Hypothesis]) -> _Rows:
        """BatchBeamSearch.post_process (batch_beam_search.py:359-423), in place on `best`."""
        keys = self._keys()
        if i == maxlen - 1:
            self.events.append("adding <eos> in the last position")
            for y in best.yseq:
                y.append(self.eos)
        keep = []
        for j in range(len(best)):
            if best.yseq[j][-1] == self.eos:
                if i >= minlen:
                    ended.append(best.hyp(j, keys))
            else:
                keep.append(j)
        return _Rows([best.slots[j] for j in keep], [best.yseq[j] for j in keep], [best.score[j] for j in keep],
                     [best.scores[j] for j in keep])

    # ------------------------------------------------------------------ reference control flow
    def process_one_block(self, h_len, is_final, maxlen, minlen, maxlenratio):
        """:394-493."""
        self._see(h_len)
        local_ended_hyps = None
        keys = self._keys()
        while self.process_idx < maxlen:
            if len(self.running) == 0:
                raise RuntimeError("no running hypothesis left in a non-final block (the beam reached maxlen = "
                                   "number of encoder frames before the utterance ended)")
            best = self._search()
            forced = self.process_idx == maxlen - 1
            if forced:
                self.running = self._post_process(self.process_idx, maxlen, minlen, best, self.ended_hyps)
            local_ended_hyps = []
            prev_repeat = False
            for j in range(len(best)):
                y = best.yseq[j]
                if y[-1] == self.eos:
                    local_ended_hyps.append(best.hyp(j, keys))
                elif (not self.disable_repetition_detection and not prev_repeat and y[-1] in y[:-1]
                      and not is_final):
                    prev_repeat = True
            if prev_repeat:
                self.events.append("Detected repetition")
                break
            if (is_final and maxlenratio == 0.0
                    and end_detect([lh.asdict() for lh in self.ended_hyps], self.process_idx)):
                self.events.append("end detected at")
                return self.assemble_hyps(self.ended_hyps)
            if len(local_ended_hyps) > 0 and not is_final:
                self.events.append("reaching EOS in this block")
                break
            self.prev_hyps = self.running
            self.running = self._post_process(self.process_idx, maxlen, minlen, best, self.ended_hyps)
            if not forced:  # at maxlen - 1 every row ended: nothing runs on
                self._commit()
            if is_final:
                self.ended_hyps.extend(local_ended_hyps)
            if len(self.running) == 0:
                self.events.append("no hypothesis. Finish")
                return self.assemble_hyps(self.ended_hyps)
            self.process_idx += 1
        if is_final:
            return self.assemble_hyps(self.ended_hyps)
        rets = self.assemble_hyps((local_ended_hyps or []) + self.ended_hyps)
        if self.process_idx > 1 and self.prev_hyps is not None and len(self.prev_hyps) > 0:
            self._rewind()
            self.running = self.prev_hyps
            self.process_idx -= 1
            self.prev_hyps = None
        return rets

    def assemble_hyps(self, ended_hyps):
        """:495-516 (sorted() is stable: equal scores keep their insertion order)."""
        if self.normalize_length:
            return sorted(ended_hyps, key=lambda h: float(h.score) / (len(h.yseq) - 1), reverse=True)
        return sorted(ended_hyps, key=lambda h: float(h.score), reverse=True)

    @torch.no_grad()
    def forward(self, x: torch.Tensor, maxlenratio: float = 0.0, minlenratio: float = 0.0,
                is_final: bool = True) -> List[Hypothesis]:
        """x (T_new, d): the encoder frames of this chunk ON THE GPU (may be empty).  :155-376."""
        if self._dev is None:
            self._setup(x)
        n_new = x.size(0)
        if self.n_enc + n_new > self.max_frames:
            raise RuntimeError(f"stream longer than max_frames={self.max_frames} encoder frames")
        if n_new:
            self.encbuffer[self.n_enc : self.n_enc + n_new].copy_(x)
            self.n_enc += n_new
        T = self.n_enc
        maxlen = T if maxlenratio == 0 else max(1, int(maxlenratio * T))
        minlen = -1 * int(minlenratio) if minlenratio < 0 else int(minlenratio * T)
        ret = None
        while True:
            cur_end_frame = self.block_size - self.look_ahead + self.hop_size * self.processed_block
            if cur_end_frame < T:
                h_len, block_is_final = cur_end_frame, False
            elif is_final:
                h_len, block_is_final = T, True
            else:
                break
            if h_len < 1:
                raise RuntimeError("BatchBeamSearchOnline got a final block without encoder frames")
            if self.running is None:  # init_hyp (:331-332)
                self.running = _Rows([0], [[self.sos]], [0.0], [{k: 0.0 for k in self._keys()}])
            ret = self.process_one_block(h_len, block_is_final, maxlen, minlen, maxlenratio)
            self.processed_block += 1
            if block_is_final:
                return ret
        if ret is None:
            return [] if self.prev_output is None else self.prev_output
        self.prev_output = ret
        return ret

    __call__ = forward
