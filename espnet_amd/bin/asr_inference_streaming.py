"""Speech2TextStreaming: the streaming object API of the drop-in (BASELINE config 5).

Mirrors espnet2/bin/asr_inference_streaming.py:36-365: `apply_frontend(speech, prev_states,
is_final)` (waveform overlap buffer + edge-frame trimming, :205-293, reproduced on the host with
the features coming from the HIP frontend), `__call__(speech, is_final)` feeding
`ContextualBlockConformerEncoder.forward_infer` chunk by chunk (:316-322), `reset()`.

Decoding (`search=`):
  * "online" (default when beam_size > 1; what the reference does, :122-136, :316-330): every chunk's
    encoder frames go to `BatchBeamSearchOnline` — block-synchronous joint CTC/attention search with CTC
    `extend_prob` / `extend_state` (espnet_amd/nets/batch_beam_search_online.py).  Like the reference, a
    non-final call returns hypotheses only when some reached <eos> inside a block.
  * "greedy" (default when beam_size == 1; BASELINE config 5's per-chunk step): incremental greedy CTC
    (G1: per-frame argmax + collapse carried across chunk seams, bin/asr_inference.py:574-575), a
    partial transcript after every chunk.
  * "offline": greedy partials, and the offline joint CTC/attention beam search over the accumulated
    encoder output at `is_final`.
"""
import contextlib
import logging
import math
from pathlib import Path
from typing import List, Optional, Union

import numpy as np
import torch

from espnet_amd import lib as L
from espnet_amd.asr.encoder.contextual_block_conformer_encoder import (
    ContextualBlockConformerEncoder, StreamingStepGraph)
from espnet_amd.nets.beam_search import Hypothesis
from espnet_amd.tasks.asr import ASRTask
from espnet_amd.text.token_id_converter import TokenIDConverter, build_tokenizer

logger = logging.getLogger(__name__)


class Speech2TextStreaming:
    def __init__(self, asr_train_config: Union[Path, str], asr_model_file: Union[Path, str, None] = None,
                 lm_train_config=None, lm_file=None, token_type: Optional[str] = None,
                 bpemodel: Optional[str] = None, device: str = "cuda", maxlenratio: float = 0.0,
                 minlenratio: float = 0.0, batch_size: int = 1, dtype: str = "float32",
                 beam_size: int = 20, ctc_weight: float = 0.5, lm_weight: float = 1.0,
                 penalty: float = 0.0, nbest: int = 1, disable_repetition_detection: bool = False,
                 decoder_text_length_limit: int = 0, encoded_feat_length_limit: int = 0,
                 normalize_length: bool = False, use_hipgraph: bool = True, search: Optional[str] = None):
        if search is None:
            search = "online" if beam_size > 1 else "greedy"
        if search not in ("online", "greedy", "offline"):
            raise ValueError(f"search={search!r}")
        if decoder_text_length_limit or encoded_feat_length_limit:
            raise NotImplementedError("decoder_text_length_limit / encoded_feat_length_limit")
        if not str(device).startswith("cuda"):
            raise RuntimeError("espnet_amd runs on an MI355X only (device='cuda'); no CPU fallback")
        assert batch_size == 1
        from espnet_amd.bin.asr_inference import resolve_dtype

        dtype = resolve_dtype(dtype)
        asr_model, args = ASRTask.build_model_from_file(asr_train_config, asr_model_file, device,
                                                        compute_dtype=dtype)
        if not isinstance(asr_model.encoder, ContextualBlockConformerEncoder):
            raise NotImplementedError("Speech2TextStreaming needs encoder: contextual_block_conformer")
        self.asr_model, self.asr_train_args = asr_model, args
        self.device, self.dtype = device, dtype
        self.maxlenratio, self.minlenratio, self.nbest = maxlenratio, minlenratio, nbest
        self.beam_size, self.search = beam_size, search
        self.beam_search = None
        lm = None
        if lm_train_config is not None:  # :97-102
            from espnet_amd.tasks.lm import LMTask

            lm = LMTask.build_model_from_file(lm_train_config, lm_file, device, compute_dtype=dtype)[0].lm
        if search == "offline":
            from espnet_amd.nets.batch_beam_search import build_beam_search

            self.beam_search = build_beam_search(asr_model, beam_size=beam_size, ctc_weight=ctc_weight,
                                                 penalty=penalty, token_list=asr_model.token_list,
                                                 lm_weight=lm_weight if lm is not None else 0.0, lm=lm,
                                                 normalize_length=normalize_length)
        elif search == "online":
            from espnet_amd.nets.batch_beam_search_online import BatchBeamSearchOnline
            from espnet_amd.nets.scorers.ctc import CTCPrefixScorer
            from espnet_amd.nets.scorers.length_bonus import LengthBonus

            token_list = asr_model.token_list
            # :88-136 (the search always runs with block 40 / hop 16 / look-ahead 16: the reference
            # leaves its encoder_conf read-out commented and BatchBeamSearchOnline's defaults apply)
            scorers = dict(decoder=asr_model.decoder,
                           ctc=CTCPrefixScorer(ctc=asr_model.ctc, eos=asr_model.eos) if asr_model.ctc else None,
                           length_bonus=LengthBonus(len(token_list)))
            if lm is not None:
                scorers["lm"] = lm
            weights = dict(decoder=1.0 - ctc_weight, ctc=ctc_weight, lm=lm_weight if lm is not None else 0.0,
                           length_bonus=penalty)
            self.beam_search = BatchBeamSearchOnline(
                beam_size=beam_size, weights=weights, scorers=scorers, sos=asr_model.sos, eos=asr_model.eos,
                vocab_size=len(token_list), token_list=token_list,
                pre_beam_score_key=None if ctc_weight == 1.0 else "full", normalize_length=normalize_length,
                disable_repetition_detection=disable_repetition_detection)
        token_type = token_type if token_type is not None else getattr(args, "token_type", None)
        bpemodel = bpemodel if bpemodel is not None else getattr(args, "bpemodel", None)
        self.tokenizer = (None if token_type is None or (token_type == "bpe" and bpemodel is None)
                          else build_tokenizer(token_type=token_type, bpemodel=bpemodel))
        self.converter = TokenIDConverter(token_list=asr_model.token_list)
        fconf = getattr(args, "frontend_conf", None) or {}
        self.n_fft = fconf.get("n_fft", 512)  # :123-137
        self.hop_length = fconf.get("hop_length", 128)
        self.win_length = fconf["win_length"] if fconf.get("win_length") is not None else self.n_fft
        self.use_hipgraph = use_hipgraph
        self._runner = None
        self._flens_cache = {}  # (streams, samples, device) -> frame counts on the device (apply_frontend_batch)
        self._pin_ring = {}  # chunk length -> two pinned staging rows + a turn counter (apply_frontend)
        self._tick_graphs = {}  # (group, chunk shape) -> BatchTickGraph of finished utterances (batch_call_async)
        self.reset()

    def reset(self):
        self.frontend_states = None
        self.encoder_states = None
        self._batches = {}  # group -> state of one set of lock-step streams (batch_call / batch_call_async)
        self._enc_chunks = []
        self._last_id = -1
        self._partial_ids: List[int] = []
        if self._runner is not None:
            self._runner.reset()
        if self.search == "online" and self.beam_search is not None:
            self.beam_search.reset()

    # ------------------------------------------------------------------ frontend (:205-293)
    def apply_frontend(self, speech: torch.Tensor, prev_states=None, is_final: bool = False):
        if prev_states is not None:
            speech = torch.cat([prev_states["waveform_buffer"], speech], dim=0)
        if speech.size(0) <= self.win_length:
            if is_final:
                speech = torch.cat([speech, torch.zeros(self.win_length - speech.size(0), dtype=speech.dtype)], dim=0)
            else:
                return None, None, {"waveform_buffer": speech.clone()}
        edge = math.ceil(math.ceil(self.win_length / self.hop_length) / 2)
        if is_final:
            speech_to_process, waveform_buffer = speech, None
        else:
            n_frames = speech.size(0) // self.hop_length
            n_residual = speech.size(0) % self.hop_length
            speech_to_process = speech.narrow(0, 0, n_frames * self.hop_length)
            keep = (edge * 2 - 1) * self.hop_length + n_residual
            waveform_buffer = speech.narrow(0, speech.size(0) - keep, keep).clone()
        # (the chunk through one of two pinned staging rows and an asynchronous copy, the frame count from a per-length cache:
        # a pageable host -> device copy is synchronous, and there were two of them in front of every call's first launch)
        n = speech_to_process.size(0)
        if speech_to_process.is_cuda or torch.device(self.device).type != "cuda":
            wav = speech_to_process.unsqueeze(0).to(torch.float32).to(self.device)
        else:
            pins = self.__dict__.setdefault("_pin_ring", {})  # (objects assembled without __init__ in the tests)
            ring = pins.get(n)
            if ring is None:
                if len(pins) > 16:
                    pins.clear()
                ring = pins[n] = [torch.empty(1, n, dtype=torch.float32).pin_memory() for _ in range(2)] + [0]
            buf = ring[ring[2] & 1]
            ring[2] += 1
            buf[0].copy_(speech_to_process)
            wav = buf.to(self.device, non_blocking=True)
        m = self.asr_model
        key = (1, n, wav.device)
        fcache = self.__dict__.setdefault("_flens_cache", {})
        flens_dev = fcache.get(key)
        if flens_dev is None:
            if len(fcache) > 64:
                fcache.clear()
            flens_dev = fcache[key] = torch.tensor(m.frontend.feature_lengths([n]), dtype=torch.int32).to(wav.device)
            if wav.is_cuda:
                torch.cuda.current_stream().synchronize()
        feats = m.frontend.forward_device(wav, flens_dev)  # espnet_model.py:450-467
        if m.normalize is not None:
            feats = m.normalize.forward_device(feats, flens_dev)
        # trimming of the frames that see the artificial chunk edges (:261-286)
        if is_final:
            if prev_states is not None:
                feats = feats.narrow(1, edge, feats.size(1) - edge)
        elif prev_states is None:
            feats = feats.narrow(1, 0, feats.size(1) - edge)
        else:
            feats = feats.narrow(1, edge, feats.size(1) - 2 * edge)
        feats_lengths = torch.full([1], feats.size(1), dtype=torch.long)
        return feats, feats_lengths, (None if is_final else {"waveform_buffer": waveform_buffer})

    # ------------------------------------------------------------------ one chunk (:295-336)
    @torch.no_grad()
    def __call__(self, speech: Union[torch.Tensor, np.ndarray], is_final: bool = True):
        if isinstance(speech, np.ndarray):
            speech = torch.tensor(speech)
        feats, feats_lengths, self.frontend_states = self.apply_frontend(
            speech, self.frontend_states, is_final=is_final)
        ret = []
        if feats is not None:
            enc = self._encode_chunk(feats[0].contiguous(), is_final)
            if self.search == "online":  # :323-330
                nbest_hyps = self.beam_search(x=enc, maxlenratio=self.maxlenratio, minlenratio=self.minlenratio,
                                              is_final=is_final)
                ret = self.assemble_hyps(nbest_hyps)
            else:
                if enc.size(0) > 0:
                    self._enc_chunks.append(enc.clone())
                    self._extend_partial(enc)
                ret = self._results(is_final)
        if is_final:
            self.reset()
        return ret

    # ------------------------------------------------------------------ a batch of lock-step streams (greedy CTC)
    def apply_frontend_batch(self, speech: torch.Tensor, prev_states=None, is_final: bool = False):
        """`apply_frontend` (:205-293) for S streams that are fed equal chunks at the same times: speech (S, n) host or
        device tensor.  Returns (feats (S, t, n_mels) on the device or None, state)."""
        # the chunk goes to the device first (one copy, asynchronous from pinned memory); the carried buffer lives there
        speech = speech.to(self.device, dtype=torch.float32, non_blocking=True)
        if prev_states is not None:
            speech = torch.cat([prev_states["waveform_buffer"], speech], dim=1)
        n_all = speech.size(1)
        if n_all <= self.win_length:
            if is_final:
                speech = torch.cat([speech, speech.new_zeros(speech.size(0), self.win_length - n_all)], dim=1)
            else:
                return None, {"waveform_buffer": speech.clone()}
        edge = math.ceil(math.ceil(self.win_length / self.hop_length) / 2)
        if is_final:
            to_process, waveform_buffer = speech, None
        else:
            n_frames = speech.size(1) // self.hop_length
            n_residual = speech.size(1) % self.hop_length
            to_process = speech.narrow(1, 0, n_frames * self.hop_length)
            keep = (edge * 2 - 1) * self.hop_length + n_residual
            waveform_buffer = speech.narrow(1, speech.size(1) - keep, keep).clone()
        wav = to_process.contiguous()
        S, n = wav.shape
        m = self.asr_model
        # (the frame counts of a tick are the same tick after tick: their device copy is made once per (S, n), not with a
        # pageable host -> device copy in front of every tick's launches)
        key = (S, n, wav.device)
        flens_dev = self._flens_cache.get(key)
        if flens_dev is None:
            if len(self._flens_cache) > 64:
                self._flens_cache.clear()
            flens = m.frontend.feature_lengths([n] * S)
            flens_dev = self._flens_cache[key] = torch.tensor(flens, dtype=torch.int32).to(wav.device)
            torch.cuda.current_stream().synchronize()  # (other streams' ticks read it too: batch_call_async)
        feats = m.frontend.forward_device(wav, flens_dev)
        if m.normalize is not None:
            feats = m.normalize.forward_device(feats, flens_dev)
        if is_final:
            if prev_states is not None:
                feats = feats.narrow(1, edge, feats.size(1) - edge)
        elif prev_states is None:
            feats = feats.narrow(1, 0, feats.size(1) - edge)
        else:
            feats = feats.narrow(1, edge, feats.size(1) - 2 * edge)
        return feats, (None if is_final else {"waveform_buffer": waveform_buffer})

    @torch.no_grad()
    def batch_call(self, speech: torch.Tensor, is_final: bool = False) -> List[List[int]]:
        """One chunk of S lock-step streams, greedy CTC (what a server that batches its live connections calls per tick):
        speech (S, n) f32.  Returns, per stream, the token ids decoded so far (incremental G1: per-frame arg-max, repeats
        collapsed across chunk seams, blank / <sos/eos> dropped).  Stream s sees exactly what `__call__` gives it alone
        (tests/test_gpu_streaming.py::test_batch_call_equals_single_streams); one launch sequence serves all S streams
        (ContextualBlockConformerEncoder.forward_infer_batch).  The object holds the batch's state until is_final."""
        return self.batch_call_async(speech, is_final).result()

    @torch.no_grad()
    def batch_call_async(self, speech: torch.Tensor, is_final: bool = False, group=0, stream=None) -> "PendingTick":
        """`batch_call` without the wait: the tick's launches are queued (on `stream`, a torch.cuda.Stream; default: the current
        one), the new token ids travel to pinned host memory behind them, and the returned PendingTick's `result()` waits for
        them and gives what `batch_call` gives.  `group`: a key for the state of one set of lock-step streams - several
        groups (each on a stream of its own) may have a tick in flight at the same time: a tick ends with a host read, and
        between that read and the next tick's first launch the device would idle (94 us of a 1.06 ms tick of 32 streams,
        profiles/r06w_stream_tick_order.txt), as it mostly does under the ~25 small launches either side of the layers.
        A group's results must be taken in order (a later tick's `result()` takes the earlier ones first)."""
        if self.search == "online":
            raise NotImplementedError("batch_call decodes greedily; the block-synchronous beam search is per stream")
        if isinstance(speech, np.ndarray):
            speech = torch.tensor(speech)
        S = speech.size(0)
        bst = self._batches.get(group)
        if bst is None:
            bst = self._batches[group] = dict(frontend=None, encoder=None, last=[-1] * S, ids=[[] for _ in range(S)], pending=None,
                                              graph=self._tick_graphs.get((group, tuple(speech.shape))))
        m = self.asr_model
        ids_host, event = None, None
        ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
        with ctx:
            ids = None
            tg = bst.get("graph")
            sig = BatchTickGraph.signature(speech, bst) if (self.use_hipgraph and not is_final) else None
            if tg is not None and tg.graph is not None and sig is not None and sig == tg.sig:
                ids = tg.replay(speech, bst)  # steady state: one graph launch
            else:
                if tg is not None:
                    tg.leave(bst)
                speech_dev = speech.to(self.device, dtype=torch.float32, non_blocking=True)
                feats, bst["frontend"] = self.apply_frontend_batch(speech_dev, bst["frontend"], is_final=is_final)
                if feats is not None:
                    enc, y_len, bst["encoder"] = m.encoder.forward_infer_batch(feats, bst["encoder"], is_final)  # (a view: the encoder's own cat / contiguous() copies it once)
                    if y_len > 0:
                        ids = m.ctc.argmax(enc, as_int32=True)
                # a second tick of the same shape in a row: the ticks from here on are this one again - capture it
                if sig is not None and (tg is None or tg.graph is None) and BatchTickGraph.signature(speech, bst) == sig:
                    if tg is None:
                        tg = bst["graph"] = BatchTickGraph(self)
                    tg.sig = sig
                    tg.capture(speech_dev, bst)
            if ids is not None:
                ids_host = torch.empty(ids.shape, dtype=ids.dtype, pin_memory=True)
                ids_host.copy_(ids, non_blocking=True)  # ONE device -> host read per tick for all streams
                event = torch.cuda.Event()
                event.record()
        tick = PendingTick(self, bst, ids_host, event, bst["pending"])
        bst["pending"] = tick
        if is_final:
            tg = bst.get("graph")
            if tg is not None and tg.graph is not None:  # (the captured tick outlives the utterances: the group's next ones replay it)
                self._tick_graphs[(group, tg.sig[0])] = tg
            del self._batches[group]
        return tick

    def stream_pool(self) -> "StreamPool":
        """A pool of independent streams served by batched launches (streams may join, pause, finish at any tick)."""
        return StreamPool(self)

    def _encode_chunk(self, feats: torch.Tensor, is_final: bool) -> torch.Tensor:
        enc = self.asr_model.encoder
        if self.use_hipgraph:
            # the runner owns the encoder state; it replays a hipGraph for steady-state chunks of
            # the size it was created for and runs every other call eagerly
            if self._runner is None:
                self._runner = StreamingStepGraph(enc, feats.size(0))
            return self._runner(feats, is_final=is_final)
        y, _, self.encoder_states = enc.forward_infer(feats[None], torch.tensor([feats.size(0)]),
                                                      self.encoder_states, is_final)
        return y[0]

    def _extend_partial(self, enc: torch.Tensor):
        """Incremental G1: argmax of the new frames, collapsing repeats across the chunk seam."""
        m = self.asr_model
        ids = m.ctc.argmax(enc.unsqueeze(0), as_int32=True)[0].tolist()
        for t in ids:
            if t != self._last_id and t not in (m.blank_id, m.sos, m.eos):
                self._partial_ids.append(t)
            self._last_id = t

    def _results(self, is_final: bool):
        m = self.asr_model
        if is_final and self.beam_search is not None and self._enc_chunks:
            x = torch.cat(self._enc_chunks, dim=0)
            hyps = self.beam_search.search_batch(x.unsqueeze(0), [x.size(0)], self.maxlenratio,
                                                 self.minlenratio)[0][: self.nbest]
        else:
            yseq = torch.tensor([m.sos] + self._partial_ids + [m.eos], dtype=torch.long)
            hyps = [Hypothesis(yseq=yseq)]
        return self.assemble_hyps(hyps)

    def assemble_hyps(self, hyps):  # :338-365
        results = []
        for hyp in hyps[: self.nbest]:
            token_int = list(filter(lambda x: x != 0, hyp.yseq[1:-1].tolist()))
            token = self.converter.ids2tokens(token_int)
            text = self.tokenizer.tokens2text(token) if self.tokenizer is not None else None
            results.append((text, token, token_int, hyp))
        return results


class BatchTickGraph:
    """hipGraph replay of the steady-state tick of one group of lock-step streams (`batch_call` / `batch_call_async` with
    `use_hipgraph`): fed equal chunks, the carried buffers (waveform tail, feature tail, subsampled tail, context vectors)
    keep their shapes after a few ticks and every tick is the same ~70 launches - frontend, subsampling, block building, 36
    layer launches, CTC arg-max - except for the positional-encoding offset, which the captured tick reads from device memory
    (`n_processed_blocks_dev`, em_cb_build_blocks_rows_f32).  The first ticks run eagerly; once two consecutive ticks have the same
    signature the next one is captured over static input / state buffers and replayed from then on: one graph launch per tick
    instead of a launch sequence the HOST cannot issue as fast as the device runs its first fifteen kernels (140 us of a 1.04 ms
    tick, profiles/r06y_stream_tick_order.txt).  The final tick and any tick of another shape leave graph mode (the state is
    copied back out) and run eagerly."""

    _KEYS = ("prev_addin", "buffer_before_downsampling", "buffer_after_downsampling", "past_encoder_ctx")

    def __init__(self, s2t):
        self.s2t, self.graph, self.sig, self.in_graph, self.n_replays = s2t, None, None, False, 0

    @classmethod
    def signature(cls, speech, bst):
        fe, en = bst["frontend"], bst["encoder"]
        if fe is None or en is None or any(en.get(k) is None for k in cls._KEYS) or not isinstance(en["n_processed_blocks"], int) \
                or en["n_processed_blocks"] <= 0:
            return None
        return (tuple(speech.shape), tuple(fe["waveform_buffer"].shape)) + tuple(tuple(en[k].shape) for k in cls._KEYS)

    def _body(self):
        s2t, m = self.s2t, self.s2t.asr_model
        feats, nfe = s2t.apply_frontend_batch(self.s_in, self.s_fe, is_final=False)
        enc, y_len, nen = m.encoder.forward_infer_batch(feats, dict(self.s_en), False)
        return m.ctc.argmax(enc, as_int32=True), y_len, nfe, nen

    def capture(self, speech_dev, bst):
        """`bst`: the state BEHIND an eager tick of this shape (what the next tick starts from)."""
        fe, en = bst["frontend"], bst["encoder"]
        self.s_in = speech_dev.clone()
        self.s_fe = {"waveform_buffer": fe["waveform_buffer"].clone()}
        self.s_en = {k: en[k].clone() for k in self._KEYS}
        self.s_en["n_processed_blocks"] = 1  # (> 0: steady state; the real counts live on the device)
        self.s_en["n_processed_blocks_dev"] = torch.full((speech_dev.size(0),), en["n_processed_blocks"], dtype=torch.int32,
                                                         device=speech_dev.device)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=speech_dev.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):  # warm-up on a side stream (allocator, workspaces, shape caches); the static state is not written
            self._body()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ids, y_len, nfe, nen = self._body()
            self.s_fe["waveform_buffer"].copy_(nfe["waveform_buffer"])  # the state carried forward inside the graph
            for k in self._KEYS:
                self.s_en[k].copy_(nen[k])
            self.blocks_per_tick = nen["n_processed_blocks"] - 1
            self.s_en["n_processed_blocks_dev"].add_(self.blocks_per_tick)
        assert y_len > 0
        self.graph, self.s_ids, self.in_graph = g, ids, False

    def replay(self, speech, bst):
        """One tick from the static state; returns the (static) ids tensor.  `bst` keeps only the host-side block count while in
        graph mode."""
        if not self.in_graph:  # enter graph mode: this group's state into the static buffers
            fe, en = bst["frontend"], bst["encoder"]
            self.s_fe["waveform_buffer"].copy_(fe["waveform_buffer"])
            for k in self._KEYS:
                self.s_en[k].copy_(en[k])
            self.s_en["n_processed_blocks_dev"].fill_(en["n_processed_blocks"])
            self.in_graph = True
        self.s_in.copy_(speech, non_blocking=True)
        self.graph.replay()
        self.n_replays += 1
        bst["encoder"]["n_processed_blocks"] += self.blocks_per_tick
        return self.s_ids

    def leave(self, bst):
        """Back to eager ticks: the state out of the static buffers."""
        if self.in_graph:
            bst["frontend"] = {"waveform_buffer": self.s_fe["waveform_buffer"].clone()}
            for k in self._KEYS:
                bst["encoder"][k] = self.s_en[k].clone()
            self.in_graph = False


class PendingTick:
    """A tick of `Speech2TextStreaming.batch_call_async` whose launches are queued: `result()` waits for its token ids and
    returns, per stream, the ids decoded so far."""

    def __init__(self, s2t, bst, ids_host, event, prev):
        self._s2t, self._bst, self._ids_host, self._event, self._prev = s2t, bst, ids_host, event, prev
        self._res = None

    def done(self) -> bool:
        return self._res is not None or self._event is None or self._event.query()

    def result(self) -> List[List[int]]:
        if self._res is None:
            if self._prev is not None:
                self._prev.result()  # (the seam state - last id per stream - is carried in tick order)
                self._prev = None
            bst, m = self._bst, self._s2t.asr_model
            if self._ids_host is not None:
                self._event.synchronize()
                drop = (m.blank_id, m.sos, m.eos)
                for s_, row in enumerate(self._ids_host.tolist()):
                    last, out = bst["last"][s_], bst["ids"][s_]
                    for t in row:
                        if t != last and t not in drop:
                            out.append(t)
                        last = t
                    bst["last"][s_] = last
                self._ids_host = None
            self._res = [list(v) for v in bst["ids"]]
            if bst["pending"] is self:
                bst["pending"] = None
        return self._res


class StreamPool:
    """Many independent streams on one GPU WITHOUT lock step (a server's connections join, leave and finish
    at different times; the reference keeps all streaming state per `Speech2TextStreaming` object,
    espnet2/bin/asr_inference_streaming.py:205-336, so nothing forbids it).  Every stream has its own state - carried
    waveform samples, feature / subsampled-frame buffers, context vectors, number of processed blocks (= its position
    offset), last frame id and tokens so far.  `tick({stream id: (samples, is_final)})` serves whatever subset of the
    streams delivered audio at this tick (the rest are simply not active): the active streams are GROUPED by the shapes
    of their state - chunk length, buffer lengths, first block or not, final or not - and every group runs through the
    batched frontend / encoder launch sequence once (`apply_frontend_batch`, `forward_infer_batch` with one block count
    per row).  Streams fed equal chunks walk through the same short cycle of shapes whenever they joined, so a pool of
    live connections falls into a handful of groups.  Greedy CTC (incremental G1), as `batch_call`.  A stream's results
    equal what `Speech2TextStreaming.__call__` gives it alone (tests/test_gpu_streaming.py::test_stream_pool_ragged)."""

    def __init__(self, s2t: "Speech2TextStreaming"):
        if s2t.search == "online":
            raise NotImplementedError("the pool decodes greedily; the block-synchronous beam search is per stream")
        self.s2t = s2t
        self.streams = {}   # id -> dict(frontend, encoder, last, ids)
        self.groups_last_tick = 0

    @staticmethod
    def _shape(t):
        return None if t is None else tuple(t.shape[1:])

    def _signature(self, st, n, is_final):
        fe, en = st["frontend"], st["encoder"]
        wb = None if fe is None else self._shape(fe["waveform_buffer"])
        if en is None:
            es = None
        else:
            es = (self._shape(en["buffer_before_downsampling"]), self._shape(en["buffer_after_downsampling"]),
                  en["prev_addin"] is None, en["past_encoder_ctx"] is None, en["n_processed_blocks"] == 0)
        return (int(n), bool(is_final), fe is None, wb, es)

    @staticmethod
    def _stack(parts):
        """Per-stream state dicts (tensors with a leading stream dimension of 1, ints, None) -> one batched dict."""
        out = {}
        for k in parts[0]:
            vals = [p[k] for p in parts]
            if vals[0] is None:
                out[k] = None
            elif isinstance(vals[0], torch.Tensor):
                out[k] = torch.cat(vals, dim=0)
            else:
                out[k] = [int(v) for v in vals]  # n_processed_blocks: one per row
        return out

    @staticmethod
    def _unstack(state, i):
        if state is None:
            return None
        out = {}
        for k, v in state.items():
            if v is None:
                out[k] = None
            elif isinstance(v, torch.Tensor):
                out[k] = v[i : i + 1]
            elif isinstance(v, (list, tuple)):
                out[k] = int(v[i])
            else:
                out[k] = v
        return out

    @torch.no_grad()
    def tick(self, chunks) -> dict:
        """chunks: {stream id: (samples (n,) float tensor / array, is_final)}.  Unknown ids join the pool; a stream whose
        chunk is final leaves it after this tick.  Returns {stream id: token ids decoded so far} for the active streams."""
        s2t, m = self.s2t, self.s2t.asr_model
        groups = {}
        for sid, (speech, is_final) in chunks.items():
            if isinstance(speech, np.ndarray):
                speech = torch.from_numpy(speech)
            st = self.streams.setdefault(sid, dict(frontend=None, encoder=None, last=-1, ids=[]))
            groups.setdefault(self._signature(st, speech.numel(), is_final), []).append((sid, speech))
        self.groups_last_tick = len(groups)
        dev = next(m.parameters()).device
        drop = (m.blank_id, m.sos, m.eos)
        for sig, members in groups.items():
            is_final = sig[1]
            sts = [self.streams[sid] for sid, _ in members]
            wav = torch.stack([sp.to(torch.float32) for _, sp in members]).to(dev, non_blocking=True)
            fe = None if sts[0]["frontend"] is None else self._stack([st["frontend"] for st in sts])
            feats, fe_next = s2t.apply_frontend_batch(wav, fe, is_final=is_final)
            en_next, ids = None, None
            if feats is not None:
                en = None if sts[0]["encoder"] is None else self._stack([st["encoder"] for st in sts])
                if en is not None and isinstance(en["n_processed_blocks"], list) and len(set(en["n_processed_blocks"])) == 1:
                    en["n_processed_blocks"] = en["n_processed_blocks"][0]
                enc, y_len, en_next = m.encoder.forward_infer_batch(feats.contiguous(), en, is_final)
                if y_len > 0:
                    ids = m.ctc.argmax(enc, as_int32=True).cpu().tolist()  # one device -> host read per group and tick
            else:
                en_next = None if sts[0]["encoder"] is None else self._stack([st["encoder"] for st in sts])
            for i, (sid, _) in enumerate(members):
                st = self.streams[sid]
                st["frontend"] = self._unstack(fe_next, i)
                st["encoder"] = self._unstack(en_next, i) if feats is not None else st["encoder"]
                if ids is not None:
                    last, out = st["last"], st["ids"]
                    for t in ids[i]:
                        if t != last and t not in drop:
                            out.append(t)
                        last = t
                    st["last"] = last
        res = {sid: list(self.streams[sid]["ids"]) for sid in chunks}
        for sid, (_, is_final) in chunks.items():
            if is_final:
                del self.streams[sid]
        return res


# ---------------------------------------------------------------------- streaming decode CLI (asr.sh stage 12,
# `use_streaming=true`: python -m espnet2.bin.asr_inference_streaming)
def inference(output_dir: str, maxlenratio: float = 0.0, minlenratio: float = 0.0, batch_size: int = 1,
              dtype: str = "float32", beam_size: int = 20, ngpu: int = 1, seed: int = 0, ctc_weight: float = 0.5,
              lm_weight: float = 1.0, penalty: float = 0.0, nbest: int = 1, normalize_length: bool = False,
              num_workers: int = 1, log_level: Union[int, str] = "INFO", data_path_and_name_and_type=None,
              key_file: Optional[str] = None, asr_train_config: Optional[str] = None,
              asr_model_file: Optional[str] = None, lm_train_config: Optional[str] = None,
              lm_file: Optional[str] = None, word_lm_train_config: Optional[str] = None,
              word_lm_file: Optional[str] = None, token_type: Optional[str] = None, bpemodel: Optional[str] = None,
              allow_variable_data_keys: bool = False, sim_chunk_length: int = 0,
              disable_repetition_detection: bool = False, encoded_feat_length_limit: int = 0,
              decoder_text_length_limit: int = 0):
    """The reference's streaming `inference()` (espnet2/bin/asr_inference_streaming.py:362-494): one utterance at
    a time in input order, fed to `Speech2TextStreaming` whole (`sim_chunk_length == 0`) or in simulated chunks of
    `sim_chunk_length` samples with the remainder as the final call (:462-475); same result files and the same
    TooShortUttError placeholder (:476-479).  Returns an RTF summary like the offline CLI."""
    import time

    from espnet_amd.fileio.datadir_writer import DatadirWriter
    from espnet_amd.lib import TooShortUttError

    if batch_size > 1:
        raise NotImplementedError("batch decoding is not implemented")
    if word_lm_train_config is not None:
        raise NotImplementedError("Word LM is not implemented")
    if ngpu > 1:
        raise NotImplementedError("only single GPU decoding is supported")
    if ngpu < 1:
        raise RuntimeError("espnet_amd decodes on an MI355X only: pass --ngpu 1 (no CPU fallback)")
    logging.basicConfig(level=log_level,
                        format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s")
    torch.manual_seed(seed)
    np.random.seed(seed)
    speech2text = Speech2TextStreaming(
        asr_train_config=asr_train_config, asr_model_file=asr_model_file, lm_train_config=lm_train_config,
        lm_file=lm_file, token_type=token_type, bpemodel=bpemodel, device="cuda", maxlenratio=maxlenratio,
        minlenratio=minlenratio, dtype=dtype, beam_size=beam_size, ctc_weight=ctc_weight, lm_weight=lm_weight,
        penalty=penalty, nbest=nbest, normalize_length=normalize_length,
        disable_repetition_detection=disable_repetition_detection,
        decoder_text_length_limit=decoder_text_length_limit, encoded_feat_length_limit=encoded_feat_length_limit)
    loader = ASRTask.build_streaming_iterator(
        data_path_and_name_and_type, dtype="float32", batch_size=1, key_file=key_file, num_workers=num_workers,
        preprocess_fn=ASRTask.build_preprocess_fn(speech2text.asr_train_args, False),
        collate_fn=ASRTask.build_collate_fn(speech2text.asr_train_args, False),
        allow_variable_data_keys=allow_variable_data_keys, inference=True, ngpu=ngpu,
        bucket_window=1)  # a window of one utterance: results leave in input order, as the reference writes them
    fs = 16000
    fconf = getattr(speech2text.asr_train_args, "frontend_conf", None) or {}
    if isinstance(fconf.get("fs", None), int):
        fs = fconf["fs"]
    n_utts, n_samples = 0, 0
    t0 = time.perf_counter()
    with DatadirWriter(output_dir) as writer:
        for keys, batch in loader:
            assert all(isinstance(s, str) for s in keys), keys
            assert len(keys) == 1 == batch["speech"].size(0), keys
            speech = batch["speech"][0]
            try:
                if sim_chunk_length == 0:
                    results = speech2text(speech=speech, is_final=True)
                else:
                    n_full = len(speech) // sim_chunk_length
                    for i in range(n_full):
                        speech2text(speech=speech[i * sim_chunk_length : (i + 1) * sim_chunk_length],
                                    is_final=False)
                    # the remainder (possibly empty) closes the utterance; an utterance shorter than one
                    # chunk is a single final call (the reference's loop variable is undefined there)
                    results = speech2text(speech=speech[n_full * sim_chunk_length :], is_final=True)
            except TooShortUttError as e:
                logger.warning(f"Utterance {keys} {e}")
                speech2text.reset()
                hyp = Hypothesis(score=0.0, scores={}, states={}, yseq=[])
                results = [[" ", ["<space>"], [2], hyp]] * nbest
            key = keys[0]
            for n, (text, token, token_int, hyp) in zip(range(1, nbest + 1), results):
                w = writer[f"{n}best_recog"]
                w["token"][key] = " ".join(token)
                w["token_int"][key] = " ".join(map(str, token_int))
                w["score"][key] = str(hyp.score)
                if text is not None:
                    w["text"][key] = text
            n_utts += 1
            n_samples += len(speech)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    audio_s = n_samples / fs
    summary = dict(utterances=n_utts, audio_seconds=audio_s, wall_seconds=dt,
                   rtf=dt / audio_s if audio_s else float("nan"))
    logger.info("decoded %d utterances, %.1f audio-s in %.2f s: RTF %.5f", n_utts, audio_s, dt, summary["rtf"])
    return summary


def get_parser():
    """Option names, types and defaults of espnet2/bin/asr_inference_streaming.py:497-629 (`--config` yaml files
    work like config_argparse).  Differences, as in the offline CLI: `--ngpu` defaults to 1 (there is no CPU
    path), `--dtype` names the MFMA mode, `asr_model_file` is optional (random init when omitted)."""
    import argparse

    from espnet_amd.bin.asr_inference import _str2bool, _str2triple_str, _str_or_none

    p = argparse.ArgumentParser(description="ASR Decoding (MI355X, streaming)",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--config", type=str, default=None, help="yaml file with option defaults")
    p.add_argument("--log_level", type=lambda x: x.upper(), default="INFO",
                   choices=("CRITICAL", "ERROR", "WARNING", "INFO", "DEBUG", "NOTSET"))
    p.add_argument("--output_dir", type=str, required=True)
    p.add_argument("--ngpu", type=int, default=1, help="must be 1: one MI355X per process")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--dtype", default="float32", choices=["float16", "float32", "float64", "bfloat16"],
                   help="Data type (the reference's option and default).  float32 = exact-f32 MFMA (the parity mode); "
                        "bfloat16 = bf16 MFMA with f32 accumulation (the fast mode); float16 runs as bfloat16 and "
                        "float64 as float32, with a warning")
    p.add_argument("--num_workers", type=int, default=1, help="audio reader threads")
    g = p.add_argument_group("Input data related")
    g.add_argument("--data_path_and_name_and_type", type=_str2triple_str, required=True, action="append")
    g.add_argument("--key_file", type=_str_or_none)
    g.add_argument("--allow_variable_data_keys", type=_str2bool, default=False)
    g.add_argument("--sim_chunk_length", type=int, default=0,
                   help="The length of one chunk, to which speech will be divided for evaluation of streaming "
                        "processing.")
    g = p.add_argument_group("The model configuration related")
    g.add_argument("--asr_train_config", type=str, required=True)
    for name in ("asr_model_file", "lm_train_config", "lm_file", "word_lm_train_config", "word_lm_file"):
        g.add_argument(f"--{name}", type=str)
    g = p.add_argument_group("Beam-search related")
    g.add_argument("--batch_size", type=int, default=1)
    g.add_argument("--nbest", type=int, default=1)
    g.add_argument("--beam_size", type=int, default=20)
    g.add_argument("--penalty", type=float, default=0.0)
    g.add_argument("--maxlenratio", type=float, default=0.0)
    g.add_argument("--minlenratio", type=float, default=0.0)
    g.add_argument("--ctc_weight", type=float, default=0.5)
    g.add_argument("--lm_weight", type=float, default=1.0)
    g.add_argument("--disable_repetition_detection", type=_str2bool, default=False)
    g.add_argument("--encoded_feat_length_limit", type=int, default=0)
    g.add_argument("--decoder_text_length_limit", type=int, default=0)
    g = p.add_argument_group("Text converter related")
    g.add_argument("--token_type", type=_str_or_none, default=None, choices=["char", "bpe", "word", None])
    g.add_argument("--bpemodel", type=_str_or_none, default=None)
    g.add_argument("--normalize_length", type=_str2bool, default=False)
    return p


def main(cmd=None):
    import sys

    import yaml

    print(" ".join(sys.argv), file=sys.stderr)
    parser = get_parser()
    pre, _ = parser.parse_known_args(cmd)
    if pre.config is not None:  # config_argparse: yaml values become defaults, flags still win
        with open(pre.config, encoding="utf-8") as f:
            parser.set_defaults(**(yaml.safe_load(f) or {}))
    kwargs = vars(parser.parse_args(cmd))
    kwargs.pop("config", None)
    return inference(**kwargs)


if __name__ == "__main__":
    main()
