"""Speech2Text: the object API of the drop-in.

Mirrors espnet2/bin/asr_inference.py:73-677: same constructor keywords (those outside the hot
path are accepted and must keep their default), `__call__(speech) -> [(text, token, token_int,
Hypothesis)]` for ONE utterance, plus `batch_decode(...)` — the utterance-batched entry the
MI355X path adds (precedents in the reference: bin/asr_inference_k2.py:233-262,
bin/s2t_inference_ctc.py:700-749).

Decoding modes
  * ctc_greedy=True (G1): per-frame argmax + groupby + drop
    blank/sos/eos (asr_inference.py:574-575) — bit-exact integer contract, fully on device.
  * otherwise: label-synchronous joint CTC/attention beam search (BatchBeamSearch semantics,
    espnet2/legacy/nets/batch_beam_search.py) — see espnet_amd/nets/.
The log markers "speech length: N" and "best hypo: ..." that utils/calculate_rtf.py parses are kept.
"""
import logging
from pathlib import Path
from typing import Any, Dict, List, NamedTuple, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from espnet_amd.lib import TooShortUttError
from espnet_amd.nets.beam_search import Hypothesis
from espnet_amd.tasks.asr import ASRTask
from espnet_amd.text.token_id_converter import TokenIDConverter, build_tokenizer

logger = logging.getLogger(__name__)


def resolve_dtype(dtype: str) -> str:
    """The reference's --dtype values onto the two MFMA modes: float32 (exact-f32 MFMA, the parity mode) and
    bfloat16 (bf16 MFMA, f32 accumulate).  float16 / float64 have no counterpart on this path: they run as the
    nearest mode and say so."""
    if dtype in ("float32", "bfloat16"):
        return dtype
    if dtype == "float16":
        logger.warning("dtype float16 runs as bfloat16 on the MI355X path (bf16 MFMA, f32 accumulate)")
        return "bfloat16"
    if dtype == "float64":
        logger.warning("dtype float64 runs as float32 on the MI355X path (exact-f32 MFMA)")
        return "float32"
    raise ValueError(f"unknown dtype {dtype!r}")



class Speech2Text:
    def __init__(self, asr_train_config: Union[Path, str, None] = None,
                 asr_model_file: Union[Path, str, None] = None, transducer_conf: Optional[Dict] = None,
                 lm_train_config: Union[Path, str, None] = None, lm_file: Union[Path, str, None] = None,
                 ngram_scorer: str = "full", ngram_file: Union[Path, str, None] = None,
                 token_type: Optional[str] = None, bpemodel: Optional[str] = None, device: str = "cuda",
                 maxlenratio: float = 0.0, minlenratio: float = 0.0, batch_size: int = 1,
                 dtype: str = "float32", beam_size: int = 20, ctc_weight: float = 0.5,
                 lm_weight: float = 1.0, ngram_weight: float = 0.9, penalty: float = 0.0,
                 nbest: int = 1, normalize_length: bool = False, streaming: bool = False,
                 ctc_greedy: bool = False, **unsupported):
        for k, v in unsupported.items():
            if v not in (None, False, {}, [], 0.99, 5, -1, ["Linear"], "qint8"):
                raise NotImplementedError(f"Speech2Text({k}={v!r}) is outside the MI355X hot path")
        if transducer_conf is not None or ngram_file is not None or streaming:
            raise NotImplementedError("transducer / n-gram scorers and streaming=True (use Speech2TextStreaming): "
                                      "SURVEY.md §8(f) 'next' rows")
        if not str(device).startswith("cuda"):
            raise RuntimeError("espnet_amd.Speech2Text runs on an MI355X only (device='cuda'); no CPU fallback")
        # the reference's `dtype` is the model dtype; here it selects the MFMA mode
        dtype = resolve_dtype(dtype)
        asr_model, asr_train_args = ASRTask.build_model_from_file(
            asr_train_config, asr_model_file, device, compute_dtype=dtype)
        self.asr_model = asr_model
        self.asr_train_args = asr_train_args
        for m in (asr_model.frontend, asr_model.encoder):  # repack the weights for the kernels now, not
            pk = getattr(m, "_ensure_packed", None)        # inside the first decode call
            if pk is not None:
                pk(torch.device(device if ":" in str(device) else f"{device}:{torch.cuda.current_device()}"))
        self.device, self.dtype = device, dtype
        self.beam_size, self.ctc_weight, self.penalty = beam_size, ctc_weight, penalty
        self.maxlenratio, self.minlenratio, self.nbest = maxlenratio, minlenratio, nbest
        self.normalize_length = normalize_length
        self.ctc_greedy = ctc_greedy
        token_list = asr_model.token_list
        if token_type is None:
            token_type = getattr(asr_train_args, "token_type", None)
        if bpemodel is None:
            bpemodel = getattr(asr_train_args, "bpemodel", None)
        if token_type is None or (token_type == "bpe" and bpemodel is None):
            self.tokenizer = None
        else:
            self.tokenizer = build_tokenizer(token_type=token_type, bpemodel=bpemodel)
        self.converter = TokenIDConverter(token_list=token_list)
        self.beam_search = None
        if not ctc_greedy:
            from espnet_amd.nets.batch_beam_search import build_beam_search

            lm = None
            if lm_train_config is not None:  # asr_inference.py:179-191
                from espnet_amd.tasks.lm import LMTask

                lm_model, _ = LMTask.build_model_from_file(lm_train_config, lm_file, device, compute_dtype=dtype)
                lm = lm_model.lm
            self.lm = lm
            self.beam_search = build_beam_search(
                asr_model, beam_size=beam_size, ctc_weight=ctc_weight, penalty=penalty,
                lm_weight=lm_weight if lm is not None else 0.0, token_list=token_list,
                normalize_length=normalize_length, lm=lm)

    # ------------------------------------------------------------------ single utterance (reference API)
    @torch.no_grad()
    def __call__(self, speech: Union[torch.Tensor, np.ndarray]):
        if isinstance(speech, np.ndarray):
            speech = torch.tensor(speech)
        speech = speech.unsqueeze(0).to(torch.float32)  # (1, N), asr_inference.py:514
        logger.info("speech length: " + str(speech.size(1)))
        return self.batch_decode(speech, [speech.size(1)])[0]

    # ------------------------------------------------------------------ utterance batch (MI355X API)
    @torch.no_grad()
    def batch_decode(self, speech: torch.Tensor, speech_lengths: Sequence[int]):
        """speech (B, N) zero padded (host or device), lengths host ints.  Returns one reference-
        shaped result list per utterance; row b is what `__call__` returns for utterance b alone
        (isolated-utterance encoding: the reference decodes one utterance per call)."""
        speech = speech.to(self.device, torch.float32, non_blocking=True)
        st = self.asr_model.encode_device(speech, [int(n) for n in speech_lengths], isolate=True)
        if self.ctc_greedy:
            return self._finish_greedy(*self.decode_greedy_device(st))
        hyps = self.beam_search.search_batch(st.enc_act, st.olens, maxlenratio=self.maxlenratio,
                                             minlenratio=self.minlenratio)
        return [self._format(h[: self.nbest]) for h in hyps]

    @torch.no_grad()
    def batch_decode_async(self, speech: torch.Tensor, speech_lengths: Sequence[int]):
        """Like `batch_decode`, but for greedy CTC the device work is only ENQUEUED: the returned
        handle's `.result()` waits for this batch alone (an event after its D2H copy), so the caller can
        enqueue the next batch first and format this one while the GPU runs.  The beam search polls the
        device between step chunks and therefore completes inside this call."""
        if not self.ctc_greedy:
            # Round 6: several joint searches in flight (espnet_amd.nets.batch_beam_search.SearchLanes) - this batch is encoded
            # and its search started on a free lane's stream, the handle's `.result()` drives the lanes until it has ended.
            # A caller that submits batch n + 1 before it asks for batch n (the decode CLI does) keeps the lanes busy.
            lanes = self.__dict__.get("_lanes")
            if lanes is None:
                from espnet_amd.nets.batch_beam_search import SearchLanes

                # (four lanes, a host thread each: with ONE host thread two lanes already saturate the host's launch rate -
                # 2 / 3 / 4 lanes 60.4 / 60.3 / 60.9 ms per batch of 16 - with a thread per lane 4 lanes reach 48.0;
                # profiles/r06z_beam_lanes_threads_ab.txt.  `search_lanes` / `lane_threads` attributes: set before the first call)
                # (round 6, late: eight lanes while a search's launches leave most of the chip idle - up to ~200 rows = batch x beam:
                # 4 / 6 / 8 lanes 4 033 / 4 083 / 4 705 audio-s/s at 16 x beam 10; at 64 x beam 10 four lanes are as good as eight,
                # profiles/r06al_beam_lanes_sweep.txt)
                rows = int(speech.shape[0]) * int(getattr(self.beam_search, "beam_size", 10))
                n_lanes = max(1, int(getattr(self, "search_lanes", None) or (8 if rows <= 200 else 4)))
                lanes = self._lanes = SearchLanes([self.beam_search] + [self.beam_search.clone() for _ in range(n_lanes - 1)],
                                                  self.device, threaded=bool(getattr(self, "lane_threads", True)))
            k = lanes.free_lane()
            if k is None:  # (every lane carries a search nobody has asked for yet: finish this one on the calling stream)
                return _Done(self.batch_decode(speech, speech_lengths))
            with torch.cuda.stream(lanes.stream(k)):
                speech = speech.to(self.device, torch.float32, non_blocking=True)
                st = self.asr_model.encode_device(speech, [int(n) for n in speech_lengths], isolate=True)
            lanes.start(k, st.enc_act, st.olens, tag=(st, speech), maxlenratio=self.maxlenratio, minlenratio=self.minlenratio)
            return _PendingBeam(self, lanes, k)
        # (round 6: consecutive batches go to alternating HIP streams - every kernel of the greedy step takes the whole chip,
        # but one stream's launch fills the boundary between two dependent launches of the other: DESIGN.md 4g)
        pair = self.__dict__.get("_greedy_streams")
        if pair is None:
            pair = self._greedy_streams = [torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device)]
            self._greedy_turn = 0
        stream = pair[self._greedy_turn & 1]
        self._greedy_turn += 1
        stream.wait_stream(torch.cuda.current_stream())  # (whatever produced `speech` on the caller's stream)
        with torch.cuda.stream(stream):
            speech = speech.to(self.device, torch.float32, non_blocking=True)
            st = self.asr_model.encode_device(speech, [int(n) for n in speech_lengths], isolate=True)
            tokens, tlens = self.decode_greedy_device(st)
            tok_h = torch.empty(tokens.shape, dtype=tokens.dtype).pin_memory()
            len_h = torch.empty(tlens.shape, dtype=tlens.dtype).pin_memory()
            tok_h.copy_(tokens, non_blocking=True)
            len_h.copy_(tlens, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        return _PendingGreedy(self, tok_h, len_h, ev, (tokens, tlens, st, speech))

    def decode_greedy_device(self, st):
        """Device-resident G1 result: (tokens (B,T) i32 padded with -1, token_lens (B,) i32)."""
        _, tokens, tlens = self.asr_model.greedy_ctc_device(st)
        return tokens, tlens

    def _finish_greedy(self, tokens, tlens):
        tokens, tlens = tokens.cpu(), tlens.cpu()  # the one D2H copy of the batch
        return self._finish_greedy_host(tokens, tlens)

    def _finish_greedy_host(self, tokens, tlens):
        """Host tensors (B, T) / (B,) of one batch -> reference-shaped results.  One (B, T+2) int64 matrix holds
        every `yseq` (<sos> ids <eos>); the per-utterance tensors are row views of it."""
        sos, eos = self.asr_model.sos, self.asr_model.eos
        tok, lens = tokens.numpy(), tlens.numpy().tolist()
        B, T = tok.shape
        y = np.empty((B, T + 2), dtype=np.int64)
        y[:, 0] = sos
        y[:, 1 : T + 1] = tok
        for b, n in enumerate(lens):
            y[b, n + 1] = eos
        yt = torch.from_numpy(y)
        return [self._format([Hypothesis(yseq=yt[b, : n + 2])]) for b, n in enumerate(lens)]

    def _format(self, nbest_hyps):
        """asr_inference.py:652-677: strip sos/eos, drop id 0, ids -> tokens -> text."""
        results = []
        for hyp in nbest_hyps:
            token_int = [x for x in hyp.yseq[1:-1].tolist() if x != 0]
            token = self.converter.ids2tokens(token_int)
            text = self.tokenizer.tokens2text(token) if self.tokenizer is not None else None
            results.append((text, token, token_int, hyp))
        if results and logger.isEnabledFor(logging.INFO):
            logger.info("best hypo: " + "".join(results[0][1]) + "\n")
        return results

    @staticmethod
    def from_pretrained(model_tag: Optional[str] = None, **kwargs: Optional[Any]):
        if model_tag is not None:
            raise NotImplementedError("model zoo download needs network access (espnet_model_zoo)")
        return Speech2Text(**kwargs)


class _Done:
    def __init__(self, res):
        self._res = res

    def result(self):
        return self._res


class _PendingBeam:
    """A batch whose joint search is in flight on lane `k` of the Speech2Text's `SearchLanes`."""

    def __init__(self, s2t, lanes, k):
        self.s2t, self.lanes, self.k, self._res = s2t, lanes, k, None

    def result(self):
        if self._res is None:
            _tag, hyps = self.lanes.wait(self.k)
            self._res = [self.s2t._format(h[: self.s2t.nbest]) for h in hyps]
        return self._res


class _PendingGreedy:
    """Greedy batch whose kernels and D2H copy are enqueued; `keep` pins the device tensors until then."""

    def __init__(self, s2t, tok_h, len_h, ev, keep):
        self.s2t, self.tok_h, self.len_h, self.ev, self.keep = s2t, tok_h, len_h, ev, keep

    def result(self):
        self.ev.synchronize()
        self.keep = None
        return self.s2t._finish_greedy_host(self.tok_h, self.len_h)


# ---------------------------------------------------------------------- decode CLI (asr.sh stage 12)
def inference(output_dir: str, maxlenratio: float = 0.0, minlenratio: float = 0.0, batch_size: int = 1,
              dtype: str = "float32", beam_size: int = 20, ngpu: int = 1, seed: int = 0,
              ctc_weight: float = 0.5, lm_weight: float = 1.0, ngram_weight: float = 0.9,
              penalty: float = 0.0, nbest: int = 1, normalize_length: bool = False, num_workers: int = 1,
              log_level: Union[int, str] = "INFO", data_path_and_name_and_type=None,
              key_file: Optional[str] = None, asr_train_config: Optional[str] = None,
              asr_model_file: Optional[str] = None, lm_train_config: Optional[str] = None,
              lm_file: Optional[str] = None, word_lm_train_config: Optional[str] = None,
              word_lm_file: Optional[str] = None, ngram_file: Optional[str] = None,
              model_tag: Optional[str] = None, token_type: Optional[str] = None,
              bpemodel: Optional[str] = None, allow_variable_data_keys: bool = False,
              transducer_conf: Optional[dict] = None, streaming: bool = False, ctc_greedy: bool = False,
              bucket_window: int = 8, window_claim=None, **unsupported):
    """The reference's `inference()` (espnet2/bin/asr_inference.py:716-906) for the MI355X path: same
    keywords, same `output_dir/{n}best_recog/{token,token_int,score,text}` files in the key order of the
    input, same per-utterance TooShortUttError fallback (:851-858) — but `batch_size > 1` decodes
    length-bucketed utterance batches (`Speech2Text.batch_decode`) while reader threads load the next
    window.  Returns the RTF summary it also logs (`utils/calculate_rtf.py` reads the log markers)."""
    import time
    from collections import deque

    from espnet_amd.fileio.datadir_writer import DatadirWriter

    if word_lm_train_config is not None:
        raise NotImplementedError("Word LM is not implemented")
    if ngpu > 1:  # the reference raises here (:760-765) and scales by split key files in asr.sh; one node, one call:
        kw = dict(locals())
        kw.update(kw.pop("unsupported"))
        for k in ("time", "deque", "DatadirWriter"):
            kw.pop(k, None)
        return inference_multi_gpu(**kw)
    if ngpu < 1:
        raise RuntimeError("espnet_amd decodes on an MI355X only: pass --ngpu 1 (no CPU fallback)")
    logging.basicConfig(level=log_level,
                        format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s")
    torch.manual_seed(seed)
    np.random.seed(seed)
    speech2text = Speech2Text.from_pretrained(
        model_tag=model_tag, asr_train_config=asr_train_config, asr_model_file=asr_model_file,
        transducer_conf=transducer_conf, lm_train_config=lm_train_config, lm_file=lm_file,
        ngram_file=ngram_file, token_type=token_type, bpemodel=bpemodel, device="cuda",
        maxlenratio=maxlenratio, minlenratio=minlenratio, dtype=dtype, beam_size=beam_size,
        ctc_weight=ctc_weight, lm_weight=lm_weight, ngram_weight=ngram_weight, penalty=penalty, nbest=nbest,
        normalize_length=normalize_length, streaming=streaming, ctc_greedy=ctc_greedy, **unsupported)
    loader = ASRTask.build_streaming_iterator(
        data_path_and_name_and_type, dtype="float32", batch_size=batch_size, key_file=key_file,
        num_workers=num_workers, preprocess_fn=ASRTask.build_preprocess_fn(speech2text.asr_train_args, False),
        collate_fn=ASRTask.build_collate_fn(speech2text.asr_train_args, False),
        allow_variable_data_keys=allow_variable_data_keys, inference=True, ngpu=ngpu,
        bucket_window=bucket_window, window_claim=window_claim)
    fs = 16000
    fconf = getattr(speech2text.asr_train_args, "frontend_conf", None) or {}
    if isinstance(fconf.get("fs", None), int):
        fs = fconf["fs"]

    def too_short():  # asr_inference.py:852-854
        hyp = Hypothesis(score=0.0, scores={}, states={}, yseq=[])
        return [(" ", ["<space>"], [2], hyp)] * nbest

    def submit(keys, batch):
        """Enqueue one batch; returns a zero-argument callable giving its per-utterance results."""
        speech, lens = batch["speech"], [int(n) for n in batch["speech_lengths"]]
        for k, n in zip(keys, lens):
            logger.info(f"speech length: {n}")  # one line per utterance, as :520
        try:
            return speech2text.batch_decode_async(speech, lens).result
        except TooShortUttError as e:
            # a short utterance must not take the batch down: placeholder rows for the short ones
            # (:851-858), the rest decoded as one smaller batch
            bad = set(e.indices if e.indices is not None else range(len(keys)))
            for i in sorted(bad):
                logger.warning(f"Utterance {[keys[i]]} {e if len(bad) == 1 else 'is too short for subsampling'}")
            good = [i for i in range(len(keys)) if i not in bad]
            out = [too_short() for _ in keys]
            if good:
                sub = speech2text.batch_decode(speech[good, : max(lens[i] for i in good)], [lens[i] for i in good])
                for i, r in zip(good, sub):
                    out[i] = r
            return lambda: out

    pending, written, n_samples = {}, 0, 0
    t0 = time.perf_counter()
    with DatadirWriter(output_dir) as writer:
        def drain(keys, get):
            nonlocal written
            for k, res in zip(keys, get()):
                pending[k] = res
            order = loader.key_order
            while written < len(order) and order[written] in pending:  # emit in input order
                key = order[written]
                for n, (text, token, token_int, hyp) in zip(range(1, nbest + 1), pending.pop(key)):
                    w = writer[f"{n}best_recog"]
                    w["token"][key] = " ".join(token)
                    w["token_int"][key] = " ".join(map(str, token_int))
                    w["score"][key] = str(hyp.score)
                    if text is not None:
                        w["text"][key] = text
                written += 1

        # batches stay enqueued on the GPU while earlier ones are formatted and written: one for greedy CTC (two batches in
        # flight on alternating streams), three for the joint search (four searches in flight, a host thread each)
        import collections

        lanes = getattr(speech2text, "search_lanes", None) or (8 if int(batch_size) * int(beam_size) <= 200 else 4)  # (as batch_decode_async)
        depth = 1 if getattr(speech2text, "ctc_greedy", False) else max(1, int(lanes) - 1)
        inflight = collections.deque()
        for keys, batch in loader:
            assert all(isinstance(s, str) for s in keys), keys
            assert len(keys) == batch["speech"].size(0)
            n_samples += int(batch["speech_lengths"].sum())
            inflight.append((keys, submit(keys, batch)))
            while len(inflight) > depth:
                drain(*inflight.popleft())
        while inflight:
            drain(*inflight.popleft())
        assert not pending, sorted(pending)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    audio_s = n_samples / fs
    summary = dict(utterances=written, audio_seconds=audio_s, wall_seconds=dt,
                   rtf=dt / audio_s if audio_s else float("nan"),
                   native_reader_windows=getattr(loader, "native_windows", 0),
                   reader_seconds=dict(getattr(getattr(loader, "_wav", None), "seconds", {})))
    logger.info("decoded %d utterances, %.1f audio-s in %.2f s: RTF %.5f (%.0f audio-s/s)", written, audio_s,
                dt, summary["rtf"], audio_s / dt if dt else float("nan"))
    return summary


# ---------------------------------------------------------------------- --ngpu N: one process per GPU
def _read_keys(data_path_and_name_and_type, key_file):
    path = key_file if key_file is not None else data_path_and_name_and_type[0][0]
    with open(path, encoding="utf-8") as f:
        return [ln.split(maxsplit=1)[0] for ln in f if ln.strip()]


def _first_float(text: str) -> float:
    """The score column as the writer formats it (`str(hyp.score)`: a float, or `tensor(-12.3, ...)`)."""
    import re

    m = re.search(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?|[-+]?(?:inf|nan)", text)
    return float(m.group(0)) if m else 0.0


def merge_shard_outputs(output_dir, keys, world: int, nbest: int):
    """`output_dir/output.{r+1}/{n}best_recog/{token,token_int,score,text}` of the ranks -> the same files under
    `output_dir`, rows in key order (asr.sh:1636-1648 does this with cat + sort -k1)."""
    from pathlib import Path

    out = Path(output_dir)
    for n in range(1, nbest + 1):
        for name in ("token", "token_int", "score", "text"):
            rows = {}
            for r in range(world):
                f = out / f"output.{r + 1}" / f"{n}best_recog" / name
                if f.exists():
                    for ln in f.read_text(encoding="utf-8").splitlines():
                        k, _, v = ln.partition(" ")
                        rows[k] = v
            if rows:
                (out / f"{n}best_recog").mkdir(parents=True, exist_ok=True)
                with (out / f"{n}best_recog" / name).open("w", encoding="utf-8") as w:
                    for k in keys:
                        if k in rows:
                            w.write(f"{k} {rows[k]}\n")


def sharded_decode_rank(decode_claimed, keys, output_dir, nbest: int, max_len: int, device, window: int = 1):
    """One rank of `--ngpu N`, DYNAMIC dispatch (round 4; was: one static contiguous slab per rank).  The key list is
    cut into windows of `window` utterances (the loader's read-ahead window); `decode_claimed(claim, shard_dir)` runs
    this rank's ONE decode loop over the windows `claim(w)` grants it (espnet_amd.distributed.WindowClaimer over a
    shared counter: a rank asks for its next window when it reaches the previous one, so a slow GPU or a run of long
    utterances takes fewer) into `output_dir/output.{rank+1}` and returns {key: (token ids, score)} of its 1-best.
    The records travel with their global utterance index and are collated ONCE, at the end
    (`gather_variable_records`); rank 0 merges the shard files in key order and cross-checks them against the
    collated records.  A failing rank makes every rank raise (no one is left inside a collective)."""
    from pathlib import Path

    import torch.distributed as dist

    from espnet_amd import distributed as D

    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    shard_dir = Path(output_dir) / f"output.{rank + 1}"
    work = D.work_store() if world > 1 else None
    win_key = D.call_key("asr_inference_windows", work, world)
    claim = D.WindowClaimer(D.SharedCounter(work, win_key))
    # rows of an earlier run into the same output_dir must not be taken for this rank's (the writer opens its files
    # lazily: a rank that claims no window would never truncate them)
    if shard_dir.is_dir():
        import shutil

        for sub in shard_dir.glob("*best_recog"):
            shutil.rmtree(sub, ignore_errors=True)
    err, stats, rec = None, {}, None
    try:
        mine, stats = decode_claimed(claim, shard_dir)
        index = {k: u for u, k in enumerate(keys)}
        ks = sorted(mine, key=index.__getitem__)
        rec = D.pack_indexed_records([index[k] for k in ks], [mine[k][0] for k in ks], [mine[k][1] for k in ks],
                                     max_len, device)
    except Exception as e:  # re-raised below, after the peers have been told
        err = e
    if world > 1:
        flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        failed = int(flag.item()) != 0
    else:
        failed = err is not None
    if err is not None:
        raise err
    if failed:
        raise RuntimeError(f"rank {rank}: another rank failed before the hypothesis collation; nothing was gathered")
    hyps = D.unpack_indexed_records(D.gather_variable_records(rec), len(keys))
    D.release_key(work, win_key, rank)  # (behind the collation: every rank is past its last claim)
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        merge_shard_outputs(output_dir, keys, world, nbest)
        f = Path(output_dir) / "1best_recog" / "token_int"
        merged = {ln.partition(" ")[0]: [int(t) for t in ln.partition(" ")[2].split()]
                  for ln in f.read_text(encoding="utf-8").splitlines()} if f.exists() else {}
        for k, (toks, _) in zip(keys, hyps):
            assert merged.get(k, []) == toks, f"collated record of {k} differs from the merged shard file"
    return hyps, dict(stats, windows=list(claim.claimed))


def _multi_gpu_worker(rank: int, world: int, port: int, kw: dict, q):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        res = _inference_rank(kw)
    except BaseException as e:  # tell the parent, then leave WITHOUT a collective: the peers are not in a matching one
        import traceback

        q.put((rank, RuntimeError(f"rank {rank}: {type(e).__name__}: {e}\n{traceback.format_exc()}")))
        try:
            dist.destroy_process_group()
        except Exception:
            pass
        raise SystemExit(1)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def _inference_rank(kw: dict):
    keys = _read_keys(kw["data_path_and_name_and_type"], kw["key_file"])
    nbest = kw["nbest"]
    window = max(1, int(kw.get("batch_size", 1))) * max(1, int(kw.get("bucket_window", 8)))

    def decode_claimed(claim, shard_dir):
        shard_dir.mkdir(parents=True, exist_ok=True)
        sub = dict(kw, output_dir=str(shard_dir), ngpu=1, window_claim=claim)  # the whole key list, claimed windows only
        st = inference(**sub)
        ti, sc = shard_dir / "1best_recog" / "token_int", shard_dir / "1best_recog" / "score"
        rows = {ln.partition(" ")[0]: ln.partition(" ")[2] for ln in ti.read_text().splitlines()} if ti.exists() else {}
        srow = {ln.partition(" ")[0]: ln.partition(" ")[2] for ln in sc.read_text().splitlines()} if sc.exists() else {}
        mine = {k for w in claim.claimed for k in keys[w * window : (w + 1) * window]}  # only what THIS call decoded
        return {k: ([int(t) for t in v.split()], _first_float(srow.get(k, "0"))) for k, v in rows.items() if k in mine}, st

    hyps, st = sharded_decode_rank(decode_claimed, keys, kw["output_dir"], nbest, 4096, torch.device("cuda"), window)
    return dict(st, utterances_total=len(hyps))


def _collect_ranks(procs, q, poll_s: float = 0.5):
    """One result per rank process from `q`, or an exception: a rank that reports a failure or dies without
    reporting (killed, out of memory in native code) ends the whole job - the surviving ranks are terminated
    instead of being left inside a collective nobody will complete (the reference's split-job flow surfaces a
    failed job the same way, asr.sh:1636-1648)."""
    import queue as _queue

    res, err = {}, None
    while len(res) < len(procs) and err is None:
        try:
            rank, r = q.get(timeout=poll_s)
        except _queue.Empty:
            dead = [i for i, p in enumerate(procs) if p.exitcode not in (None, 0) and i not in res]
            if dead:
                err = RuntimeError(f"rank process {dead[0]} exited with {procs[dead[0]].exitcode} before reporting")
            elif all(p.exitcode is not None for p in procs) and q.empty():
                err = RuntimeError("rank processes exited without reporting a result")
            continue
        if isinstance(r, BaseException):
            err = r
        else:
            res[rank] = r
    if err is not None:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=10)
        raise RuntimeError(f"--ngpu {len(procs)}: {err}") from (err if isinstance(err, BaseException) else None)
    for p in procs:
        p.join()
        if p.exitcode != 0:
            raise RuntimeError(f"rank process exited with {p.exitcode}")
    return res


def inference_multi_gpu(ngpu: int, **kw):
    """`--ngpu N` on one node: N processes (one per MI355X) pull read-ahead windows of the key file from a shared
    counter (dynamic dispatch, `sharded_decode_rank`); the hypotheses are collated once at the end with
    `espnet_amd.distributed.gather_variable_records` (RCCL all-gather of fixed-width indexed records) and the
    per-rank files merged in key order.  Under torchrun (RANK / WORLD_SIZE set) this process is one of
    the ranks; otherwise the ranks are spawned here."""
    import os
    import socket

    import torch.distributed as dist

    kw = dict(kw, ngpu=ngpu)
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) == ngpu:
        local = int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        try:
            return _inference_rank(kw)
        finally:
            dist.barrier()
            dist.destroy_process_group()
    if torch.cuda.device_count() < ngpu:
        raise RuntimeError(f"--ngpu {ngpu} but {torch.cuda.device_count()} GPUs are visible")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = torch.multiprocessing.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_multi_gpu_worker, args=(r, ngpu, port, kw, q)) for r in range(ngpu)]
    for p in procs:
        p.start()
    res = _collect_ranks(procs, q)
    audio = sum(r.get("audio_seconds", 0.0) for r in res.values())
    wall = max(r.get("wall_seconds", 0.0) for r in res.values())
    return dict(utterances=sum(r.get("utterances", 0) for r in res.values()), audio_seconds=audio, wall_seconds=wall,
                rtf=wall / audio if audio else float("nan"), ranks=ngpu)


def _str2bool(v: str) -> bool:
    if v.lower() in ("true", "1", "yes", "y", "t"):
        return True
    if v.lower() in ("false", "0", "no", "n", "f"):
        return False
    raise ValueError(f"not a boolean: {v!r}")


def _str_or_none(v: str):
    return None if v.strip().lower() in ("none", "null", "nil", "") else v


def _str2triple_str(v: str):
    """`path,name,type` (espnet2/utils/types.py str2triple_str)."""
    parts = [p.strip().strip("()").strip("'\"") for p in v.split(",")]
    if len(parts) != 3:
        raise ValueError(f"expected 'path,name,type': {v!r}")
    return tuple(parts)


def get_parser():
    """Option names, types and defaults of espnet2/bin/asr_inference.py:911-1137 for the options this
    path implements; `--config` yaml files work like config_argparse; options of other tasks are
    accepted at their reference default and rejected otherwise (Speech2Text checks)."""
    import argparse

    p = argparse.ArgumentParser(description="ASR Decoding (MI355X)",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--config", type=str, default=None, help="yaml file with option defaults")
    p.add_argument("--log_level", type=lambda x: x.upper(), default="INFO",
                   choices=("CRITICAL", "ERROR", "WARNING", "INFO", "DEBUG", "NOTSET"))
    p.add_argument("--output_dir", type=str, required=True)
    p.add_argument("--ngpu", type=int, default=1,
                   help="1 = this process decodes on one MI355X; N > 1 = N processes, one per GPU, each decoding a "
                        "slab of the key file (espnet_amd.distributed); 0 is rejected (no CPU fallback)")
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
    g = p.add_argument_group("The model configuration related")
    for name in ("asr_train_config", "asr_model_file", "lm_train_config", "lm_file", "word_lm_train_config",
                 "word_lm_file", "ngram_file", "model_tag"):
        g.add_argument(f"--{name}", type=str)
    g = p.add_argument_group("Beam-search related")
    g.add_argument("--batch_size", type=int, default=1, help="utterances per device batch")
    g.add_argument("--bucket_window", type=int, default=8,
                   help="read-ahead window, in batches, that is sorted by length before batching")
    g.add_argument("--nbest", type=int, default=1)
    g.add_argument("--beam_size", type=int, default=20)
    g.add_argument("--penalty", type=float, default=0.0)
    g.add_argument("--maxlenratio", type=float, default=0.0)
    g.add_argument("--minlenratio", type=float, default=0.0)
    g.add_argument("--ctc_weight", type=float, default=0.5)
    g.add_argument("--lm_weight", type=float, default=1.0)
    g.add_argument("--ngram_weight", type=float, default=0.9)
    g.add_argument("--streaming", type=_str2bool, default=False)
    g.add_argument("--normalize_length", type=_str2bool, default=False)
    g.add_argument("--ctc_greedy", type=_str2bool, default=False,
                   help="G1: arg-max + collapse on the device instead of the beam search")
    g = p.add_argument_group("Text converter related")
    g.add_argument("--token_type", type=_str_or_none, default=None, choices=["char", "bpe", "word", None])
    g.add_argument("--bpemodel", type=_str_or_none, default=None)
    g = p.add_argument_group("Options of other decoding modes (accepted at the reference default only)")
    for name, typ, default in _OTHER_MODE_OPTIONS:
        if typ == "list":
            g.add_argument(f"--{name}", type=str, nargs="*", default=default)
        else:
            g.add_argument(f"--{name}", type=typ, default=default)
    return p


def _yaml_or_none(v: str):
    import yaml

    return None if v.strip().lower() in ("none", "null", "") else yaml.safe_load(v)


# (name, type, reference default) of asr_inference.py:1010-1136 options that select other models
_OTHER_MODE_OPTIONS = [
    ("enh_s2t_task", _str2bool, False), ("multi_asr", _str2bool, False),
    ("quantize_asr_model", _str2bool, False), ("quantize_lm", _str2bool, False),
    ("quantize_modules", "list", ["Linear"]), ("quantize_dtype", str, "qint8"),
    ("transducer_conf", _yaml_or_none, None), ("hugging_face_decoder", _str2bool, False),
    ("hugging_face_decoder_conf", _yaml_or_none, {}), ("time_sync", _str2bool, False),
    ("prompt_token_file", _str_or_none, None), ("lang_prompt_token", _str_or_none, None),
    ("nlp_prompt_token", _str_or_none, None), ("partial_ar", _str2bool, False),
    ("threshold_probability", float, 0.99), ("max_seq_len", int, 5), ("max_mask_parallel", int, -1),
]


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
