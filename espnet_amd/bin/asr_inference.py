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

from espnet_amd.nets.beam_search import Hypothesis
from espnet_amd.tasks.asr import ASRTask
from espnet_amd.text.token_id_converter import TokenIDConverter, build_tokenizer

logger = logging.getLogger(__name__)


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
        asr_model, asr_train_args = ASRTask.build_model_from_file(
            asr_train_config, asr_model_file, device, compute_dtype=dtype)
        self.asr_model = asr_model
        self.asr_train_args = asr_train_args
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
        shaped result list per utterance."""
        speech = speech.to(self.device, torch.float32, non_blocking=True)
        st = self.asr_model.encode_device(speech, [int(n) for n in speech_lengths])
        if self.ctc_greedy:
            return self._finish_greedy(*self.decode_greedy_device(st))
        hyps = self.beam_search.search_batch(st.enc_act, st.olens, maxlenratio=self.maxlenratio,
                                             minlenratio=self.minlenratio)
        return [self._format(h[: self.nbest]) for h in hyps]

    def decode_greedy_device(self, st):
        """Device-resident G1 result: (tokens (B,T) i32 padded with -1, token_lens (B,) i32)."""
        _, tokens, tlens = self.asr_model.greedy_ctc_device(st)
        return tokens, tlens

    def _finish_greedy(self, tokens, tlens):
        tokens, tlens = tokens.cpu(), tlens.cpu()  # the one D2H copy of the batch
        sos, eos = self.asr_model.sos, self.asr_model.eos
        out = []
        for b in range(tokens.size(0)):
            ids = tokens[b, : int(tlens[b])].tolist()
            yseq = torch.tensor([sos] + ids + [eos], dtype=torch.long)
            out.append(self._format([Hypothesis(yseq=yseq)]))
        return out

    def _format(self, nbest_hyps):
        """asr_inference.py:652-677: strip sos/eos, drop id 0, ids -> tokens -> text."""
        results = []
        for hyp in nbest_hyps:
            token_int = hyp.yseq[1:-1].tolist()
            token_int = list(filter(lambda x: x != 0, token_int))
            token = self.converter.ids2tokens(token_int)
            text = self.tokenizer.tokens2text(token) if self.tokenizer is not None else None
            results.append((text, token, token_int, hyp))
        if results:
            logger.info("best hypo: " + "".join(results[0][1]) + "\n")
        return results

    @staticmethod
    def from_pretrained(model_tag: Optional[str] = None, **kwargs: Optional[Any]):
        if model_tag is not None:
            raise NotImplementedError("model zoo download needs network access (espnet_model_zoo)")
        return Speech2Text(**kwargs)
