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
                 beam_size: int = 1, ctc_weight: float = 0.5, lm_weight: float = 0.0,
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
        self.reset()

    def reset(self):
        self.frontend_states = None
        self.encoder_states = None
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
        wav = speech_to_process.unsqueeze(0).to(torch.float32).to(self.device)
        n = wav.size(1)
        m = self.asr_model
        flens = m.frontend.feature_lengths([n])
        flens_dev = torch.tensor(flens, dtype=torch.int32).to(wav.device)
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
        ids = m.ctc.argmax(enc.unsqueeze(0))[0].tolist()
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
