"""ESPnetASRModel (inference side) on the MI355X.

Mirrors espnet2/asr/espnet_model.py:45-467 for the attributes and methods the inference path
touches: `encode(speech, speech_lengths)`, `_extract_feats`, `.frontend/.normalize/.encoder/
.decoder/.ctc`, `.sos/.eos/.blank_id/.token_list/.vocab_size`.  State-dict keys are the
reference's (`frontend.logmel.melmat`, `encoder.*`, `decoder.*`, `ctc.ctc_lo.*`).

`encode` keeps the reference signature (padded batch + lengths, any B); `encode_device` is the
batched device-resident entry the MI355X drop-in adds (precedent in the reference:
espnet2/bin/asr_inference_k2.py:233-262 and bin/s2t_inference_ctc.py:700-749).
"""
from typing import List, Optional, Tuple

import torch

from espnet_amd import lib as L


class EncoderState:
    """Device-resident result of one encode call."""

    __slots__ = ("enc_out", "enc_act", "olens", "olens_dev", "feats", "flens", "ctc_ids")

    def __init__(self, enc_out, enc_act, olens, olens_dev, feats, flens, ctc_ids=None):
        self.enc_out, self.enc_act = enc_out, enc_act
        self.olens, self.olens_dev = olens, olens_dev
        self.feats, self.flens = feats, flens
        self.ctc_ids = ctc_ids  # (B, T) i32 per-frame CTC arg-max when the encoder's last kernel produced it


class ESPnetASRModel(torch.nn.Module):
    def __init__(self, vocab_size: int, token_list, frontend, specaug, normalize, preencoder,
                 encoder, postencoder, decoder, ctc, joint_network=None, aux_ctc: dict = None,
                 ctc_weight: float = 0.5, interctc_weight: float = 0.0, ignore_id: int = -1,
                 lsm_weight: float = 0.0, length_normalized_loss: bool = False,
                 report_cer: bool = True, report_wer: bool = True, sym_space: str = "<space>",
                 sym_blank: str = "<blank>", transducer_multi_blank_durations: List = [],
                 transducer_multi_blank_sigma: float = 0.05, sym_sos: str = "<sos/eos>",
                 sym_eos: str = "<sos/eos>", extract_feats_in_collect_stats: bool = True,
                 lang_token_id: int = -1, autocast_frontend: bool = False):
        assert 0.0 <= ctc_weight <= 1.0, ctc_weight
        assert 0.0 <= interctc_weight < 1.0, interctc_weight
        super().__init__()
        if specaug is not None or preencoder is not None or postencoder is not None or joint_network is not None:
            raise NotImplementedError("specaug/preencoder/postencoder/transducer are outside the hot path")
        token_list = list(token_list)
        # espnet_model.py:76-87
        self.blank_id = token_list.index(sym_blank) if sym_blank in token_list else 0
        self.sos = token_list.index(sym_sos) if sym_sos in token_list else vocab_size - 1
        self.eos = token_list.index(sym_eos) if sym_eos in token_list else vocab_size - 1
        self.vocab_size = vocab_size
        self.ignore_id = ignore_id
        self.ctc_weight = ctc_weight
        self.interctc_weight = interctc_weight
        self.token_list = token_list.copy()
        self.frontend = frontend
        self.specaug = None
        self.normalize = normalize
        self.preencoder = None
        self.postencoder = None
        self.encoder = encoder
        self.use_transducer_decoder = False
        self._len_cache = {}
        # espnet_model.py:167-192
        self.decoder = decoder if ctc_weight < 1.0 else None
        self.ctc = None if ctc_weight == 0.0 else ctc
        # the fused encoder path can take the CTC head's arg-max inside its last kernel (csrc/block.hip EM_BLOCK_CTC);
        # a plain attribute, not a submodule: the head's parameters stay under `ctc.` only
        if self.ctc is not None and hasattr(self.encoder, "_pack_fused_ctc"):
            object.__setattr__(self.encoder, "fused_ctc", self.ctc)

    # ------------------------------------------------------------------ packing
    def set_compute_dtype(self, dtype: str):
        """'float32' (exact-f32 MFMA parity mode) or 'bfloat16' (bf16 MFMA, f32 accumulate)."""
        for m in (self.encoder, self.ctc, self.decoder):
            if m is not None and hasattr(m, "compute_dtype"):
                m.compute_dtype = dtype
                m.invalidate()

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        for m in (self.encoder, self.ctc, self.decoder):
            if m is not None and hasattr(m, "invalidate"):
                m.invalidate()
        if self.frontend is not None:
            self.frontend._packed = None
        return r

    # ------------------------------------------------------------------ encode
    def encode_device(self, speech: torch.Tensor, speech_lengths: List[int],
                      isolate: bool = False) -> EncoderState:
        """speech (B, N) f32 ON THE GPU (zero padded), speech_lengths host ints.  No host sync.
        isolate=False: the reference's padded-batch semantics (`encode`, espnet_model.py:380-448: STFT
        reflect padding at the padded end, unmasked depthwise conv).  isolate=True: batching is
        transparent — row b equals what `Speech2Text.__call__` computes for utterance b alone."""
        L.require_gpu(speech, "speech")
        nmax = max(int(n) for n in speech_lengths)
        speech = speech[:, :nmax].contiguous()  # espnet_model.py:454 (crop to the longest)
        dev = speech.device
        flens = self.frontend.feature_lengths(speech_lengths)
        # length vectors live on the device; the last few distinct ones are kept so a repeating batch
        # shape costs no H2D copy (and a captured hipGraph of the pass keeps valid pointers)
        ckey = (tuple(int(n) for n in speech_lengths), bool(isolate), dev)
        cached = self._len_cache.get(ckey)
        if cached is None:
            flens_dev = torch.tensor(flens, dtype=torch.int32).to(dev, non_blocking=True)
            wlens_dev = None
            if isolate and len(set(ckey[0])) > 1:
                if min(ckey[0]) <= self.frontend.n_fft // 2:
                    raise ValueError(f"an input of {min(ckey[0])} samples is too short for reflect padding "
                                     f"of {self.frontend.n_fft // 2}")  # torch.stft raises for it as well
                wlens_dev = torch.tensor(ckey[0], dtype=torch.int32).to(dev, non_blocking=True)
            if len(self._len_cache) >= 8:
                self._len_cache.pop(next(iter(self._len_cache)))
            self._len_cache[ckey] = (flens_dev, wlens_dev)
        else:
            flens_dev, wlens_dev = cached
        feats = self.frontend.forward_device(speech, flens_dev, wlens_dev)
        partial = None
        if self.normalize is not None:
            if hasattr(self.normalize, "partial_sums"):  # UtteranceMVN: subtraction fused into conv1
                partial = self.normalize.partial_sums(feats, flens_dev)
            else:  # GlobalMVN: in place
                feats = self.normalize.forward_device(feats, flens_dev)
        enc_out, enc_act, olens, olens_dev = self.encoder.forward_device(feats, flens, flens_dev, partial,
                                                                         isolate=isolate)
        return EncoderState(enc_out, enc_act, olens, olens_dev, feats, flens,
                            getattr(self.encoder, "last_ctc_ids", None))

    def encode(self, speech: torch.Tensor, speech_lengths: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Frontend + Encoder (espnet_model.py:380-448).  speech (B, N), speech_lengths (B,)."""
        assert speech_lengths.dim() == 1, speech_lengths.shape
        st = self.encode_device(speech.to(torch.float32), [int(v) for v in speech_lengths.tolist()])
        self._last_state = st
        return st.enc_out, torch.tensor(st.olens, dtype=torch.long, device=speech.device)

    def _extract_feats(self, speech: torch.Tensor, speech_lengths: torch.Tensor):
        """espnet_model.py:450-467."""
        speech = speech[:, : int(speech_lengths.max())]
        return self.frontend(speech, speech_lengths)

    # ------------------------------------------------------------------ greedy CTC (G1)
    def greedy_ctc_device(self, st: EncoderState, out=None):
        """argmax + groupby + drop blank/sos/eos (bin/asr_inference.py:574-575) for the whole batch
        on the device.  Returns (ids, tokens, token_lens) device tensors (int32); `out` = (tokens, token_lens)
        tensors to write into instead of fresh ones (the collation's record slot)."""
        if self.ctc is None:
            raise RuntimeError("model has no CTC head (ctc_weight == 0)")
        sos_eos = self.sos if self.sos == self.eos else -2
        if st.ctc_ids is not None:  # arg-max already taken inside the encoder's last kernel
            return self.ctc.collapse_device(st.ctc_ids, st.olens_dev, self.blank_id, sos_eos, out=out)
        return self.ctc.greedy_device(st.enc_act, st.olens_dev, self.blank_id, sos_eos, out=out)
