"""ASRTask: the plugin/registry boundary of the drop-in.

Mirrors espnet2/tasks/asr.py:96-217 (name -> class tables), :512-651 (`build_model`) and
espnet2/tasks/abs_task.py:2456-2561 (`build_model_from_file`): a `config.yaml` written by the
reference (`ASRTask.main --dry_run` or any training run) names `frontend/normalize/encoder/decoder`
plus `*_conf` dicts; the same yaml builds the MI355X model here, and the reference's `.pth`
state_dict loads unchanged (`strict=False`, unwrapping "module"/"state_dict" like the reference).
Names outside the accelerated path raise NotImplementedError (the stock classes stay registered
in the reference for those).
"""
import argparse
from pathlib import Path
from typing import Optional, Union

import torch
import yaml

from espnet_amd.asr.ctc import CTC
from espnet_amd.asr.decoder.transformer_decoder import TransformerDecoder
from espnet_amd.asr.encoder.conformer_encoder import ConformerEncoder
from espnet_amd.asr.encoder.contextual_block_conformer_encoder import ContextualBlockConformerEncoder
from espnet_amd.asr.encoder.e_branchformer_encoder import BranchformerEncoder, EBranchformerEncoder
from espnet_amd.asr.espnet_model import ESPnetASRModel
from espnet_amd.asr.frontend.default import DefaultFrontend
from espnet_amd.layers.global_mvn import GlobalMVN
from espnet_amd.layers.utterance_mvn import UtteranceMVN

frontend_choices = {"default": DefaultFrontend}
normalize_choices = {"utterance_mvn": UtteranceMVN, "global_mvn": GlobalMVN}
encoder_choices = {"conformer": ConformerEncoder,
                   "contextual_block_conformer": ContextualBlockConformerEncoder,
                   "e_branchformer": EBranchformerEncoder, "branchformer": BranchformerEncoder}
decoder_choices = {"transformer": TransformerDecoder}
model_choices = {"espnet": ESPnetASRModel}


def _get(args, name, default=None):
    return getattr(args, name, default) if not isinstance(args, dict) else args.get(name, default)


def _choice(table, kind, name):
    if name is None:
        return None
    if name not in table:
        raise NotImplementedError(
            f"{kind}={name!r} is outside the MI355X hot path (supported: {sorted(table)})")
    return table[name]


class ASRTask:
    num_optimizers = 1

    @classmethod
    def build_model(cls, args) -> ESPnetASRModel:
        """espnet2/tasks/asr.py:512-651."""
        if isinstance(args, dict):
            args = argparse.Namespace(**args)
        token_list = _get(args, "token_list")
        if isinstance(token_list, str):
            with open(token_list, encoding="utf-8") as f:
                token_list = [line.rstrip("\n") for line in f]
        token_list = list(token_list)
        vocab_size = len(token_list)
        compute_dtype = _get(args, "compute_dtype", "bfloat16")

        # 1. frontend
        input_size = _get(args, "input_size")
        if input_size is None:
            fcls = _choice(frontend_choices, "frontend", _get(args, "frontend", "default"))
            frontend = fcls(**(_get(args, "frontend_conf") or {}))
            input_size = frontend.output_size()
        else:
            raise NotImplementedError("pre-extracted features (input_size != None)")
        # 2. specaug: training only (espnet_model.py:394)  3. normalize
        ncls = _choice(normalize_choices, "normalize", _get(args, "normalize", "utterance_mvn"))
        normalize = ncls(**(_get(args, "normalize_conf") or {})) if ncls is not None else None
        if _get(args, "preencoder") is not None or _get(args, "postencoder") is not None:
            raise NotImplementedError("preencoder/postencoder")
        # 4. encoder
        ecls = _choice(encoder_choices, "encoder", _get(args, "encoder", "conformer"))
        encoder = ecls(input_size=input_size, compute_dtype=compute_dtype,
                       **(_get(args, "encoder_conf") or {}))
        # 5. decoder
        dname = _get(args, "decoder", "transformer")
        decoder = None
        if dname is not None:
            dcls = _choice(decoder_choices, "decoder", dname)
            decoder = dcls(vocab_size=vocab_size, encoder_output_size=encoder.output_size(),
                           compute_dtype=compute_dtype, **(_get(args, "decoder_conf") or {}))
        # 6. CTC
        ctc = CTC(odim=vocab_size, encoder_output_size=encoder.output_size(),
                  compute_dtype=compute_dtype, **(_get(args, "ctc_conf") or {}))
        # 7. model
        mcls = _choice(model_choices, "model", _get(args, "model", "espnet"))
        model = mcls(vocab_size=vocab_size, frontend=frontend, specaug=None, normalize=normalize,
                     preencoder=None, encoder=encoder, postencoder=None, decoder=decoder, ctc=ctc,
                     joint_network=None, token_list=token_list, **(_get(args, "model_conf") or {}))
        return model

    @classmethod
    def build_model_from_file(cls, config_file: Union[Path, str, None] = None,
                              model_file: Union[Path, str, None] = None, device: str = "cuda",
                              compute_dtype: Optional[str] = None):
        """espnet2/tasks/abs_task.py:2456-2561.  Returns (model, args)."""
        if config_file is None:
            assert model_file is not None
            config_file = Path(model_file).parent / "config.yaml"
        with Path(config_file).open("r", encoding="utf-8") as f:
            args = argparse.Namespace(**yaml.load(f, Loader=getattr(yaml, "CSafeLoader", yaml.SafeLoader)))
        if compute_dtype is not None:
            args.compute_dtype = compute_dtype
        model = cls.build_model(args)
        if model_file is not None:
            state = torch.load(model_file, map_location="cpu", weights_only=False)
            for key in ("module", "state_dict"):
                if isinstance(state, dict) and key in state and isinstance(state[key], dict):
                    state = state[key]
            model.load_state_dict(state, strict=False)
        model.to(device)
        model.eval()
        return model, args

    # ------------------------------------------------------------------ decode-set iterator
    @classmethod
    def required_data_names(cls, train: bool = True, inference: bool = False):
        """espnet2/tasks/asr.py:437-447."""
        return ("speech",) if inference else ("speech", "text")

    @classmethod
    def build_preprocess_fn(cls, args, train: bool):
        """Inference-time subset of `CommonPreprocessor` (espnet2/train/preprocessor.py:456-482 with
        train=False: no RIR/noise; `speech_volume_normalize` and channel averaging still apply)."""
        if train:
            raise NotImplementedError("training-time preprocessing is outside the MI355X hot path")
        if not _get(args, "use_preprocessor", True):
            return None
        pc = _get(args, "preprocessor_conf") or {}
        vol = pc.get("speech_volume_normalize", _get(args, "speech_volume_normalize"))
        single = bool(pc.get("force_single_channel", False))

        def preprocess(uid, data):
            import numpy as np

            sp = data.get("speech")
            if sp is not None:
                if vol is not None:
                    ma = np.max(np.abs(sp))
                    if ma != 0:
                        sp = sp * vol / ma
                if single and sp.ndim == 2:
                    sp = np.mean(sp, axis=1)
                data["speech"] = sp
            return data

        # 1-D float samples pass through untouched: the native batch reader may skip the call
        preprocess.keeps_mono_samples = vol is None
        return preprocess

    @classmethod
    def build_collate_fn(cls, args, train: bool):
        """espnet2/tasks/asr.py:418-426: CommonCollateFn(float_pad_value=0.0, int_pad_value=-1)."""
        from functools import partial

        from espnet_amd.train.iterable_dataset import common_collate_fn

        return partial(common_collate_fn, float_pad_value=0.0, int_pad_value=-1)

    @classmethod
    def build_streaming_iterator(cls, data_path_and_name_and_type, preprocess_fn=None, collate_fn=None,
                                 key_file: Optional[str] = None, batch_size: int = 1, dtype="float32",
                                 num_workers: int = 1, allow_variable_data_keys: bool = False, ngpu: int = 0,
                                 inference: bool = False, bucket_window: int = 8, window_claim=None):
        """espnet2/tasks/abs_task.py:2403-2451, with utterance batches (batch_size > 1) cut from a
        length-sorted read-ahead window; `.key_order` on the result gives the original order.  `window_claim`
        (`--ngpu N`): a callable window index -> bool; only claimed windows are read and decoded by this process
        (espnet_amd.distributed.WindowClaimer: dynamic dispatch over a shared counter)."""
        from espnet_amd.train.iterable_dataset import (IterableESPnetDataset, StreamingBatchIterator,
                                                       common_collate_fn)

        if dtype in ("bfloat16", "float16"):
            dtype = "float32"  # host-side sample dtype; the MFMA mode is Speech2Text's business
        ds = IterableESPnetDataset(data_path_and_name_and_type, preprocess=preprocess_fn, float_dtype=dtype,
                                   key_file=key_file)
        for k in cls.required_data_names(False, inference):  # abs_task.py check_task_requirements
            if not ds.has_name(k):
                raise RuntimeError(f'"{k}" is required for {cls.__name__}. (--data_path_and_name_and_type)')
        if not allow_variable_data_keys:
            extra = set(ds.names()) - set(cls.required_data_names(False, inference)) - {"text"}
            if extra:
                raise RuntimeError(f"The data-name must be one of ('speech', 'text'): {sorted(extra)}")
        return StreamingBatchIterator(ds, batch_size=batch_size, bucket_window=bucket_window,
                                      num_workers=num_workers, collate_fn=collate_fn or common_collate_fn,
                                      length_key="speech", pin_memory=ngpu > 0, window_claim=window_claim)
