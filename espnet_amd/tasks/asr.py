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
from espnet_amd.asr.espnet_model import ESPnetASRModel
from espnet_amd.asr.frontend.default import DefaultFrontend
from espnet_amd.layers.global_mvn import GlobalMVN
from espnet_amd.layers.utterance_mvn import UtteranceMVN

frontend_choices = {"default": DefaultFrontend}
normalize_choices = {"utterance_mvn": UtteranceMVN, "global_mvn": GlobalMVN}
encoder_choices = {"conformer": ConformerEncoder,
                   "contextual_block_conformer": ContextualBlockConformerEncoder}
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
            args = argparse.Namespace(**yaml.safe_load(f))
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
