"""LMTask: registry boundary for the language-model scorer (espnet2/tasks/lm.py:25-47, 190-215 and
espnet2/tasks/abs_task.py:2456-2561 `build_model_from_file`): the yaml written by the reference's
`LMTask.main --dry_run` / LM training names `lm` + `lm_conf`; the reference's `.pth` loads
unchanged (keys `lm.embed.*`, `lm.encoder.*`, `lm.decoder.*`)."""
import argparse
from pathlib import Path
from typing import Optional, Union

import torch
import yaml

from espnet_amd.lm.seq_rnn_lm import SequentialRNNLM
from espnet_amd.lm.transformer_lm import ESPnetLanguageModel, TransformerLM

lm_choices = {"seq_rnn": SequentialRNNLM, "transformer": TransformerLM}  # espnet2/tasks/lm.py:36-44


class LMTask:
    @classmethod
    def build_model(cls, args) -> ESPnetLanguageModel:
        if isinstance(args, dict):
            args = argparse.Namespace(**args)
        token_list = args.token_list
        if isinstance(token_list, str):
            with open(token_list, encoding="utf-8") as f:
                token_list = [line.rstrip("\n") for line in f]
        name = getattr(args, "lm", "seq_rnn")
        if name not in lm_choices:
            raise NotImplementedError(f"lm={name!r} is outside the MI355X hot path (supported: {sorted(lm_choices)})")
        lm = lm_choices[name](vocab_size=len(token_list), compute_dtype=getattr(args, "compute_dtype", "bfloat16"),
                              **(getattr(args, "lm_conf", None) or {}))
        return ESPnetLanguageModel(lm=lm, vocab_size=len(token_list), **(getattr(args, "model_conf", None) or {}))

    @classmethod
    def build_model_from_file(cls, config_file: Union[Path, str], model_file: Union[Path, str, None] = None,
                              device: str = "cuda", compute_dtype: Optional[str] = None):
        with Path(config_file).open("r", encoding="utf-8") as f:
            args = argparse.Namespace(**yaml.load(f, Loader=getattr(yaml, "CSafeLoader", yaml.SafeLoader)))
        if compute_dtype is not None:
            args.compute_dtype = compute_dtype
        model = cls.build_model(args)
        if model_file is not None:
            state = torch.load(model_file, map_location="cpu", weights_only=False)
            model.load_state_dict(state, strict=False)
            model.lm.invalidate()
        model.to(device).eval()
        return model, args
