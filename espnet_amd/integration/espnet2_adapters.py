"""Reference-side binding: the adapters a maintainer of espnet/espnet adds to run this path.

Importing this module needs `espnet2` importable (it is the reference-side half of the boundary and is not used
by espnet_amd itself).  Every adapter inherits the MI355X implementation AND the reference's abstract base
class, so `ClassChoices.__init__`'s subclass check (espnet2/train/class_choices.py:46-49) and `BeamSearch.__init__`'s
`isinstance(v, ScorerInterface)` / `PartialScorerInterface` dispatch (espnet2/legacy/nets/beam_search.py:80-96)
accept them.  `register()` adds them to the task registries (espnet2/tasks/asr.py:96-217, espnet2/tasks/lm.py)
under `mi355x_*` names; state-dict keys are the stock ones, so reference checkpoints load with strict=True.

Checked in the build container by tests/test_cpu_reference_binding.py (every adapter instantiates, registers,
and the reference's own BatchBeamSearch accepts the accelerated scorers)."""
from espnet2.asr.decoder.abs_decoder import AbsDecoder
from espnet2.asr.encoder.abs_encoder import AbsEncoder
from espnet2.asr.frontend.abs_frontend import AbsFrontend
from espnet2.layers.abs_normalize import AbsNormalize
from espnet2.legacy.nets.scorer_interface import BatchPartialScorerInterface, BatchScorerInterface
from espnet2.lm.abs_model import AbsLM

from espnet_amd.asr.decoder.transformer_decoder import TransformerDecoder as _TransformerDecoder
from espnet_amd.asr.encoder.conformer_encoder import ConformerEncoder as _ConformerEncoder
from espnet_amd.asr.encoder.contextual_block_conformer_encoder import (
    ContextualBlockConformerEncoder as _ContextualBlockConformerEncoder,
)
from espnet_amd.asr.encoder.e_branchformer_encoder import BranchformerEncoder as _BranchformerEncoder
from espnet_amd.asr.encoder.e_branchformer_encoder import EBranchformerEncoder as _EBranchformerEncoder
from espnet_amd.asr.frontend.default import DefaultFrontend as _DefaultFrontend
from espnet_amd.layers.global_mvn import GlobalMVN as _GlobalMVN
from espnet_amd.layers.utterance_mvn import UtteranceMVN as _UtteranceMVN
from espnet_amd.lm.seq_rnn_lm import SequentialRNNLM as _SequentialRNNLM
from espnet_amd.lm.transformer_lm import TransformerLM as _TransformerLM
from espnet_amd.nets.scorers.ctc import CTCPrefixScorer as _CTCPrefixScorer
from espnet_amd.nets.scorers.length_bonus import LengthBonus as _LengthBonus


class MI355XConformerEncoder(_ConformerEncoder, AbsEncoder):
    """forward(xs_pad (B,L,D), ilens (B,), prev_states=None) -> (ys (B,L',d), olens, None);
    keywords of espnet2/asr/encoder/conformer_encoder.py:89-121.

    Option combinations the MI355X kernels do not cover (the reference's own defaults among them:
    `macaron_style=False`, conformer_encoder.py:103) do not fail at `build_model`: the same yaml name then builds the
    STOCK espnet2 `ConformerEncoder` with the same arguments, and says so in the log (never silently: the stock
    module is the reference's code path, not an accelerated one)."""

    def __new__(cls, *args, **kwargs):
        if not args and not kwargs:  # copy.deepcopy / pickle re-create a Module through cls.__new__(cls) alone
            return super().__new__(cls)
        bad = _ConformerEncoder.unsupported_options(*args, **kwargs)
        if bad:
            import logging

            from espnet2.asr.encoder.conformer_encoder import ConformerEncoder as _Stock

            kwargs.pop("compute_dtype", None)  # the one keyword the stock class does not know
            logging.warning("mi355x_conformer: %s outside the MI355X fast path -> stock espnet2 ConformerEncoder "
                            "(reference code path, not accelerated)", ", ".join(bad))
            return _Stock(*args, **kwargs)  # not an instance of cls: Python does not call cls.__init__ on it
        return super().__new__(cls)


class MI355XEBranchformerEncoder(_EBranchformerEncoder, AbsEncoder):
    """espnet2/asr/encoder/e_branchformer_encoder.py:185-520."""


class MI355XBranchformerEncoder(_BranchformerEncoder, AbsEncoder):
    """espnet2/asr/encoder/branchformer_encoder.py:293-620."""


class MI355XContextualBlockConformerEncoder(_ContextualBlockConformerEncoder, AbsEncoder):
    """espnet2/asr/encoder/contextual_block_conformer_encoder.py:34-600 (forward_infer streaming path)."""


class MI355XDefaultFrontend(_DefaultFrontend, AbsFrontend):
    """espnet2/asr/frontend/default.py:22-171."""


class MI355XGlobalMVN(_GlobalMVN, AbsNormalize):
    """espnet2/layers/global_mvn.py:13-122."""


class MI355XUtteranceMVN(_UtteranceMVN, AbsNormalize):
    """espnet2/layers/utterance_mvn.py:10-88."""


class MI355XTransformerDecoder(_TransformerDecoder, AbsDecoder, BatchScorerInterface):
    """espnet2/asr/decoder/transformer_decoder.py:393-468; scorer interface :191-311."""


class MI355XTransformerLM(_TransformerLM, AbsLM):
    """espnet2/lm/transformer_lm.py:12-137."""


class MI355XSequentialRNNLM(_SequentialRNNLM, AbsLM):
    """espnet2/lm/seq_rnn_lm.py:14-177."""


class MI355XCTCPrefixScorer(_CTCPrefixScorer, BatchPartialScorerInterface):
    """espnet2/legacy/nets/scorers/ctc.py:10-157."""


class MI355XLengthBonus(_LengthBonus, BatchScorerInterface):
    """espnet2/legacy/nets/scorers/length_bonus.py:10-58."""


ADAPTERS = {
    "encoder": {"mi355x_conformer": MI355XConformerEncoder, "mi355x_e_branchformer": MI355XEBranchformerEncoder,
                "mi355x_branchformer": MI355XBranchformerEncoder,
                "mi355x_contextual_block_conformer": MI355XContextualBlockConformerEncoder},
    "frontend": {"mi355x_default": MI355XDefaultFrontend},
    "normalize": {"mi355x_global_mvn": MI355XGlobalMVN, "mi355x_utterance_mvn": MI355XUtteranceMVN},
    "decoder": {"mi355x_transformer": MI355XTransformerDecoder},
    "lm": {"mi355x_transformer": MI355XTransformerLM, "mi355x_seq_rnn": MI355XSequentialRNNLM},
}


def register():
    """Add the adapters to the reference's ClassChoices tables.  Each table's own `base_type` (its type_check) is applied, as
    `ClassChoices.__init__` does for its initial classes (class_choices.py:46-49)."""
    from espnet2.tasks import asr as asr_task
    from espnet2.tasks import lm as lm_task

    tables = {"encoder": asr_task.encoder_choices, "frontend": asr_task.frontend_choices,
              "normalize": asr_task.normalize_choices, "decoder": asr_task.decoder_choices,
              "lm": lm_task.lm_choices}
    for kind, classes in ADAPTERS.items():
        table = tables[kind]
        for name, cls in classes.items():
            if table.base_type is not None and not issubclass(cls, table.base_type):
                raise ValueError(f"must be {table.base_type.__name__}, but got {cls}")
            table.classes[name] = cls
    return tables
