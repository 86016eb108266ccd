"""Run by tests/test_cpu_reference_binding.py in a subprocess (it puts /root/reference and the import shims on
sys.path, which must not leak into the test session).  Prints one JSON line of results."""
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.dont_write_bytecode = True
sys.path[:0] = [str(REPO / "tests" / "golden" / "_shims"), "/root/reference", str(REPO)]

import torch  # noqa: E402

torch.set_grad_enabled(False)
out = {}

# ---- 1. every INTEGRATION.md adapter instantiates against the reference ABCs and registers -----------------
from espnet_amd.integration import espnet2_adapters as A  # noqa: E402

V, d = 50, 64
inst = {
    # the Conformer recipes' options (macaron FFN pair, rel-pos attention): what the fast path covers
    "MI355XConformerEncoder": A.MI355XConformerEncoder(80, output_size=d, attention_heads=1, linear_units=128,
                                                       num_blocks=1, macaron_style=True, rel_pos_type="latest"),
    "MI355XEBranchformerEncoder": A.MI355XEBranchformerEncoder(80, output_size=d, attention_heads=1, linear_units=128,
                                                               num_blocks=1, cgmlp_linear_units=128, rel_pos_type="latest",
                                                               use_ffn=True, macaron_ffn=True),
    "MI355XBranchformerEncoder": A.MI355XBranchformerEncoder(80, output_size=d, attention_heads=1, num_blocks=1,
                                                             cgmlp_linear_units=128, rel_pos_type="latest"),
    "MI355XContextualBlockConformerEncoder": A.MI355XContextualBlockConformerEncoder(
        80, output_size=d, attention_heads=1, linear_units=128, num_blocks=1, macaron_style=True),
    "MI355XDefaultFrontend": A.MI355XDefaultFrontend(),
    "MI355XUtteranceMVN": A.MI355XUtteranceMVN(),
    "MI355XTransformerDecoder": A.MI355XTransformerDecoder(V, d, attention_heads=1, linear_units=128, num_blocks=1),
    "MI355XTransformerLM": A.MI355XTransformerLM(V, att_unit=64, head=2, unit=128, layer=1, embed_unit=64),
    "MI355XSequentialRNNLM": A.MI355XSequentialRNNLM(V, unit=64, nlayers=1),
    "MI355XLengthBonus": A.MI355XLengthBonus(V),
}
from espnet2.asr.ctc import CTC as RefCTC  # noqa: E402
from espnet_amd.asr.ctc import CTC  # noqa: E402

inst["MI355XCTCPrefixScorer"] = A.MI355XCTCPrefixScorer(CTC(V, d), V - 1)
out["instantiated"] = sorted(inst)
# the reference's OWN defaults (macaron_style=False, rel_pos_type="legacy": conformer_encoder.py:89-121) are outside
# the fast path: the same name must then hand back the stock class instead of failing at build_model
from espnet2.asr.encoder.conformer_encoder import ConformerEncoder as RefConformer  # noqa: E402

stock = A.MI355XConformerEncoder(80, output_size=d, attention_heads=1, linear_units=128, num_blocks=1)
out["defaults_build_stock"] = type(stock) is RefConformer and not isinstance(stock, A.MI355XConformerEncoder)
out["fast_path_is_adapter"] = type(inst["MI355XConformerEncoder"]) is A.MI355XConformerEncoder
out["unsupported_of_defaults"] = A.MI355XConformerEncoder.unsupported_options(80)
# copy.deepcopy / pickle re-create a Module through cls.__new__(cls) with no arguments: the adapter's
# stock-class fall-back in __new__ must not break them (espnet2's quantize_dynamic path, EMA copies, torch.save(model))
import copy  # noqa: E402
import pickle  # noqa: E402

enc0 = inst["MI355XConformerEncoder"]
dup = copy.deepcopy(enc0)
rt = pickle.loads(pickle.dumps(enc0))
out["deepcopy_ok"] = all(type(m) is A.MI355XConformerEncoder and
                         all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), enc0.state_dict().values()))
                         for m in (dup, rt))
tables = A.register()
out["registered"] = {k: sorted(n for n in t.classes if n.startswith("mi355x_")) for k, t in tables.items()}
out["get_class"] = tables["decoder"].get_class("mi355x_transformer").__name__

# ---- 2. the reference's own BatchBeamSearch accepts the accelerated scorers ---------------------------------
from espnet2.legacy.nets.batch_beam_search import BatchBeamSearch  # noqa: E402
from espnet2.legacy.nets.scorer_interface import PartialScorerInterface, ScorerInterface  # noqa: E402

scorers = dict(decoder=inst["MI355XTransformerDecoder"], ctc=inst["MI355XCTCPrefixScorer"],
               length_bonus=inst["MI355XLengthBonus"], lm=inst["MI355XTransformerLM"])
bs = BatchBeamSearch(beam_size=5, weights=dict(decoder=0.7, ctc=0.3, length_bonus=0.1, lm=0.2), scorers=scorers,
                     sos=V - 1, eos=V - 1, vocab_size=V, token_list=None, pre_beam_score_key="full")
out["ref_search_full"] = sorted(bs.full_scorers)
out["ref_search_part"] = sorted(bs.part_scorers)
out["ref_search_nn"] = sorted(bs.nn_dict.keys())
out["all_scorer_interface"] = all(isinstance(v, ScorerInterface) for v in scorers.values())
out["ctc_is_partial"] = isinstance(scorers["ctc"], PartialScorerInterface)

# ---- 3. the restated driver (tests/scorer_driver.py) == the reference search, on the reference's own scorers -
import numpy as np  # noqa: E402

from espnet2.asr.decoder.transformer_decoder import TransformerDecoder as RefDecoder  # noqa: E402
from espnet2.legacy.nets.scorers.ctc import CTCPrefixScorer as RefCTCPrefixScorer  # noqa: E402
from espnet2.legacy.nets.scorers.length_bonus import LengthBonus as RefLengthBonus  # noqa: E402
from tests.helpers import golden_state_dict, load_golden  # noqa: E402
from tests.scorer_driver import drive_search  # noqa: E402

res = {}
for name in ("tiny_beam5", "tiny_beam4_early_eos", "tiny_beam4_minlen"):
    g = load_golden(name)
    sd = golden_state_dict(g)
    Vg = int(g["vocab"])
    dc = g["config"]["decoder_conf"]
    dg = g["config"]["encoder_conf"]["output_size"]
    dec = RefDecoder(Vg, dg, **dc)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}, strict=True)
    dec.eval()
    ctc = RefCTC(Vg, dg)
    ctc.load_state_dict({k[len("ctc."):]: v for k, v in sd.items() if k.startswith("ctc.")}, strict=True)
    x = torch.from_numpy(np.asarray(g["enc_out"])).float()
    cw = float(g["ctc_weight"])
    kw = {k: float(g[k]) for k in ("maxlenratio", "minlenratio") if k in g}
    nbest = drive_search(dict(decoder=dec, ctc=RefCTCPrefixScorer(ctc=ctc, eos=Vg - 1), length_bonus=RefLengthBonus(Vg)),
                         dict(decoder=1.0 - cw, ctc=cw, length_bonus=float(g["penalty"]) if "penalty" in g else 0.0),
                         int(g["beam"]), Vg, Vg - 1, Vg - 1, x, pre_beam_score_key=None if cw == 1.0 else "full", **kw)
    n = len(g["yseq_lens"])
    ok = len(nbest) >= n
    err = 0.0
    for k in range(n if ok else 0):
        ref = g["yseq"][k, : g["yseq_lens"][k]].tolist()
        ok = ok and nbest[k]["yseq"] == ref
        err = max(err, abs(nbest[k]["score"] - float(g["score"][k])))
    res[name] = dict(tokens_equal=bool(ok), max_score_err=err, n=n)
out["driver_vs_reference_fixture"] = res
print("RESULT " + json.dumps(out))
