"""Build-container check of the reference-side binding.  Needs the
reference tree (/root/reference; absent on the GPU box -> skipped there): every adapter of
`espnet_amd.integration.espnet2_adapters` instantiates against the reference's ABCs and registers in its
`ClassChoices` tables (espnet2/train/class_choices.py:46-49), the reference's own `BatchBeamSearch` accepts the
accelerated scorers (legacy/nets/beam_search.py:80-96), and tests/scorer_driver.py -- the search flow the GPU
tests drive the scorers with -- reproduces the reference's n-best fixtures when it drives the REFERENCE's scorers."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(not Path("/root/reference/espnet2").is_dir(), reason="reference tree not present")
def test_adapters_bind_to_reference_and_driver_equals_reference_search():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, str(REPO / "tests" / "reference_binding_check.py")], capture_output=True,
                       text=True, timeout=900, env=env, cwd=str(REPO))
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    assert len(out["instantiated"]) == 11
    assert out["registered"] == {
        "encoder": ["mi355x_branchformer", "mi355x_conformer", "mi355x_contextual_block_conformer",
                    "mi355x_e_branchformer"],
        "frontend": ["mi355x_default"], "normalize": ["mi355x_global_mvn", "mi355x_utterance_mvn"],
        "decoder": ["mi355x_transformer"], "lm": ["mi355x_seq_rnn", "mi355x_transformer"]}
    assert out["defaults_build_stock"] and out["fast_path_is_adapter"]
    assert out["deepcopy_ok"]  # deepcopy / pickle of the adapter encoder (cls.__new__(cls) with no arguments)
    assert out["unsupported_of_defaults"] == ["macaron_style=False"]
    assert out["get_class"] == "MI355XTransformerDecoder"
    assert out["ref_search_full"] == ["decoder", "length_bonus", "lm"] and out["ref_search_part"] == ["ctc"]
    assert out["ref_search_nn"] == ["decoder", "lm"]
    assert out["all_scorer_interface"] and out["ctc_is_partial"]
    for name, res in out["driver_vs_reference_fixture"].items():
        assert res["tokens_equal"] and res["max_score_err"] < 1e-4, (name, res)
