"""GPU parity of the device-resident beam search — all through the C-ABI.

Layers of evidence:
  1. search only, f32 mode, encoder output supplied by the oracle: the n-best list must contain the
     reference's `Speech2Text` hypotheses (tests/golden/*beam*.npz, small_g2_3s.npz) with identical
     token sequences and scores within an fp32 tolerance written below;
  2. end to end (HIP frontend + encoder + search) on the large model;
  3. bf16 mode (the mode configs[2]/[3] are timed in): every hypothesis the device returns is re-scored
     teacher-forced under the oracle's scorers and must carry that score to BF16_EPS per token and scorer
     (`bf16_rescore_check`); on the PEAKED fixture (heads fitted to a transcript, n-best 0.5 apart) the
     bf16 search must return the reference's n-best token sequences exactly, in order;
  4. batching is transparent: a ragged batch gives the same hypotheses as one utterance at a time;
  5. kernel-level checks of the decoder attention kernels against plain torch fp32.
"""
import json
import math
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import golden_speech, golden_state_dict, hparams, load_golden  # noqa: E402

SEARCH_CASES = ["tiny_beam5", "tiny_beam3_attn_only", "tiny_beam4_early_eos", "tiny_beam4_minlen",
                "small_g2_3s", "large_beam10_3s", "large_beam10_3s_peaked"]

# bf16 search against the oracle's scorers along the SAME token path (teacher-forced): |device - oracle| per scored
# token, per scorer.  Measured on MI355X (round 4, printed by every run): decoder 2.3e-4 and ctc 1.9e-3 on the peaked
# fixture (logits of +-12 through heads with row norms up to 7), 1e-4 at the bench's own size
# (tests/test_gpu_fullsize.py::test_beam10_b16_rows_bf16_vs_oracle); bounds ~2x the largest value seen.
BF16_EPS = {"decoder": 1e-3, "ctc": 4e-3, "lm": 1e-3}  # largest seen: 4.4e-4 (tiny_beam5_gru), 1.9e-3 (peaked), 2.5e-4


def bf16_rescore_check(tag, g, sd, enc_row, hyps, ctc_weight, lm_conf=None, eps=None):
    """Every device hypothesis re-scored teacher-forced under the oracle's scorers (tests/helpers.py::
    oracle_rescore_batch over `enc_row`, the encoder frames the device search was given).  Asserts the per-scorer
    bound and returns the measured per-token errors."""
    from tests.helpers import oracle_rescore_batch

    eps = eps or BF16_EPS
    dc = g["config"]["decoder_conf"]
    V = int(g["vocab"])
    ys = [h.yseq.tolist() for h in hyps]
    ref = oracle_rescore_batch(sd, enc_row, ys, dc["attention_heads"], dc["num_blocks"], ctc_weight, V - 1,
                               lm_conf=lm_conf)
    worst = {}
    for h, r in zip(hyps, ref):
        for k in h.scores:
            if k not in r:
                continue  # length_bonus: exact by construction
            e = abs(float(h.scores[k]) - r[k]) / max(r["n_scored"], 1)
            worst[k] = max(worst.get(k, 0.0), e)
    print(f"[{tag}] bf16 vs oracle, teacher-forced, per token: " +
          ", ".join(f"{k} {v:.2e} (bound {eps[k]:.0e})" for k, v in worst.items()) + f" over {len(hyps)} hypotheses")
    for k, v in worst.items():
        assert v <= eps[k], (tag, k, v)
    return worst, ref


def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def build_lm(g, dtype="float32", dev="cuda"):
    """TransformerLM of an LM golden case: recipe weights under the reference's own key names."""
    from espnet_amd.tasks.lm import lm_choices
    from oracle.weights import recipe_state_dict

    shapes = {"lm." + k: tuple(v) for k, v in json.loads(str(g["lm_state_shapes"])).items()}
    lsd = recipe_state_dict(shapes, int(g["wseed"]), skip=())
    cls = lm_choices[str(g["lm_name"]) if "lm_name" in g else "transformer"]
    lm = cls(int(g["vocab"]), compute_dtype=dtype, **json.loads(str(g["lm_conf"])))
    lm.load_state_dict(_sub(lsd, "lm."), strict=True)
    return lm.to(dev)


def build_search(g, sd, dtype="float32", dev="cuda", lm=None):
    """decoder + CTC head + BatchBeamSearch for a golden case (no encoder needed)."""
    from espnet_amd.asr.ctc import CTC
    from espnet_amd.asr.decoder.transformer_decoder import TransformerDecoder
    from espnet_amd.nets.batch_beam_search import build_beam_search
    from oracle.weights import token_list

    V = int(g["vocab"])
    d = g["config"]["encoder_conf"]["output_size"]
    dc = {k: v for k, v in g["config"]["decoder_conf"].items()}
    dec = TransformerDecoder(V, d, compute_dtype=dtype, **dc)
    dec.load_state_dict(_sub(sd, "decoder."), strict=True)
    ctc = CTC(V, d, compute_dtype=dtype)
    ctc.load_state_dict(_sub(sd, "ctc."), strict=True)
    dec.to(dev)
    ctc.to(dev)
    cw = float(g["ctc_weight"])
    model = types.SimpleNamespace(decoder=dec if cw < 1.0 else None, ctc=ctc if cw > 0.0 else None,
                                  sos=V - 1, eos=V - 1)
    return build_beam_search(model, beam_size=int(g["beam"]), ctc_weight=cw,
                             penalty=float(g["penalty"]) if "penalty" in g else 0.0,
                             lm_weight=float(g["lm_weight"]) if lm is not None else 0.0,
                             token_list=token_list(V), lm=lm)


def oracle_enc(g, sd):
    from oracle import conformer as oc

    hp = hparams(g)
    speech, lens = golden_speech(g)
    with torch.no_grad():
        enc, olens = oc.encode(sd, speech, lens, hp["heads"], hp["num_blocks"], hp["n_fft"],
                               hp["win_length"], hp["hop"])
    return enc, olens


def check_against_golden(g, hyps, tol_abs, tol_rel, require_all=True, report=None):
    keys = json.loads(str(g["score_keys"]))
    mine = {tuple(h.yseq.tolist()): h for h in hyps}
    n = len(g["yseq_lens"])
    found = 0
    for k in range(n):
        ref = tuple(g["yseq"][k, : g["yseq_lens"][k]].tolist())
        tol = tol_abs + tol_rel * abs(float(g["score"][k]))
        if ref not in mine:
            assert not require_all, f"reference hypothesis #{k} missing from the device n-best"
            if report is not None:
                report.append((k, float(g["score"][k])))
            continue
        found += 1
        h = mine[ref]
        assert abs(float(h.score) - float(g["score"][k])) < tol, (k, float(h.score), float(g["score"][k]))
        for j, kk in enumerate(keys):
            assert abs(float(h.scores[kk]) - float(g["scores"][k, j])) < tol + tol_rel * abs(float(g["scores"][k, j]))
    return found


def explain_nbest_difference(g, sd, hyps, missing, tol_abs):
    """The end-to-end n-best may differ from the reference's where the encoder's 1e-4-level activation
    differences flip a beam-pruning near-tie.  Instead of waving that through, every difference is
    accounted for: (1) each hypothesis the device returned that the reference list does not hold is
    re-scored by the oracle's scorers along its own token path over the ORACLE's encoder output and must
    carry exactly that score (so it is a legitimate, correctly scored hypothesis of the reference search
    space); (2) the reference hypotheses that are missing are listed with their reference rank and score,
    and none of them may beat the device's best (the top of the list is never lost)."""
    from tests.helpers import oracle_rescore

    enc, olens = oracle_enc(g, sd)
    e = enc[0, : int(olens[0])]
    dc = g["config"]["decoder_conf"]
    V = int(g["vocab"])
    refs = {tuple(g["yseq"][k, : g["yseq_lens"][k]].tolist()) for k in range(len(g["yseq_lens"]))}
    extra = [h for h in hyps if tuple(h.yseq.tolist()) not in refs]
    for h in extra:
        r = oracle_rescore(sd, e, h.yseq.tolist(), dc["attention_heads"], dc["num_blocks"],
                           float(g["ctc_weight"]), V - 1)
        assert abs(r["score"] - float(h.score)) < tol_abs + 5e-5 * abs(r["score"]), (r, float(h.score))
        assert abs(r["decoder"] - float(h.scores["decoder"])) < 2 * tol_abs
        assert abs(r["ctc"] - float(h.scores["ctc"])) < 2 * tol_abs
    best = float(hyps[0].score)
    for k, sc in missing:
        print(f"reference hypothesis #{k} (score {sc:.4f}) not in the device n-best; device best {best:.4f}, "
              f"device worst {float(hyps[-1].score):.4f}")
        assert k > 0 and sc <= best + tol_abs
    print(f"{len(missing)} reference hypotheses missing, {len(extra)} device-only hypotheses re-scored by the "
          f"oracle and confirmed")


@pytest.mark.parametrize("name", SEARCH_CASES)
def test_search_f32_matches_reference_nbest(name):
    g = load_golden(name)
    sd = golden_state_dict(g)
    enc, olens = oracle_enc(g, sd)
    bs = build_search(g, sd, "float32")
    kw = {k: float(g[k]) for k in ("maxlenratio", "minlenratio") if k in g}
    hyps = bs.search_batch(enc.cuda(), [int(olens[0])], **kw)[0]
    # identical token sequences; scores: fp32 round-off of a different summation order,
    # accumulated over up to T steps
    check_against_golden(g, hyps, tol_abs=2e-3, tol_rel=2e-5)
    # ranking: my best is the reference's best unless the reference's own top-2 gap is round-off
    if len(g["score"]) > 1 and float(g["score"][0] - g["score"][1]) > 1e-2:
        assert hyps[0].yseq.tolist() == g["yseq"][0, : g["yseq_lens"][0]].tolist()
    # the ended list is sorted
    sc = [float(h.score) for h in hyps]
    assert sc == sorted(sc, reverse=True)


@pytest.mark.parametrize("name", ["tiny_beam5_lm", "tiny_beam4_lm_posenc", "tiny_beam60_lm_v300",
                                  "tiny_beam5_rnnlm", "tiny_beam4_rnnlm_nhid", "tiny_beam5_gru", "tiny_beam4_gru_nhid",
                                  "tiny_beam4_rnn_tanh", "tiny_beam4_rnn_relu"])
@pytest.mark.parametrize("graph", [False, True])
def test_search_with_lm_scorer_f32_matches_reference_nbest(name, graph):
    """decoder + CTC prefix + TransformerLM scorers fused in the device search;
    the reference's n-best (token sequences, total and per-scorer scores incl. "lm") must be in mine."""
    g = load_golden(name)
    sd = golden_state_dict(g)
    enc, olens = oracle_enc(g, sd)
    bs = build_search(g, sd, "float32", lm=build_lm(g, "float32"))
    bs.use_hipgraph = graph
    assert "lm" in json.loads(str(g["score_keys"]))
    for _ in range(2 if graph else 1):  # second call replays the captured graph
        hyps = bs.search_batch(enc.cuda(), [int(olens[0])])[0]
        check_against_golden(g, hyps, tol_abs=2e-3, tol_rel=2e-5)
        if float(g["score"][0] - g["score"][1]) > 1e-2:
            assert hyps[0].yseq.tolist() == g["yseq"][0, : g["yseq_lens"][0]].tolist()


@pytest.mark.parametrize("gname", ["tiny_beam4_lm_posenc", "tiny_beam5_rnnlm", "tiny_beam5_gru"])
def test_search_with_lm_scorer_bf16_and_batched(gname):
    """bf16 LM + decoder: score level vs the fp32 reference, additivity of the per-scorer scores, and
    utterance batching stays transparent with the LM cache / LSTM state in play."""
    g = load_golden(gname)
    sd = golden_state_dict(g)
    enc, olens = oracle_enc(g, sd)
    bs = build_search(g, sd, "bfloat16", lm=build_lm(g, "bfloat16"))
    T = int(olens[0])
    hyps = bs.search_batch(enc.cuda(), [T])[0]
    # every bf16 hypothesis carries the score the oracle's scorers (decoder, CTC, LM) give its token path
    from oracle.weights import recipe_state_dict

    sdl = dict(sd)
    sdl.update(recipe_state_dict({"lm." + k: tuple(v) for k, v in json.loads(str(g["lm_state_shapes"])).items()},
                                 int(g["wseed"]), skip=()))
    _, ref = bf16_rescore_check(gname, g, sdl, enc[0, :T], hyps, float(g["ctc_weight"]),
                                lm_conf=json.loads(str(g["lm_conf"])))
    # ... and bf16 pruning loses little: the oracle's score of the device's best is within BEST_LOSS of the
    # reference's best (joint score incl. the LM; the reference's own top-5 span more than that)
    w = bs.weights
    mine_best = max(w["decoder"] * r["decoder"] + w["ctc"] * r["ctc"] + w["lm"] * r["lm"] for r in ref)
    assert mine_best >= float(g["score"][0]) - 0.5, (mine_best, float(g["score"][0]))
    for h in hyps:
        tot = sum(bs.weights[k] * float(v) for k, v in h.scores.items())
        assert abs(tot - float(h.score)) < 1e-2 + 1e-4 * abs(tot)
    bs32 = build_search(g, sd, "float32", lm=build_lm(g, "float32"))
    e2 = torch.zeros(2, T, enc.shape[-1])
    e2[0], e2[1, : T - 9] = enc[0], enc[0, 9:]
    both = bs32.search_batch(e2.cuda(), [T, T - 9])
    for b, (x, n) in enumerate([(enc[:, :T], T), (enc[:, 9:T], T - 9)]):
        single = bs32.search_batch(x.contiguous().cuda(), [n])[0]
        assert [h.yseq.tolist() for h in single] == [h.yseq.tolist() for h in both[b]]
        for hs, hb in zip(single, both[b]):
            assert abs(float(hs.score) - float(hb.score)) < 1e-3
            assert abs(float(hs.scores["lm"]) - float(hb.scores["lm"])) < 1e-3


def test_search_batched_equals_single():
    """Utterance batching is transparent: B=3 ragged memories vs one at a time (f32)."""
    g = load_golden("tiny_beam4_early_eos")
    sd = golden_state_dict(g)
    bs = build_search(g, sd, "float32")
    torch.manual_seed(5)
    d = g["config"]["encoder_conf"]["output_size"]
    lens = [49, 31, 40]
    enc = torch.randn(3, max(lens), d) * 0.5
    for b, n in enumerate(lens):
        enc[b, n:] = 0.0
    batched = bs.search_batch(enc.cuda(), lens)
    for b, n in enumerate(lens):
        single = bs.search_batch(enc[b : b + 1, :n].contiguous().cuda(), [n])[0]
        assert len(single) == len(batched[b]) > 0
        for hs, hb in zip(single, batched[b]):
            assert hs.yseq.tolist() == hb.yseq.tolist()
            assert abs(float(hs.score) - float(hb.score)) < 1e-3


def test_search_fragment_major_memory_is_bit_identical(monkeypatch):
    """Round 6: from 96 rows the label step's source attention reads the memory's K / V^T from FRAGMENT-MAJOR copies
    (EmSearchBuffers.mem_kf / mem_vf, written by em_search_init; dec_pack_mem_frag_kernel) and linear_q from its fragment-major
    copy - the same arithmetic in the same order as the row-major launch.  A ragged batch of 12 memories x beam 10 (120 rows,
    lengths that are not multiples of 16 or 32, one of a single frame) must return the SAME hypotheses with the SAME scores,
    bit for bit, with the buffers (default) and without them (ESPNET_AMD_NO_MEM_FRAG=1, read when the buffers are allocated)."""
    g = load_golden("large_beam10_3s")
    sd = golden_state_dict(g)
    d = g["config"]["encoder_conf"]["output_size"]
    torch.manual_seed(23)
    lens = [74, 33, 17, 64, 1, 50, 47, 32, 15, 70, 9, 41]
    enc = torch.randn(len(lens), max(lens), d) * 0.5
    for b, n in enumerate(lens):
        enc[b, n:] = 0.0
    enc = enc.to(torch.bfloat16).cuda()
    with_frag = build_search(g, sd, "bfloat16")
    a = with_frag.search_batch(enc, lens)
    assert any("mem_kf" in t for t in with_frag._bufs.values()), "the fragment-major memory was not allocated"
    monkeypatch.setenv("ESPNET_AMD_NO_MEM_FRAG", "1")
    without = build_search(g, sd, "bfloat16")
    b_ = without.search_batch(enc, lens)
    assert not any("mem_kf" in t for t in without._bufs.values())
    for ha, hb in zip(a, b_):
        assert len(ha) == len(hb) > 0
        for x, y in zip(ha, hb):
            assert x.yseq.tolist() == y.yseq.tolist()
            assert float(x.score) == float(y.score)
            assert {k: float(v) for k, v in x.scores.items()} == {k: float(v) for k, v in y.scores.items()}


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_search_lanes_equal_searches_run_alone(dtype):
    """Round 6, SearchLanes: several joint searches in flight on as many HIP streams (each lane its own buffer set and
    hipGraph over the SAME scorers) return, bit for bit, what each search returns run alone - five different ragged
    batches dealt over two lanes and over three, more batches than lanes (a lane is reused), searches of different lengths
    in flight together (the short one finishes while the long one is mid-way); the same with a host thread per lane (two and four)."""
    from espnet_amd.nets.batch_beam_search import SearchLanes

    g = load_golden("tiny_beam4_early_eos")
    sd = golden_state_dict(g)
    d = g["config"]["encoder_conf"]["output_size"]
    torch.manual_seed(11)
    batches = []
    for lens in ([49, 31, 40], [12], [64, 64], [7, 55, 23, 41], [30, 9]):
        enc = torch.randn(len(lens), max(lens), d) * 0.5
        for b, n in enumerate(lens):
            enc[b, n:] = 0.0
        batches.append((enc.to(torch.bfloat16 if dtype == "bfloat16" else torch.float32).cuda(), lens))
    alone_bs = build_search(g, sd, dtype)
    alone = [alone_bs.search_batch(e, l) for e, l in batches]
    torch.cuda.synchronize()
    # (threaded: a host thread per lane runs the lane's search - the lanes' launches are then issued in parallel)
    for n_lanes, threaded in ((2, False), (3, False), (2, True), (4, True)):
        lanes = SearchLanes([build_search(g, sd, dtype) for _ in range(n_lanes)], torch.device("cuda"), threaded=threaded)
        todo, got, unit_of = list(range(len(batches))), {}, [None] * n_lanes
        while todo or any(u is not None for u in unit_of):
            for k in range(n_lanes):
                if unit_of[k] is None and todo:
                    u = todo.pop(0)
                    lanes.start(k, batches[u][0], batches[u][1], tag=u)
                    unit_of[k] = u
            for k in range(n_lanes):
                if unit_of[k] is not None:
                    r = lanes.poll(k)
                    if r is not None:
                        assert r[0] == unit_of[k]
                        got[r[0]] = r[1]
                        unit_of[k] = None
        for u, want in enumerate(alone):
            assert len(got[u]) == len(want)
            for hw, hg in zip(want, got[u]):
                assert len(hw) == len(hg) > 0
                for a, b in zip(hw, hg):
                    assert a.yseq.tolist() == b.yseq.tolist()
                    assert float(a.score) == float(b.score)
                    assert {k: float(v) for k, v in a.scores.items()} == {k: float(v) for k, v in b.scores.items()}
        lanes.close()


def test_search_structure_invariants():
    """Size-independent properties on a full-size run: every hypothesis starts with <sos>, ends
    with <eos>, has no <eos> inside (unless forced at maxlen), length <= maxlen + 2, and
    score == sum_k weight_k * scores_k."""
    g = load_golden("large_beam10_3s")
    sd = golden_state_dict(g)
    enc, olens = oracle_enc(g, sd)
    bs = build_search(g, sd, "bfloat16")
    hyps = bs.search_batch(enc.cuda(), [int(olens[0])])[0]
    V = int(g["vocab"])
    assert len(hyps) >= 1
    for h in hyps:
        y = h.yseq.tolist()
        assert y[0] == V - 1 and y[-1] == V - 1
        assert len(y) <= int(olens[0]) + 2
        tot = sum(bs.weights[k] * float(v) for k, v in h.scores.items())
        assert abs(tot - float(h.score)) < 1e-2 + 1e-4 * abs(tot)
    # bf16 vs the fp32 oracle along the device's own token paths: per-token, per-scorer bound ...
    _, ref = bf16_rescore_check("large_beam10_3s bf16", g, sd, enc[0, : int(olens[0])], hyps, float(g["ctc_weight"]))
    # ... and what bf16 pruning can lose: the oracle's score of the device's best hypothesis against the reference's
    # best (random-init posteriors are nearly flat: the reference's own 10-best span 0.5)
    assert max(r["score"] for r in ref) >= float(g["score"][0]) - 0.5


def test_search_bf16_peaked_returns_reference_nbest_exactly():
    """The timed mode on posteriors a trained model would give (tests/golden/make_golden.py::fit_peaked_search_heads:
    CTC head and decoder output layer fitted to a transcript with five confusable positions; the reference's ten best
    hypotheses are >= 0.54 apart and were qualified against N(0, 0.05^2) noise on every log-probability): the bf16 search
    must return the reference's n-best token sequences EXACTLY and in order, with and without hipGraph replay, scores
    within the per-token bf16 bound."""
    g = load_golden("large_beam10_3s_peaked")
    sd = golden_state_dict(g)
    enc, olens = oracle_enc(g, sd)
    T = int(olens[0])
    ref_nbest = [g["yseq"][k, : g["yseq_lens"][k]].tolist() for k in range(len(g["yseq_lens"]))]
    bs = build_search(g, sd, "bfloat16")
    for graph in (False, True, True):
        bs.use_hipgraph = graph
        hyps = bs.search_batch(enc.cuda(), [T])[0]
        assert [h.yseq.tolist() for h in hyps[: len(ref_nbest)]] == ref_nbest, graph
    worst, _ = bf16_rescore_check("large_beam10_3s_peaked bf16", g, sd, enc[0, :T], hyps, float(g["ctc_weight"]))
    keys = json.loads(str(g["score_keys"]))
    for k, h in enumerate(hyps[: len(ref_nbest)]):
        n_tok = len(ref_nbest[k]) - 1
        assert abs(float(h.score) - float(g["score"][k])) <= (BF16_EPS["decoder"] + BF16_EPS["ctc"]) * n_tok
        for j, kk in enumerate(keys):
            assert abs(float(h.scores[kk]) - float(g["scores"][k, j])) <= BF16_EPS[kk] * n_tok, (k, kk)
    gaps = [float(a.score) - float(b.score) for a, b in zip(hyps, hyps[1:10])]
    print(f"[peaked bf16] device n-best gaps min {min(gaps):.3f} (reference {float(g['nbest_min_gap']):.3f})")


def test_speech2text_bf16_peaked_end_to_end(tmp_path):
    """The same fixture through Speech2Text in bfloat16 - bf16 frontend-to-search, the heads were fitted on the
    reference's f32 encoder output: the 1-best tokens must be the reference's, and the n-best the reference's set."""
    from espnet_amd.bin.asr_inference import Speech2Text

    g = load_golden("large_beam10_3s_peaked")
    sd = golden_state_dict(g)
    cfg = tmp_path / "config.yaml"
    cfg.write_text(str(g["config_yaml"]))
    torch.save(sd, tmp_path / "model.pth")
    s2t = Speech2Text(asr_train_config=str(cfg), asr_model_file=str(tmp_path / "model.pth"), device="cuda",
                      dtype="bfloat16", beam_size=int(g["beam"]), ctc_weight=float(g["ctc_weight"]),
                      nbest=int(g["nbest"]), penalty=0.0, lm_weight=0.0)
    speech, _ = golden_speech(g)
    res = s2t(speech[0].numpy())
    assert res[0][2] == g["token_int_best"].tolist()
    ref_nbest = [g["yseq"][k, : g["yseq_lens"][k]].tolist() for k in range(len(g["yseq_lens"]))]
    mine = [r[3].yseq.tolist() for r in res]
    print(f"[peaked bf16 e2e] {sum(y in ref_nbest for y in mine)} of {len(ref_nbest)} reference hypotheses returned; "
          f"order equal: {mine == ref_nbest}")
    assert mine == ref_nbest


@pytest.mark.parametrize("name", ["large_beam10_3s", "large_beam10_10s"])
def test_speech2text_end_to_end_f32(name, tmp_path):
    """HIP frontend + encoder + device beam search behind the reference's Speech2Text API."""
    from espnet_amd.bin.asr_inference import Speech2Text

    g = load_golden(name)
    sd = golden_state_dict(g)
    cfg = tmp_path / "config.yaml"
    cfg.write_text(str(g["config_yaml"]))
    torch.save(sd, tmp_path / "model.pth")
    s2t = Speech2Text(asr_train_config=str(cfg), asr_model_file=str(tmp_path / "model.pth"),
                      device="cuda", dtype="float32", beam_size=int(g["beam"]),
                      ctc_weight=float(g["ctc_weight"]), nbest=int(g["nbest"]), penalty=0.0,
                      lm_weight=0.0)
    speech, _ = golden_speech(g)
    res = s2t(speech[0].numpy())
    assert len(res) == int(g["nbest"])
    text, token, token_int, hyp = res[0]
    assert text is None or isinstance(text, str)
    assert all(isinstance(t, str) for t in token) and all(isinstance(t, int) for t in token_int)
    hyps = [r[3] for r in res]
    # the encoder runs on the GPU here (1e-5-level activation differences): scores get a tolerance, but the
    # n-best list itself must be the reference's -- every reference hypothesis present (measured: 10 of 10 on
    # both fixtures), no device-only hypothesis left unexplained
    missing = []
    found = check_against_golden(g, hyps, tol_abs=2e-2, tol_rel=5e-5, require_all=True, report=missing)
    assert found == len(g["yseq_lens"])
    explain_nbest_difference(g, sd, hyps, missing, tol_abs=2e-2)
    assert token_int == g["token_int_best"].tolist()


def test_speech2text_with_lm_files_f32(tmp_path):
    """Speech2Text(lm_train_config, lm_file, lm_weight) like asr_inference.py:179-191: the LM yaml is the
    shape LMTask writes (lm / lm_conf / token_list), the checkpoint uses the reference keys (`lm.*`)."""
    import yaml

    from espnet_amd.bin.asr_inference import Speech2Text
    from oracle.weights import recipe_state_dict, token_list

    g = load_golden("e2e_beam5_lm")
    sd = golden_state_dict(g)
    V = int(g["vocab"])
    (tmp_path / "config.yaml").write_text(str(g["config_yaml"]))
    torch.save(sd, tmp_path / "model.pth")
    (tmp_path / "lm.yaml").write_text(yaml.safe_dump(dict(lm="transformer", lm_conf=json.loads(str(g["lm_conf"])),
                                                         token_list=token_list(V))))
    shapes = {"lm." + k: tuple(v) for k, v in json.loads(str(g["lm_state_shapes"])).items()}
    torch.save(recipe_state_dict(shapes, int(g["wseed"]), skip=()), tmp_path / "lm.pth")
    s2t = Speech2Text(asr_train_config=str(tmp_path / "config.yaml"), asr_model_file=str(tmp_path / "model.pth"),
                      lm_train_config=str(tmp_path / "lm.yaml"), lm_file=str(tmp_path / "lm.pth"),
                      device="cuda", dtype="float32", beam_size=int(g["beam"]), ctc_weight=float(g["ctc_weight"]),
                      lm_weight=float(g["lm_weight"]), nbest=int(g["nbest"]), penalty=0.0)
    speech, _ = golden_speech(g)
    res = s2t(speech[0].numpy())
    hyps = [r[3] for r in res]
    assert "lm" in hyps[0].scores
    found = check_against_golden(g, hyps, tol_abs=2e-2, tol_rel=5e-5, require_all=False)
    assert found >= int(0.7 * len(g["yseq_lens"])), found
    if float(g["score"][0] - g["score"][1]) > 5e-2:
        assert res[0][2] == g["token_int_best"].tolist()


# --------------------------------------------------------------------------- kernel level
@pytest.fixture(scope="module")
def lib():
    from espnet_amd import lib as L

    return L.load()


def _act(prec):
    from espnet_amd import lib as L

    return (L.EM_F32, torch.float32, 2e-4) if prec == "f32" else (L.EM_BF16, torch.bfloat16, 3e-2)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("heads,d,n", [(4, 256, 7), (2, 64, 7), (8, 512, 640), (8, 512, 160)])
def test_dec_self_attention(lib, prec, heads, d, n):
    """n = 640 (configs[3]'s per-GPU rows, heads x n > 2048): the launch with ONE wave per row (csrc/decoder.hip
    self_attn_launch, SA_SPLIT = 1), which only that shape reaches; n = 160: two waves per row."""
    from espnet_amd import lib as L

    em, dt, tol = _act(prec)
    torch.manual_seed(1)
    Lmax, pos = (40, 21) if n < 100 else (160, 133)  # (long prefixes: three 64-position batches per row)
    dk = d // heads
    qkv = torch.randn(n, 3 * d).to(dt).cuda()
    kc = torch.randn(Lmax, n, d).to(dt).cuda()
    vc = torch.randn(Lmax, n, d).to(dt).cuda()
    anc = torch.randint(0, n, (n, Lmax), dtype=torch.int32).cuda()
    ctx = torch.empty(n, d, dtype=dt, device="cuda")
    kc0, vc0 = kc.clone(), vc.clone()
    L.check(lib.em_dec_self_attention(em, L.ptr(qkv), L.ptr(kc), L.ptr(vc), L.ptr(anc), L.ptr(anc), n,
                                      d, heads, Lmax, pos, None, 3, None, L.ptr(ctx), None), "self_attn")
    torch.cuda.synchronize()
    q, k_new, v_new = qkv.float().split(d, dim=1)
    # row r attends to cache rows kc[j, anc[r, j]], j < pos, and its own new K / V
    idx = anc[:, :pos].long().t()                                        # (pos, n)
    jj = torch.arange(pos, device="cuda")[:, None].expand(pos, n)
    ks = torch.cat([kc0[jj, idx].float(), k_new[None]], 0).cpu()         # (pos + 1, n, d)
    vs = torch.cat([vc0[jj, idx].float(), v_new[None]], 0).cpu()
    qh = q.cpu().view(n, heads, 1, dk)
    kh = ks.view(pos + 1, n, heads, dk).permute(1, 2, 3, 0)              # (n, h, dk, pos + 1)
    vh = vs.view(pos + 1, n, heads, dk).permute(1, 2, 0, 3)              # (n, h, pos + 1, dk)
    sc = torch.matmul(qh, kh) / math.sqrt(dk)
    ref = torch.matmul(torch.softmax(sc, -1), vh).reshape(n, d)
    assert (ctx.float().cpu() - ref).abs().max().item() < tol
    # the new K/V were appended at `pos`, nothing else touched
    assert torch.equal(kc[pos].float().cpu(), k_new.to(dt).float().cpu())
    assert torch.equal(vc[pos].float().cpu(), v_new.to(dt).float().cpu())
    assert torch.equal(kc[:pos], kc0[:pos])


@pytest.mark.parametrize("tree", ["beam_tree", "random", "all_shared"])
@pytest.mark.parametrize("heads,d,B,W,pos,Lmax", [(8, 512, 64, 10, 133, 251), (8, 512, 16, 10, 0, 251), (4, 256, 3, 16, 255, 256),
                                                  (8, 512, 5, 7, 64, 100), (8, 512, 2, 10, 63, 251), (4, 256, 4, 3, 1, 40),
                                                  (8, 512, 16, 10, 257, 258), (4, 256, 2, 10, 390, 400)])
def test_dec_self_attention_beam(lib, tree, heads, d, B, W, pos, Lmax):
    """Round 5: `em_dec_self_attention_beam` - the decoder self-attention over the UNION of a beam's ancestors on MFMA
    (csrc/decoder.hip dec_self_attn_tree_kernel) - against plain torch fp32 attention of every row over its own path
    (attention.py:121-151 over the cached prefix).  Ancestor tables: a real token tree (beams that coalesce a few steps
    back: few distinct cache rows per position), independent random slots (every row its own ancestors: the key list at its
    longest), one shared path.  Shapes: configs[3]'s per-GPU rows at a mid position, position 0, the largest table
    (W = 16, Lmax = 256, last position: 4 096 keys), odd beam widths, a prefix of exactly one tile, the search's own
    Lmax = 258 at its last position and Lmax = 400 (two prefix positions per thread in the key-list pass)."""
    import os

    from espnet_amd import lib as L

    os.environ["ESPNET_AMD_SA_TREE_MIN_ROWS"] = "0"  # (the library takes the tree form from 200 rows: here at every size)
    L.load().em_dev_switches_reload()
    torch.manual_seed(3)
    n, dk = B * W, d // heads
    dt = torch.bfloat16
    qkv = torch.randn(n, 3 * d).to(dt).cuda()
    kc = torch.randn(Lmax, n, d).to(dt).cuda()
    vc = torch.randn(Lmax, n, d).to(dt).cuda()
    base = (torch.arange(n) // W * W)[:, None]
    if tree == "random":
        anc = base + torch.randint(0, W, (n, Lmax))
    elif tree == "all_shared":
        anc = base + torch.randint(0, W, (B, 1, Lmax)).expand(B, W, Lmax).reshape(n, Lmax)
    else:  # walk a beam forward: at every step each row continues one of the beam's rows (its parent), as the search does
        anc = torch.zeros(n, Lmax, dtype=torch.long)
        g = torch.Generator().manual_seed(5)
        for j in range(Lmax):
            parent = torch.randint(0, W, (B, W), generator=g)
            parent = torch.minimum(parent, torch.randint(0, W, (B, W), generator=g))  # (bias to low rows: beams coalesce)
            prev = anc.view(B, W, Lmax)
            new = torch.gather(prev, 1, parent[:, :, None].expand(B, W, Lmax)).clone()
            new[:, :, j] = torch.arange(W)[None, :] + (torch.arange(B) * W)[:, None]
            anc = new.view(n, Lmax)
    anc = anc.to(torch.int32).cuda()
    ctx = torch.zeros(n, d, dtype=dt, device="cuda")
    kc0, vc0 = kc.clone(), vc.clone()
    try:
        L.check(lib.em_dec_self_attention_beam(L.EM_BF16, L.ptr(qkv), L.ptr(kc), L.ptr(vc), L.ptr(anc), L.ptr(anc), n, d, heads,
                                               Lmax, pos, None, W, L.ptr(ctx), None), "self_attn_beam")
        torch.cuda.synchronize()
    finally:
        os.environ.pop("ESPNET_AMD_SA_TREE_MIN_ROWS", None)
        L.load().em_dev_switches_reload()
    q, k_new, v_new = qkv.float().split(d, dim=1)
    idx = anc[:, :pos].long().t()
    jj = torch.arange(pos, device="cuda")[:, None].expand(pos, n)
    ks = torch.cat([kc0[jj, idx].float(), k_new[None]], 0).cpu()
    vs = torch.cat([vc0[jj, idx].float(), v_new[None]], 0).cpu()
    qh = q.cpu().view(n, heads, 1, dk)
    kh = ks.view(pos + 1, n, heads, dk).permute(1, 2, 3, 0)
    vh = vs.view(pos + 1, n, heads, dk).permute(1, 2, 0, 3)
    sc = torch.matmul(qh, kh) / math.sqrt(dk)
    ref = torch.matmul(torch.softmax(sc, -1), vh).reshape(n, d)
    err = (ctx.float().cpu() - ref).abs().max().item()
    assert err < 3e-2, err
    # the new K / V were appended at `pos`, nothing else touched
    assert torch.equal(kc[pos], k_new.to(dt)) and torch.equal(vc[pos], v_new.to(dt))
    assert torch.equal(kc[:pos], kc0[:pos]) and torch.equal(kc[pos + 1 :], kc0[pos + 1 :])
    # ... and the per-row kernel agrees to bf16 round-off of the probabilities
    ctx2 = torch.zeros(n, d, dtype=dt, device="cuda")
    L.check(lib.em_dec_self_attention(L.EM_BF16, L.ptr(qkv), L.ptr(kc), L.ptr(vc), L.ptr(anc), L.ptr(anc), n, d, heads, Lmax,
                                      pos, None, (W + 1) // 2, None, L.ptr(ctx2), None), "self_attn")
    torch.cuda.synchronize()
    assert (ctx.float() - ctx2.float()).abs().max().item() < 3e-2


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("heads,d,W", [(8, 512, 10), (2, 64, 5), (4, 256, 20)])
def test_dec_src_attention(lib, prec, heads, d, W):
    from espnet_amd import lib as L

    em, dt, tol = _act(prec)
    torch.manual_seed(2)
    B, T = 3, 75
    Tpad = (T + 31) // 32 * 32
    klens = [75, 40, 1]
    dk = d // heads
    qs = torch.randn(B * W, d).to(dt).cuda()
    kv = torch.randn(B * T, 2 * d).to(dt).cuda()
    vT = torch.zeros(B, d, Tpad, dtype=dt, device="cuda")
    L.check(lib.em_dec_transpose_v(em, L.ptr(kv), B, T, d, Tpad, L.ptr(vT), None), "transpose_v")
    torch.cuda.synchronize()
    want_vT = kv.view(B, T, 2 * d)[:, :, d:].transpose(1, 2)
    assert torch.equal(vT[:, :, :T], want_vT)
    assert (vT[:, :, T:] == 0).all()
    ctx = torch.empty(B * W, d, dtype=dt, device="cuda")
    kl = torch.tensor(klens, dtype=torch.int32).cuda()
    L.check(lib.em_dec_src_attention(em, L.ptr(qs), L.ptr(kv), 2 * d, L.ptr(vT), L.ptr(kl), B, W, d,
                                     heads, T, Tpad, L.ptr(ctx), None), "src_attn")
    torch.cuda.synchronize()
    kvf = kv.float().cpu().view(B, T, 2 * d)
    for b in range(B):
        k = kvf[b, : klens[b], :d].view(-1, heads, dk).transpose(0, 1)
        v = kvf[b, : klens[b], d:].view(-1, heads, dk).transpose(0, 1)
        q = qs[b * W : (b + 1) * W].float().cpu().view(W, heads, dk).transpose(0, 1)
        sc = torch.matmul(q, k.transpose(1, 2)) / math.sqrt(dk)
        ref = torch.matmul(torch.softmax(sc, -1), v).transpose(0, 1).reshape(W, d)
        assert (ctx[b * W : (b + 1) * W].float().cpu() - ref).abs().max().item() < tol, b


@pytest.mark.parametrize("heads,d,W", [(8, 512, 10), (4, 256, 20), (8, 512, 3)])
def test_dec_src_attention_lnq_equals_two_launches(lib, heads, d, W):
    """Round 4: norm2 + src_attn.linear_q (decoder_layer.py:119-121, attention.py:94) in the prologue of the source-attention
    kernel (em_dec_src_attention_lnq, bf16): the context must equal em_ln_gemm followed by em_dec_src_attention BIT FOR BIT
    (same LayerNorm arithmetic, same summation order, q rounded to bf16 at the same point), W not a multiple of 16."""
    from espnet_amd import lib as L

    torch.manual_seed(7)
    B, T = 3, 75
    Tpad = (T + 31) // 32 * 32
    n = B * W
    dt = torch.bfloat16
    x = (torch.randn(n, d) * 2 + 0.3).cuda()
    g, be = (1 + 0.1 * torch.randn(d)).cuda(), (0.1 * torch.randn(d)).cuda()
    wq = (torch.randn(d, d) * d ** -0.5).to(dt).cuda()
    bq = (0.1 * torch.randn(d)).cuda()
    kv = torch.randn(B * T, 2 * d).to(dt).cuda()
    vT = torch.zeros(B, d, Tpad, dtype=dt, device="cuda")
    L.check(lib.em_dec_transpose_v(L.EM_BF16, L.ptr(kv), B, T, d, Tpad, L.ptr(vT), None), "transpose_v")
    kl = torch.tensor([75, 40, 1], dtype=torch.int32).cuda()
    qs = torch.empty(n, d, dtype=dt, device="cuda")
    L.check(lib.em_ln_gemm(L.EM_BF16, L.EM_EPI_STORE, L.ptr(x), L.ptr(g), L.ptr(be), 1e-12, L.ptr(wq), L.ptr(bq), L.ptr(qs),
                           n, d, d, d, None), "em_ln_gemm")
    ref = torch.empty(n, d, dtype=dt, device="cuda")
    L.check(lib.em_dec_src_attention(L.EM_BF16, L.ptr(qs), L.ptr(kv), 2 * d, L.ptr(vT), L.ptr(kl), B, W, d, heads, T, Tpad,
                                     L.ptr(ref), None), "src_attn")
    got = torch.full((n, d), 7.0, dtype=dt, device="cuda")
    L.check(lib.em_dec_src_attention_lnq(L.EM_BF16, L.ptr(x), L.ptr(g), L.ptr(be), 1e-12, L.ptr(wq), L.ptr(bq), L.ptr(kv),
                                         2 * d, L.ptr(vT), L.ptr(kl), B, W, d, heads, T, Tpad, L.ptr(got), None), "src_attn_lnq")
    torch.cuda.synchronize()
    assert torch.equal(got, ref), (got.float() - ref.float()).abs().max().item()
    # round 6: the same launch reading linear_q from its FRAGMENT-MAJOR copy (em_dec_src_attention_lnq_frag): the same bits
    got2 = torch.full((n, d), 7.0, dtype=dt, device="cuda")
    wqf = L.pack_frag16(wq)
    L.check(lib.em_dec_src_attention_lnq_frag(L.EM_BF16, L.ptr(x), L.ptr(g), L.ptr(be), 1e-12, L.ptr(wqf), L.ptr(bq), L.ptr(kv),
                                              2 * d, L.ptr(vT), L.ptr(kl), B, W, d, heads, T, Tpad, L.ptr(got2), None), "src_attn_lnq_frag")
    torch.cuda.synchronize()
    assert torch.equal(got2, ref), (got2.float() - ref.float()).abs().max().item()
    with pytest.raises(NotImplementedError):
        L.check(lib.em_dec_src_attention_lnq(L.EM_F32, L.ptr(x), L.ptr(g), L.ptr(be), 1e-12, L.ptr(wq), L.ptr(bq), L.ptr(kv),
                                             2 * d, L.ptr(vT), L.ptr(kl), B, W, d, heads, T, Tpad, L.ptr(got), None), "f32")


@pytest.mark.parametrize("T", [1, 2, 3, 6])
def test_search_very_short_memories_match_oracle(T):
    """Edge of the length logic: encoder memories of 1-6 frames (maxlen = T: the forced <eos> of
    batch_beam_search.py:393-410 fires almost immediately, the CTC recurrence has 0-5 steps).  Device n-best ==
    oracle n-best (token sequences; scores to fp32 round-off), alone and inside a ragged batch."""
    from oracle import beam_search as ob

    g = load_golden("tiny_beam5")
    sd = golden_state_dict(g)
    V = int(g["vocab"])
    dc = g["config"]["decoder_conf"]
    d = g["config"]["encoder_conf"]["output_size"]
    torch.manual_seed(100 + T)
    enc = torch.randn(T, d) * 0.7
    ref = ob.beam_search(sd, enc, dc["attention_heads"], dc["num_blocks"], int(g["beam"]), float(g["ctc_weight"]),
                         sos=V - 1, eos=V - 1)
    bs = build_search(g, sd, "float32")
    for graph in (False, True):
        bs.use_hipgraph = graph
        hyps = bs.search_batch(enc[None].cuda(), [T])[0]
        assert len(hyps) == len(ref), (len(hyps), len(ref))
        mine = {tuple(h.yseq.tolist()): float(h.score) for h in hyps}
        for r in ref:
            assert tuple(r["yseq"]) in mine
            assert abs(mine[tuple(r["yseq"])] - r["score"]) < 2e-3
    # the same utterance next to a long one
    long = torch.randn(40, d) * 0.7
    both = torch.zeros(2, 40, d)
    both[0, :T], both[1] = enc, long
    hb = bs.search_batch(both.cuda(), [T, 40])[0]
    assert [h.yseq.tolist() for h in hb] == [h.yseq.tolist() for h in hyps]


# per scored token over FIVE label steps of an unscaled random memory (no cancellation over 249 tokens as in the bench-size
# tests): measured on MI355X 1.17e-3 (T = 1 600) / 1.14e-3 (T = 1 480) through the two-launch source attention and the same
# through the fused kernel at T = 1 472 (the control of this test: both forms must sit under ONE bound)
LONG_MEMORY_EPS = 3e-3


@pytest.mark.parametrize("T", [1600, 1480, 1472])
def test_search_bf16_long_memory_falls_back_to_two_launch_source_attention(T):
    """norm2 + the query projection inside the source-attention kernel (`em_dec_src_attention_lnq`) needs more
    LDS than the plain kernel, so memories beyond Tpad 1 472 (d = 512) fit only the two-launch form - `decoder_step` must
    take it instead of failing with EM_ERR_UNSUPPORTED (the usable memory length stays ~1 690 frames).
    A 512-wide decoder over a T-frame memory (T = 1 600: only the two-launch form fits; 1 480: just past the switch; 1 472:
    the fused kernel's largest memory, the control), five label steps (maxlenratio -5), two utterances of different
    length: every device hypothesis re-scored teacher-forced under the oracle's scorers."""
    g = load_golden("large_beam10_3s")
    sd = golden_state_dict(g)
    d = g["config"]["encoder_conf"]["output_size"]
    torch.manual_seed(200 + T)
    enc = (torch.randn(2, T, d) * 0.7).to(torch.bfloat16).float()
    olens = [T, T - 37]
    bs = build_search(g, sd, "bfloat16")
    hyps = bs.search_batch(enc.cuda(), olens, maxlenratio=-5.0)
    assert len(hyps) == 2 and all(len(h) > 0 for h in hyps)
    from tests.helpers import oracle_rescore_batch

    dc = g["config"]["decoder_conf"]
    V = int(g["vocab"])
    worst = {"decoder": 0.0, "ctc": 0.0}
    for b in range(2):
        e = enc[b, : olens[b]]
        ys = [h.yseq.tolist() for h in hyps[b]]
        ref = oracle_rescore_batch(sd, e, ys, dc["attention_heads"], dc["num_blocks"], float(g["ctc_weight"]), V - 1, maxlen=5)
        for h, r in zip(hyps[b], ref):
            for k in worst:
                worst[k] = max(worst[k], abs(float(h.scores[k]) - r[k]) / max(r["n_scored"], 1))
    print(f"[long memory T = {T}] bf16 vs oracle per scored token over 5 steps: decoder {worst['decoder']:.2e}, ctc {worst['ctc']:.2e}")
    assert worst["decoder"] <= LONG_MEMORY_EPS and worst["ctc"] <= LONG_MEMORY_EPS, (T, worst)
