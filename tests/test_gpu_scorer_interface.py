"""The reference's per-step scorer interface on the device kernels:
`TransformerDecoder.batch_score`, `CTCPrefixScorer.batch_score_partial / select_state`, `LengthBonus.batch_score`,
`TransformerLM / SequentialRNNLM.batch_score` driven the way the reference's BatchBeamSearch drives them
(tests/scorer_driver.py, the reference's control flow restated and pinned to the reference by
tests/test_cpu_reference_binding.py) must reproduce the reference's own n-best fixtures -- all through the C-ABI
(`em_decoder_memory`, `em_decoder_step`, `em_ctc_log_probs_t`, `em_ctc_prefix_init/score/state`, `em_lm_step`)."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import golden_state_dict, load_golden  # noqa: E402
from tests.scorer_driver import drive_search  # noqa: E402
from tests.test_gpu_search import _sub, build_lm, oracle_enc  # noqa: E402


def build_scorers(g, sd, dtype="float32", lm=None):
    from espnet_amd.asr.ctc import CTC
    from espnet_amd.asr.decoder.transformer_decoder import TransformerDecoder
    from espnet_amd.nets.scorers.ctc import CTCPrefixScorer
    from espnet_amd.nets.scorers.length_bonus import LengthBonus

    V = int(g["vocab"])
    d = g["config"]["encoder_conf"]["output_size"]
    dec = TransformerDecoder(V, d, compute_dtype=dtype, **g["config"]["decoder_conf"])
    dec.load_state_dict(_sub(sd, "decoder."), strict=True)
    ctc = CTC(V, d, compute_dtype=dtype)
    ctc.load_state_dict(_sub(sd, "ctc."), strict=True)
    cw = float(g["ctc_weight"])
    scorers = dict(decoder=dec.cuda(), ctc=CTCPrefixScorer(ctc=ctc.cuda(), eos=V - 1), length_bonus=LengthBonus(V))
    weights = dict(decoder=1.0 - cw, ctc=cw, length_bonus=float(g["penalty"]) if "penalty" in g else 0.0)
    if lm is not None:
        scorers["lm"] = lm
        weights["lm"] = float(g["lm_weight"])
    return scorers, weights, V, cw


def check(g, nbest, tol_abs=2e-3, tol_rel=2e-5):
    keys = json.loads(str(g["score_keys"]))
    mine = {tuple(h["yseq"]): h for h in nbest}
    for k in range(len(g["yseq_lens"])):
        ref = tuple(g["yseq"][k, : g["yseq_lens"][k]].tolist())
        assert ref in mine, f"reference hypothesis #{k} missing"
        h = mine[ref]
        tol = tol_abs + tol_rel * abs(float(g["score"][k]))
        assert abs(h["score"] - float(g["score"][k])) < tol, (k, h["score"], float(g["score"][k]))
        for j, kk in enumerate(keys):
            assert abs(h["scores"][kk] - float(g["scores"][k, j])) < tol + tol_rel * abs(float(g["scores"][k, j])), kk
    if len(g["score"]) > 1 and float(g["score"][0] - g["score"][1]) > 1e-2:
        assert nbest[0]["yseq"] == g["yseq"][0, : g["yseq_lens"][0]].tolist()


@pytest.mark.parametrize("name", ["tiny_beam5", "tiny_beam3_attn_only", "tiny_beam4_early_eos", "tiny_beam4_minlen",
                                  "small_g2_3s"])
def test_reference_search_flow_over_accelerated_scorers_f32(name):
    g = load_golden(name)
    sd = golden_state_dict(g)
    enc, olens = oracle_enc(g, sd)
    scorers, weights, V, cw = build_scorers(g, sd)
    kw = {k: float(g[k]) for k in ("maxlenratio", "minlenratio") if k in g}
    x = enc[0, : int(olens[0])].cuda()
    nbest = drive_search(scorers, weights, int(g["beam"]), V, V - 1, V - 1, x,
                         pre_beam_score_key=None if cw == 1.0 else "full", **kw)
    check(g, nbest)


@pytest.mark.parametrize("name", ["tiny_beam5_lm", "tiny_beam4_lm_posenc", "tiny_beam5_rnnlm", "tiny_beam4_gru_nhid",
                                  "tiny_beam4_rnn_tanh"])
def test_reference_search_flow_with_lm_scorer_f32(name):
    g = load_golden(name)
    sd = golden_state_dict(g)
    enc, olens = oracle_enc(g, sd)
    scorers, weights, V, cw = build_scorers(g, sd, lm=build_lm(g, "float32"))
    x = enc[0, : int(olens[0])].cuda()
    nbest = drive_search(scorers, weights, int(g["beam"]), V, V - 1, V - 1, x)
    check(g, nbest)


def test_decoder_batch_score_and_forward_match_oracle_steps():
    """Unit level: `batch_score` against the oracle's K/V-cached decoder step (oracle/beam_search.py DecoderOracle)
    for n hypotheses with different prefixes, fed state by state; `score` (one hypothesis) and the teacher-forced
    `forward` give the same numbers."""
    from oracle import beam_search as ob

    g = load_golden("tiny_beam5")
    sd = golden_state_dict(g)
    enc, olens = oracle_enc(g, sd)
    scorers, _, V, _ = build_scorers(g, sd)
    dec = scorers["decoder"]
    dc = g["config"]["decoder_conf"]
    e = enc[0, : int(olens[0])]
    n, Lc = 4, 6
    gen = torch.Generator().manual_seed(5)
    ys = torch.randint(1, V - 1, (n, Lc), generator=gen)
    ys[:, 0] = V - 1
    orc = ob.DecoderOracle(sd, e, dc["attention_heads"], dc["num_blocks"], Lc + 2)
    cache = [(k.expand(n, -1, -1), v.expand(n, -1, -1)) for k, v in orc.init_cache()]
    xs = e.cuda().expand(n, *e.shape)
    states = [None] * n
    rows = []
    for pos in range(Lc):
        want, cache = orc.step(ys[:, pos], pos, cache)
        got, states = dec.batch_score(ys[:, : pos + 1].cuda(), states, xs)
        assert (got.cpu() - want).abs().max().item() < 2e-4, pos
        rows.append(want)
    one, _ = dec.score(ys[2, :1].cuda(), None, e.cuda())
    assert (one.cpu() - rows[0][2]).abs().max().item() < 2e-4
    # teacher-forced forward: scores before softmax, (B, L, V)
    out, _ = dec.forward(e.cuda().unsqueeze(0).expand(n, -1, -1).contiguous(), torch.tensor([e.size(0)] * n),
                         ys.cuda(), torch.tensor([Lc] * n))
    got = torch.log_softmax(out, dim=-1).cpu()
    assert (got - torch.stack(rows, 1)).abs().max().item() < 2e-4


@pytest.mark.parametrize("n", [160, 640, 100, 330])
def test_decoder_batch_score_bf16_at_640_rows(n):
    """The decoder step in the TIMED dtype at the row counts of configs[2] (160) and configs[3] per GPU (640; VERDICT r04
    item 1a): at 640 rows `em_decoder_step` runs the self-attention with one wave per row (heads x rows > 2048) and the
    LayerNorm + tiled GEMM pairs instead of the fused ln_gemm (grid > 256 workgroups) - variants no 160-row test reaches.
    Large decoder (6 x 512d, 8 heads, V = 5 000), n hypotheses with different 4-token prefixes over one 74-frame memory,
    against the oracle's f32 K/V-cached step: bound on the log-probabilities, and the error statistics over the entries a
    pre-beam can reach (the 15 best of every row) are printed - the device's per-entry noise level that
    tests/test_gpu_fullsize.py's path-noise model (PATH_NOISE_SIGMA) stands on.  Round 6: from 96 rows the step's projections
    run on fragment-major weights (csrc/dec_ffn.hip, mid_gemm<FRAG>); n = 100 and 330 are row counts that are NOT whole
    16-row fragments (the feed-forward keeps its row-major launches there, q | k | v / out-projections / - up to 320 rows -
    the logits take the fragment-major ones with masked edge rows)."""
    from oracle import beam_search as ob

    g = load_golden("large_beam10_3s")
    sd = golden_state_dict(g)
    enc, olens = oracle_enc(g, sd)
    scorers, _, V, _ = build_scorers(g, sd, dtype="bfloat16")
    dec = scorers["decoder"]
    dc = g["config"]["decoder_conf"]
    e = enc[0, : int(olens[0])].to(torch.bfloat16).float()  # the rows the bf16 step consumes
    Lc = 4
    gen = torch.Generator().manual_seed(11)
    ys = torch.randint(1, V - 1, (n, Lc), generator=gen)
    ys[:, 0] = V - 1
    orc = ob.DecoderOracle(sd, e, dc["attention_heads"], dc["num_blocks"], Lc + 2)
    cache = [(k.expand(n, -1, -1), v.expand(n, -1, -1)) for k, v in orc.init_cache()]
    xs = e.cuda().expand(n, *e.shape)
    states = [None] * n
    worst, sq, cnt = 0.0, 0.0, 0
    for pos in range(Lc):
        with torch.no_grad():
            want, cache = orc.step(ys[:, pos], pos, cache)
        got, states = dec.batch_score(ys[:, : pos + 1].cuda(), states, xs)
        diff = got.cpu() - want
        top = want.topk(15, dim=-1).indices
        dt = diff.gather(1, top)
        worst = max(worst, dt.abs().max().item())
        sq, cnt = sq + float((dt ** 2).sum()), cnt + dt.numel()
        assert diff.abs().max().item() < 8e-2, (pos, diff.abs().max().item())
        assert diff.abs().mean().item() < 8e-3, (pos, diff.abs().mean().item())
    rms = (sq / cnt) ** 0.5
    print(f"[decoder step bf16, {n} rows] log-prob error vs the f32 oracle over the 15 best entries of every row: "
          f"RMS {rms:.2e}, max {worst:.2e}")
    assert rms < 4e-3


def test_ctc_prefix_scorer_matches_oracle_steps():
    """Unit level: batch_score_partial / select_state against oracle CtcPrefixScorer.score, with and without a
    candidate list, incl. a repeated last label and the <eos> / blank columns."""
    from oracle import beam_search as ob
    from oracle import conformer as oc

    g = load_golden("tiny_beam5")
    sd = golden_state_dict(g)
    enc, olens = oracle_enc(g, sd)
    scorers, _, V, _ = build_scorers(g, sd)
    sc = scorers["ctc"]
    e = enc[0, : int(olens[0])]
    logp = oc.ctc_log_softmax(sd, e.unsqueeze(0))[0]
    orc = ob.CtcPrefixScorer(logp, V - 1)
    sc.batch_init_state(e.cuda())
    n = 3
    r_prev, s_prev = orc.initial_state()
    r_prev, s_prev = r_prev.expand(-1, -1, n).contiguous(), s_prev.expand(n).contiguous()
    states = [None] * n
    ys = torch.full((n, 1), V - 1, dtype=torch.int64)
    gen = torch.Generator().manual_seed(11)
    for step in range(5):
        ids = None if step == 3 else torch.stack([torch.randperm(V - 2, generator=gen)[:7] + 1 for _ in range(n)])
        if ids is not None and step > 0:
            ids[0, 0] = ys[0, -1]  # the candidate repeats the last label of prefix 0
        want, r_all, psi = orc.score(step, ys[:, -1], r_prev, s_prev, ids)
        got, st = sc.batch_score_partial(ys.cuda(), None if ids is None else ids.cuda(), states, e.cuda())
        seen = torch.zeros(n, V, dtype=torch.bool)
        if ids is None:
            seen[:] = True
        else:
            seen.scatter_(1, ids, True)
        seen[:, V - 1] = True
        diff = (got.cpu() - want)[seen].abs().max().item()
        assert diff < 2e-4 * max(1.0, want[seen].abs().max().item() * 1e-2), (step, diff)
        assert bool((got.cpu()[~seen] < -1e9).all())
        nxt = torch.tensor([int(ids[k, k % 7]) if ids is not None else 3 + k for k in range(n)])
        new_states = [sc.select_state(st, k, int(nxt[k])) for k in range(n)]
        for k in range(n):
            col = int((ids[k] == nxt[k]).nonzero()[0]) if ids is not None else int(nxt[k])
            rr = r_all[:, :, k, col]
            m = (new_states[k][0].cpu() - rr).abs()
            valid = rr > -1e9
            assert m[valid].max().item() < 2e-3, (step, k)
        r_prev = torch.stack([r_all[:, :, k, int((ids[k] == nxt[k]).nonzero()[0]) if ids is not None else int(nxt[k])]
                              for k in range(n)], dim=2)
        s_prev = psi[torch.arange(n), nxt]
        states = new_states
        ys = torch.cat([ys, nxt.unsqueeze(1)], dim=1)


def test_length_bonus_interface():
    from espnet_amd.nets.scorers.length_bonus import LengthBonus

    lb = LengthBonus(11)
    s, st = lb.batch_score(torch.zeros(3, 2, dtype=torch.int64), [None] * 3, torch.zeros(3, 4, 8, device="cuda"))
    assert s.shape == (3, 11) and bool((s == 1).all()) and st is None and s.is_cuda
    s1, _ = lb.score(torch.zeros(2, dtype=torch.int64), None, torch.zeros(4, 8, device="cuda"))
    assert s1.shape == (11,) and bool((s1 == 1).all())


def test_ctc_prefix_scorer_extend_prob_and_state_match_reference_recursion():
    """Streaming interface of the scorer (scorers/ctc.py:128-157 -> ctc_prefix_score.py:226-270): after the memory
    grows from T1 to T2 frames, extend_prob keeps the old frames' log-probs, extend_state continues every
    hypothesis state along the blank path, and the next batch_score_partial over the extended states equals the
    oracle's CtcPrefixScorer given the same (restated) extension."""
    from oracle import beam_search as ob
    from oracle import conformer as oc

    g = load_golden("tiny_beam5")
    sd = golden_state_dict(g)
    enc, olens = oracle_enc(g, sd)
    scorers, _, V, _ = build_scorers(g, sd)
    sc = scorers["ctc"]
    e = enc[0, : int(olens[0])]
    T2 = e.size(0)
    T1 = max(4, T2 // 2)
    logp = oc.ctc_log_softmax(sd, e.unsqueeze(0))[0]
    o1 = ob.CtcPrefixScorer(logp[:T1], V - 1)
    sc.batch_init_state(e[:T1].cuda())
    n = 3
    r_prev, s_prev = o1.initial_state()
    r_prev, s_prev = r_prev.expand(-1, -1, n).contiguous(), s_prev.expand(n).contiguous()
    ys = torch.full((n, 1), V - 1, dtype=torch.int64)
    states = [None] * n
    for step in range(2):  # two label steps on the first block
        ids = torch.stack([torch.arange(1 + 3 * k + step, 8 + 3 * k + step) for k in range(n)])
        want, r_all, psi = o1.score(step, ys[:, -1], r_prev, s_prev, ids)
        got, st = sc.batch_score_partial(ys.cuda(), ids.cuda(), states, e[:T1].cuda())
        nxt = ids[:, 2]
        states = [sc.select_state(st, k, int(nxt[k])) for k in range(n)]
        r_prev = torch.stack([r_all[:, :, k, 2] for k in range(n)], dim=2)
        s_prev = psi[torch.arange(n), nxt]
        ys = torch.cat([ys, nxt.unsqueeze(1)], dim=1)
    # ---- the block grows: extend_prob + extend_state
    lp_before = sc._lpT.clone()
    sc.extend_prob(e.cuda())
    assert sc._T == T2 and torch.equal(sc._lpT[:, :T1], lp_before)            # old frames keep their values (:243)
    assert (sc._lpT[:, T1:].cpu().T - logp[T1:]).abs().max().item() < 2e-4
    sc.extend_prob(e[: T2 - 1].cuda())                                        # not longer: nothing happens (:231)
    assert sc._T == T2
    new_states = sc.extend_state(states + [None])
    assert new_states[-1] is None and len(new_states) == n + 1
    o2 = ob.CtcPrefixScorer(logp, V - 1)
    r_ext = torch.full((T2, 2, n), ob.LOGZERO)
    r_ext[:T1] = r_prev
    for t in range(T1, T2):                                                   # ctc_prefix_score.py:266-268
        r_ext[t, 1] = r_ext[t - 1, 1] + o2.x[0, t, 0]
    for k in range(n):
        got_r = new_states[k][0].cpu()
        assert got_r.shape == (T2, 2)
        assert torch.equal(got_r[:T1], states[k][0].cpu())
        assert bool((got_r[T1:, 0] < -1e9).all())
        assert (got_r[T1:, 1] - r_ext[T1:, 1, k]).abs().max().item() < 2e-3
    # ---- and the search goes on over the extended states
    ids = torch.stack([torch.arange(2 + k, 9 + k) for k in range(n)])
    want, _, _ = o2.score(2, ys[:, -1], r_ext, s_prev, ids)
    got, _ = sc.batch_score_partial(ys.cuda(), ids.cuda(), new_states[:n], e.cuda())
    seen = torch.zeros(n, V, dtype=torch.bool).scatter_(1, ids, True)
    seen[:, V - 1] = True
    assert (got.cpu() - want)[seen].abs().max().item() < 2e-3


def test_ctc_prefix_scorer_non_batched_contract():
    """scorers/ctc.py:25-86: init_state -> (0, state); score_partial -> (len(ids),) scores and a state that
    select_state(state, i) indexes by the position inside ids (the reference's BeamSearch.merge_states,
    beam_search.py:313, calls it without new_id).  Checked against the batched entry of the same object."""
    g = load_golden("tiny_beam5")
    sd = golden_state_dict(g)
    enc, olens = oracle_enc(g, sd)
    scorers, _, V, _ = build_scorers(g, sd)
    sc = scorers["ctc"]
    e = enc[0, : int(olens[0])].cuda()
    st0 = sc.init_state(e)
    assert isinstance(st0, tuple) and len(st0) == 2 and st0[0] == 0
    y = torch.tensor([V - 1], dtype=torch.int64, device="cuda")
    ids = torch.tensor([5, 9, 2, V - 1, 17], dtype=torch.int64, device="cuda")
    s1, st1 = sc.score_partial(y, ids, st0, e)
    assert s1.shape == (5,)
    full, bst = sc.batch_score_partial(y.unsqueeze(0), ids.unsqueeze(0), [None], e)
    assert torch.allclose(s1, full[0, ids])
    sel = sc.select_state(st1, 1)              # position 1 -> label 9
    assert float(sel[0]) == pytest.approx(float(bst.log_psi[0, 9]))
    ref = sc.select_state(bst, 0, 9)
    assert torch.equal(sel[1][0], ref[0]) and int(sel[1][2]) == 9
    # second step from the selected state: equals the batched call on the same state
    y2 = torch.tensor([V - 1, 9], dtype=torch.int64, device="cuda")
    ids2 = torch.tensor([9, 3, 11], dtype=torch.int64, device="cuda")
    s2, _ = sc.score_partial(y2, ids2, sel, e)
    full2, _ = sc.batch_score_partial(y2.unsqueeze(0), ids2.unsqueeze(0), [ref], e)
    assert torch.allclose(s2, full2[0, ids2])
