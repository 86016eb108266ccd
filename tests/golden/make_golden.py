#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/*.npz by running THE REFERENCE ITSELF
(/root/reference, imported read-only through the no-op shims in tests/golden/_shims) on CPU/fp32.

Run in the build container only (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [case ...]

What is pinned, per case (all produced by reference code paths, cited):
  * feats      espnet2/asr/espnet_model.py:450-467 `_extract_feats` (Stft + LogMel)
  * enc_out    espnet2/asr/espnet_model.py:380-448 `encode` (UtteranceMVN + ConformerEncoder)
  * block outs ConformerEncoder.forward(return_all_hs=True) (conformer_encoder.py:377-384)
  * ctc ids    espnet2/asr/ctc.py:207-215 `argmax`; G1 tokens = groupby + drop blank/sos/eos
               (espnet2/bin/asr_inference.py:574-575)
  * n-best     espnet2/bin/asr_inference.py:490-677 `Speech2Text.__call__` (BatchBeamSearch,
               CTCPrefixScorer, TransformerDecoder, LengthBonus)

Weights are the deterministic recipe of oracle/weights.py loaded into the reference model with
`load_state_dict` (so fixtures carry only inputs' seeds and outputs); the key->shape table of the
reference state_dict is stored too and the tests assert the replacement exposes the same table.
"""
import json
import os
import sys
import tempfile
import time
from itertools import groupby
from pathlib import Path

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.dont_write_bytecode = True
sys.path[:0] = [str(HERE / "_shims"), "/root/reference", str(REPO)]
# where the fixtures are written: tests/golden itself, or a scratch directory (tests/test_cpu_golden_regen.py re-runs cheap
# cases into a tmpdir and compares them with the committed files bit for bit)
OUT = Path(os.environ.get("GOLDEN_OUT") or HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

from oracle.weights import recipe_state_dict, synth_waveform, token_list  # noqa: E402

torch.set_grad_enabled(False)

SMALL = dict(
    encoder="conformer",
    encoder_conf=dict(
        output_size=256, attention_heads=4, linear_units=1024, num_blocks=12,
        dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.1,
        input_layer="conv2d", normalize_before=True, macaron_style=True,
        rel_pos_type="latest", pos_enc_layer_type="rel_pos",
        selfattention_layer_type="rel_selfattn", activation_type="swish",
        use_cnn_module=True, cnn_module_kernel=31,
    ),
    decoder="transformer",
    decoder_conf=dict(attention_heads=4, linear_units=2048, num_blocks=6, dropout_rate=0.1,
                      positional_dropout_rate=0.1, self_attention_dropout_rate=0.1,
                      src_attention_dropout_rate=0.1),
    model_conf=dict(ctc_weight=0.3, lsm_weight=0.1, length_normalized_loss=False),
    frontend_conf=dict(n_fft=512, win_length=400, hop_length=160),
)
LARGE = dict(
    encoder="conformer",
    encoder_conf=dict(
        output_size=512, attention_heads=8, linear_units=2048, num_blocks=12,
        dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.1,
        input_layer="conv2d", normalize_before=True, macaron_style=True,
        rel_pos_type="latest", pos_enc_layer_type="rel_pos",
        selfattention_layer_type="rel_selfattn", activation_type="swish",
        use_cnn_module=True, cnn_module_kernel=31,
    ),
    decoder="transformer",
    decoder_conf=dict(attention_heads=8, linear_units=2048, num_blocks=6, dropout_rate=0.1,
                      positional_dropout_rate=0.1, self_attention_dropout_rate=0.1,
                      src_attention_dropout_rate=0.1),
    model_conf=dict(ctc_weight=0.3, lsm_weight=0.1, length_normalized_loss=False),
    frontend_conf=dict(n_fft=512, hop_length=160),
)


def tiny(d=64, heads=1, ff=128, blocks=2, dec_blocks=2, kernel=31):
    c = json.loads(json.dumps(SMALL))
    c["encoder_conf"].update(output_size=d, attention_heads=heads, linear_units=ff,
                             num_blocks=blocks, cnn_module_kernel=kernel)
    c["decoder_conf"].update(attention_heads=heads, linear_units=ff, num_blocks=dec_blocks)
    return c


def with_input_layer(conf, layer, blocks=None):
    c = json.loads(json.dumps(conf))
    c["encoder_conf"]["input_layer"] = layer
    if blocks is not None:
        c["encoder_conf"]["num_blocks"] = blocks
    return c


def with_legacy(conf):
    c = json.loads(json.dumps(conf))
    c["encoder_conf"]["rel_pos_type"] = "legacy"
    return c


def build_reference(conf, vocab, workdir, **s2t_kwargs):
    from espnet2.bin.asr_inference import Speech2Text
    from espnet2.tasks.asr import ASRTask

    workdir = Path(workdir)
    tok = workdir / "tokens.txt"
    tok.write_text("\n".join(token_list(vocab)) + "\n")
    cfg_in = workdir / "train.yaml"
    cfg_in.write_text(yaml.safe_dump(conf))
    ASRTask.main(cmd=["--dry_run", "true", "--output_dir", str(workdir / "asr"),
                      "--token_list", str(tok), "--token_type", "word",
                      "--config", str(cfg_in)])
    cfg = workdir / "asr" / "config.yaml"
    s2t = Speech2Text(asr_train_config=str(cfg), asr_model_file=None, device="cpu",
                      dtype="float32", **s2t_kwargs)
    return s2t, cfg.read_text()


def load_recipe(model, seed, tweaks=None):
    sd = model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    new = recipe_state_dict(shapes, seed)
    new["frontend.logmel.melmat"] = sd["frontend.logmel.melmat"].clone()
    for key, idx, delta in tweaks or []:  # e.g. bias the <eos> logit so hypotheses end early
        new[key][idx] += delta
    model.load_state_dict(new, strict=True)
    model.eval()
    return shapes


def g1_tokens(ids, exclude):
    return [int(x[0]) for x in groupby(ids) if int(x[0]) not in exclude]


def fit_peaked_ctc_head(model, enc, olens, seed, level=None):
    """A CTC head with PEAKED posteriors for the random-init encoder: random-init logits
    are nearly flat (top-2 margins of 1e-3 .. 1e-1), so "the bf16 path returns the reference's tokens" cannot be
    asked of them.  Here `ctc_lo` is FITTED (ridge regression, dual form) to a synthetic frame labelling - label
    runs of 3-8 frames from a 60-label set, every third run blank, some labels recurring - on the REFERENCE's own
    encoder output, the way a trained head separates its frames; everything downstream (log-softmax, arg-max, G1
    collapse, margins) is then computed by the reference with that head.  Only the rows of the labels used are
    non-zero; they are stored in the fixture (`ctc_rows`, `ctc_w_rows`, `ctc_b_rows`) and loaded on top of the
    recipe weights by tests/helpers.py::golden_state_dict.  Needs fewer frames than dimensions (one 10 s utterance
    of the 256-wide model: 249 frames)."""
    rng = np.random.RandomState(seed)
    V = model.ctc.ctc_lo.weight.size(0)
    E = torch.cat([enc[b, : int(olens[b])] for b in range(enc.size(0))]).double().numpy()
    T, d = E.shape
    assert T < d, (T, d)
    labels = rng.choice(np.arange(1, V - 1), size=60, replace=False)
    want = np.zeros(T, dtype=np.int64)
    t = k = 0
    while t < T:
        run = int(rng.randint(3, 9))
        want[t : t + run] = 0 if k % 3 == 2 else int(labels[rng.randint(0, 60)])
        t, k = t + run, k + 1
    rows = np.unique(want)
    Y = np.zeros((T, len(rows)))
    # level None: logit 8 on the intended label (margins > 1).  level (lo, hi): the logit of frame t is drawn
    # log-uniformly from [lo, hi] - the reference's top-2 margins then POPULATE that range (every other label sits at
    # ~0), which is where a trained model's hard frames live (BF16_MARGIN probed from above)
    amp = np.full(T, 8.0) if level is None else np.exp(rng.uniform(np.log(level[0]), np.log(level[1]), size=T))
    Y[np.arange(T), np.searchsorted(rows, want)] = amp
    mu = E.mean(0)
    Ec = E - mu
    A = np.linalg.solve(Ec @ Ec.T + 1.0 * np.eye(T), Y)
    W = (Ec.T @ A).T
    w = torch.zeros_like(model.ctc.ctc_lo.weight)
    b = torch.zeros_like(model.ctc.ctc_lo.bias)
    w[torch.from_numpy(rows)] = torch.from_numpy(W).float()
    b[torch.from_numpy(rows)] = torch.from_numpy(-(W @ mu)).float()
    model.ctc.ctc_lo.weight.copy_(w)
    model.ctc.ctc_lo.bias.copy_(b)
    return dict(ctc_rows=rows, ctc_w_rows=w[torch.from_numpy(rows)].numpy().copy(),
                ctc_b_rows=b[torch.from_numpy(rows)].numpy().copy(), ctc_fit_labels=want)


def _ridge_rows(H, Y, lam):
    """Rows (W, b) with W h_s + b ~= Y[s] for the states H (S, d), ridge regression in its dual form around the
    mean state (S < d: an interpolating fit, as a trained head separates its training frames)."""
    mu = H.mean(0)
    Hc = H - mu
    A = np.linalg.solve(Hc @ Hc.T + lam * np.eye(H.shape[0]), Y - Y.mean(0))
    W = (Hc.T @ A).T
    return W, Y.mean(0) - W @ mu


def fit_peaked_search_heads(model, enc, olens, seed, g_adjust=None):
    """PEAKED posteriors for the joint CTC/attention search of a random-init model: with flat
    random-init posteriors the reference's own n-best hypotheses differ by 1e-3 in score, so "the bf16 search returns
    the reference's n-best" cannot be asked of them.  Here `ctc.ctc_lo` and `decoder.output_layer` are FITTED on the
    REFERENCE's own encoder output / decoder states to a synthetic transcript y* (label runs of 2-4 frames between
    blank runs of 1-3) the way trained heads behave: on the frames of token k the CTC head puts logit 12 on y*_k and,
    at five positions, 12 - c_k on ONE confusable label alt_k; after the prefix y*_<k (or the same prefix with earlier
    tokens replaced by their confusables) the decoder puts 12 on y*_k and 12 - g_k on alt_k, <eos> after the last
    token.  The hypothesis space the search then explores is y* and its substitution variants with designed, well
    separated costs (0.7 g_k + 0.3 x CTC cost); every other row of both heads keeps its recipe weights (no exact ties).
    Everything downstream - log-softmax, prefix scores, pre-beam, pruning, n-best - is computed by the reference with
    those heads.  The fitted rows are stored in the fixture (`ctcov_*`, `decov_*`) and laid over the recipe weights by
    tests/helpers.py::golden_state_dict."""
    rng = np.random.RandomState(seed)
    V = model.ctc.ctc_lo.weight.size(0)
    T = int(olens[0])
    E = enc[0, :T].double().numpy()
    d = E.shape[1]
    npool = min(200, V - 3)
    pool = rng.choice(np.arange(2, V - 1), size=npool, replace=False)
    # frame labelling
    want = np.zeros(T, dtype=np.int64)
    runs = []
    t, used = 0, 0
    while True:
        t += int(rng.randint(1, 4))
        run = int(rng.randint(2, 5))
        if t + run > T - 1:
            break
        # (all labels distinct: a random-init decoder's state is dominated by the embedding of the last token - the
        # position and the rest of the prefix are 1/sqrt(d)-sized corrections - so a label that recurs would ask the
        # head for two different continuations of nearly the same state)
        lab, used = int(pool[used]), used + 1
        want[t : t + run] = lab
        runs.append((lab, t, run))
        t += run
    L = len(runs)
    y_star = [r[0] for r in runs]
    # Five positions get a confusable label.  Their substitution costs (in joint score: 0.7 x decoder logit gap +
    # 0.3 x CTC cost) are DESIGNED so that the ten cheapest subsets of substitutions - the n-best of a beam-10 search:
    # the decoder's state hardly depends on earlier tokens, so substitutions combine additively - are 0.54 apart
    # (random search over 4e5 cost vectors for the largest minimum gap of the eleven smallest subset sums, all ten
    # within 7 of the best so that they stay clear of the unfitted hypotheses, which cost ~ 9.5).  55 % of a cost sits
    # in the decoder (logit gap <= 5.8: the confusable stays far above the recipe rows' logits, |.| < 2, so it is in
    # every pre-beam), 45 % in the CTC head.
    design = np.array([1.51851315, 2.0921559, 2.63192038, 5.28805857, 7.39653826])
    alt_pos = np.sort(rng.choice(np.arange(1, L - 1), size=5, replace=False))
    design = design[rng.permutation(5)]
    alts = [-1] * L
    c, g = np.zeros(L), np.zeros(L)
    for q, k in enumerate(alt_pos):
        alts[k] = int(pool[npool - 5 + q])
        c[k] = 0.45 * design[q] / 0.3 / runs[k][2]
        g[k] = 0.55 * design[q] / 0.7
        if g_adjust is not None:  # (later passes: the CTC cost of a run is only roughly run x c; see run_search_case)
            g[k] += g_adjust[q]
    TOP = 12.0  # logit of the intended label (p ~ 0.97 against 5 000 recipe-weight rows)
    assert T < d
    # ---- CTC head: rows of blank, y*, alts
    rows_c = np.unique(np.array([0] + y_star + [a for a in alts if a >= 0]))
    Yc = np.zeros((T, len(rows_c)))
    col = {int(r): i for i, r in enumerate(rows_c)}
    Yc[:, col[0]] = np.where(want == 0, TOP, 0.0)
    for k, (lab, t0, run) in enumerate(runs):
        Yc[t0 : t0 + run, col[lab]] = TOP
        if alts[k] >= 0:
            Yc[t0 : t0 + run, col[alts[k]]] = TOP - c[k]
    Wc, bc = _ridge_rows(E, Yc, 1.0)
    rc = torch.from_numpy(rows_c)
    model.ctc.ctc_lo.weight[rc] = torch.from_numpy(Wc).float()
    model.ctc.ctc_lo.bias[rc] = torch.from_numpy(bc).float()
    # ---- decoder head: states of y* and of every single-substitution variant, teacher-forced through the reference
    eos = V - 1
    states, targets = [], []
    rows_d = np.unique(np.array(y_star + [a for a in alts if a >= 0] + [eos]))
    cold = {int(r): i for i, r in enumerate(rows_d)}
    grabbed = []
    hook = model.decoder.output_layer.register_forward_hook(lambda m, inp, out: grabbed.append(inp[0].detach()))
    seqs = [(0, list(y_star))] + [(int(k) + 1, y_star[:k] + [alts[k]] + y_star[k + 1 :]) for k in alt_pos]
    for first, seq in seqs:  # (the variant with position k replaced shares its states 0..k with y*)
        ys_in = torch.tensor([[eos] + seq], dtype=torch.long)
        grabbed.clear()
        model.decoder(enc[:, :T], torch.tensor([T]), ys_in, torch.tensor([L + 1]))
        h = grabbed[0][0].double().numpy()  # (L + 1, d): state j predicts token j
        for j in range(first, L + 1):
            y = np.zeros(len(rows_d))
            if j < L:
                y[cold[y_star[j]]] = TOP
                if alts[j] >= 0:
                    y[cold[alts[j]]] = TOP - g[j]
            else:
                y[cold[eos]] = TOP
            states.append(h[j])
            targets.append(y)
    hook.remove()
    H, Yd = np.stack(states), np.stack(targets)
    assert H.shape[0] < d, H.shape
    Wd, bd = _ridge_rows(H, Yd, 1.0)
    rd = torch.from_numpy(rows_d)
    model.decoder.output_layer.weight[rd] = torch.from_numpy(Wd).float()
    model.decoder.output_layer.bias[rd] = torch.from_numpy(bd).float()
    fit_err = float(np.abs(H @ Wd.T + bd - Yd).max())
    print(f"  fitted heads: L={L} tokens, {H.shape[0]} decoder states, |W_dec row| max {np.linalg.norm(Wd, axis=1).max():.2f}, "
          f"|W_ctc row| max {np.linalg.norm(Wc, axis=1).max():.2f}, decoder fit err {fit_err:.3f}")
    return dict(ctcov_rows=rows_c, ctcov_w=model.ctc.ctc_lo.weight[rc].numpy().copy(),
                ctcov_b=model.ctc.ctc_lo.bias[rc].numpy().copy(),
                decov_rows=rows_d, decov_w=model.decoder.output_layer.weight[rd].numpy().copy(),
                decov_b=model.decoder.output_layer.bias[rd].numpy().copy(),
                y_star=np.array(y_star), y_alts=np.array(alts), fit_frames=want, alt_pos=alt_pos,
                design_cost=design)


def run_encode_case(name, conf, vocab, wseed, utt_ids, lengths, keep_every=1, with_blocks=False, peaked_seed=None,
                    peaked_level=None):
    t0 = time.time()
    with tempfile.TemporaryDirectory() as td:
        s2t, cfg_text = build_reference(conf, vocab, td, beam_size=1, ctc_weight=1.0)
    model = s2t.asr_model
    shapes = load_recipe(model, wseed)
    nmax = max(lengths)
    speech = torch.zeros(len(utt_ids), nmax)
    for i, (u, n) in enumerate(zip(utt_ids, lengths)):
        speech[i, :n] = synth_waveform(u, n)
    lens = torch.tensor(lengths, dtype=torch.long)
    feats, flens = model._extract_feats(speech, lens)
    feats_keep = feats.clone()
    enc, olens = model.encode(speech, lens)
    out = dict(
        config_yaml=np.array(cfg_text), vocab=np.array(vocab), wseed=np.array(wseed),
        utt_ids=np.array(utt_ids), lengths=np.array(lengths),
        state_shapes=np.array(json.dumps({k: list(v) for k, v in shapes.items()})),
        melmat=model.frontend.logmel.melmat.numpy(),
        feats=feats_keep.numpy(), feats_lens=flens.numpy(),
        enc_out=enc[:, ::keep_every].numpy().copy(), enc_keep_every=np.array(keep_every),
        enc_olens=olens.numpy(),
    )
    if with_blocks:
        nf, nl = model.normalize(feats_keep.clone(), flens)
        from espnet2.legacy.nets.pytorch_backend.nets_utils import make_pad_mask
        masks = (~make_pad_mask(nl)[:, None, :])
        xs, masks2 = model.encoder.embed(nf, masks)
        out["embed_x"] = xs[0].numpy().copy()
        out["pos_emb"] = xs[1].numpy().copy()
        blocks = []
        for layer in model.encoder.encoders:
            xs, masks2 = layer(xs, masks2)
            blocks.append(xs[0].numpy().copy())
        out["block_outs"] = np.stack(blocks)
    if peaked_seed is not None:
        out.update(fit_peaked_ctc_head(model, enc, olens, peaked_seed, peaked_level))
    logp = model.ctc.log_softmax(enc)
    ids = model.ctc.argmax(enc)
    out["ctc_ids"] = ids.numpy()
    top2 = logp.topk(2, dim=-1).values
    out["ctc_margin"] = (top2[..., 0] - top2[..., 1]).numpy()
    out["ctc_logp_head"] = logp[:, :4].numpy().copy()
    excl = [model.blank_id, model.sos, model.eos]
    toks = [g1_tokens(ids[b, : int(olens[b])].tolist(), excl) for b in range(len(utt_ids))]
    out["g1_lens"] = np.array([len(t) for t in toks])
    g1 = np.full((len(toks), max(1, max(len(t) for t in toks))), -1, dtype=np.int64)
    for b, t in enumerate(toks):
        g1[b, : len(t)] = t
    out["g1_tokens"] = g1
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(f"[{name}] done in {time.time()-t0:.1f}s feats{tuple(feats.shape)} enc{tuple(enc.shape)} "
          f"olens={olens.tolist()} g1_lens={out['g1_lens'].tolist()} "
          f"min margin={out['ctc_margin'].min():.2e}")


def qualify_peaked_nbest(model, enc, results, beam, ctc_weight, noise, seeds=4):
    """Is the reference's n-best separated by more than a reduced-precision implementation's error?  The oracle search
    (pinned to the reference: without noise it must reproduce `results` exactly) is re-run with N(0, noise^2) added to
    every decoder log-probability and CTC log-posterior; the n-best token sequences must not move."""
    from oracle import beam_search as ob

    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    dc = model.decoder
    heads = dc.decoders[0].self_attn.h
    ref = [h.yseq.tolist() for _, _, _, h in results]
    for k in range(seeds + 1):
        got = ob.beam_search(sd, enc, heads, len(dc.decoders), beam, ctc_weight, model.sos, model.eos,
                             noise=noise if k else 0.0, noise_seed=k)
        mine = [r["yseq"] for r in got[: len(ref)]]
        if mine != ref:
            moved = [i for i, (a, b) in enumerate(zip(mine, ref)) if a != b]
            raise AssertionError(f"n-best moved under noise {noise if k else 0.0} (seed {k}): ranks {moved}; "
                                 f"scores {[round(r['score'], 3) for r in got[:len(ref)]]}")
    return noise


def run_search_case(name, conf, vocab, wseed, utt_id, n_samples, beam, ctc_weight, nbest,
                    keep_every=1, penalty=0.0, maxlenratio=0.0, minlenratio=0.0, tweaks=None, peaked_seed=None,
                    qualify_noise=0.05):
    t0 = time.time()
    with tempfile.TemporaryDirectory() as td:
        s2t, cfg_text = build_reference(conf, vocab, td, beam_size=beam, ctc_weight=ctc_weight,
                                        nbest=nbest, penalty=penalty, lm_weight=0.0,
                                        maxlenratio=maxlenratio, minlenratio=minlenratio)
    model = s2t.asr_model
    shapes = load_recipe(model, wseed, tweaks)
    wav = synth_waveform(utt_id, n_samples)
    enc, olens = model.encode(wav[None], torch.tensor([n_samples]))
    extra = {}
    if peaked_seed is not None:
        # fit, measure the realised cost of each single substitution (teacher-forced under the oracle's scorers), move
        # the decoder gaps by the difference to the designed costs, fit again
        from tests.helpers import oracle_rescore

        adjust = np.zeros(5)
        dcf = conf["decoder_conf"]
        for it in range(3):
            extra = fit_peaked_search_heads(model, enc, olens, peaked_seed, adjust)
            sd_now = {k: v.detach().clone() for k, v in model.state_dict().items()}
            e = enc[0, : int(olens[0])]
            ys = [int(x) for x in extra["y_star"]]
            eos = vocab - 1
            base = oracle_rescore(sd_now, e, [eos] + ys + [eos], dcf["attention_heads"], dcf["num_blocks"], ctc_weight, eos)
            real = []
            for q, k in enumerate(extra["alt_pos"]):
                v = list(ys)
                v[int(k)] = int(extra["y_alts"][int(k)])
                r = oracle_rescore(sd_now, e, [eos] + v + [eos], dcf["attention_heads"], dcf["num_blocks"], ctc_weight, eos)
                real.append(base["score"] - r["score"])
            real = np.array(real)
            print(f"  pass {it}: realised single-substitution costs {np.round(real, 3)} designed {np.round(extra['design_cost'], 3)}")
            adjust = adjust + (extra["design_cost"] - real) / (1.0 - ctc_weight)
        extra["realised_cost"] = real
    t1 = time.time()
    results = s2t(wav.numpy())
    t_dec = time.time() - t1
    if peaked_seed is not None:
        sc = [float(h.score) for _, _, _, h in results]
        print("  n-best scores", [round(x, 3) for x in sc], "lens", [len(h.yseq) for _, _, _, h in results])
        extra["nbest_min_gap"] = np.array(min(a - b for a, b in zip(sc, sc[1:])))
        if os.environ.get("PEAKED_NOQUAL"):  # (debugging a candidate fixture: dump it unqualified)
            qualify_noise = 0.0
        extra["noise_qualified"] = np.array(qualify_peaked_nbest(model, enc[0, : int(olens[0])], results, beam,
                                                                 ctc_weight, qualify_noise))
    L = max(len(h.yseq) for _, _, _, h in results)
    yseq = np.full((len(results), L), -1, dtype=np.int64)
    for i, (_, _, _, h) in enumerate(results):
        yseq[i, : len(h.yseq)] = h.yseq.numpy()
    keys = sorted(results[0][3].scores.keys())
    out = dict(
        config_yaml=np.array(cfg_text), vocab=np.array(vocab), wseed=np.array(wseed),
        utt_id=np.array(utt_id), n_samples=np.array(n_samples), beam=np.array(beam),
        ctc_weight=np.array(ctc_weight), nbest=np.array(nbest),
        penalty=np.array(penalty), maxlenratio=np.array(maxlenratio),
        minlenratio=np.array(minlenratio), tweaks=np.array(json.dumps(tweaks or [])),
        n_ended=np.array(len(results)),
        state_shapes=np.array(json.dumps({k: list(v) for k, v in shapes.items()})),
        melmat=model.frontend.logmel.melmat.numpy(),
        enc_out=enc[0, ::keep_every].numpy().copy(), enc_keep_every=np.array(keep_every),
        yseq=yseq, yseq_lens=np.array([len(h.yseq) for _, _, _, h in results]),
        score=np.array([float(h.score) for _, _, _, h in results]),
        score_keys=np.array(json.dumps(keys)),
        scores=np.array([[float(h.scores[k]) for k in keys] for _, _, _, h in results]),
        token_int_best=np.array(results[0][2], dtype=np.int64),
        ref_seconds=np.array(t_dec),
    )
    out.update(extra)
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(f"[{name}] done in {time.time()-t0:.1f}s (Speech2Text {t_dec:.1f}s) T={enc.shape[1]} "
          f"best len={len(results[0][3].yseq)} score={float(results[0][3].score):.4f} keys={keys}")


def cli_wave(utt_id, n):
    """int16-quantised synthetic waveform of the decode-CLI golden (so the wav file is the input)."""
    return np.clip(np.rint(synth_waveform(utt_id, n).numpy().astype(np.float64) * 32768.0), -32768, 32767).astype("<i2")


def run_cli_case(name, conf, vocab, wseed, utts, beam, ctc_weight, nbest):
    """The reference's decode CLI end to end (espnet2/bin/asr_inference.py `main`): wav.scp of 16-bit
    PCM files -> output_dir/{n}best_recog/{token,token_int,score,text}.  `utts` = [(key, utt_id, n)]
    in wav.scp order; a too-short entry exercises the TooShortUttError fallback (:851-858)."""
    import wave

    from espnet2.bin.asr_inference import main as ref_main

    t0 = time.time()
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        s2t, cfg_text = build_reference(conf, vocab, td, beam_size=beam, ctc_weight=ctc_weight, nbest=nbest)
        shapes = load_recipe(s2t.asr_model, wseed)
        torch.save(s2t.asr_model.state_dict(), td / "model.pth")
        lines = []
        for key, u, n in utts:
            with wave.open(str(td / f"{key}.wav"), "wb") as w:
                w.setnchannels(1), w.setsampwidth(2), w.setframerate(16000)
                w.writeframes(cli_wave(u, n).tobytes())
            lines.append(f"{key} {td / (key + '.wav')}")
        (td / "wav.scp").write_text("\n".join(lines) + "\n")
        ref_main(cmd=["--output_dir", str(td / "out"), "--ngpu", "0", "--dtype", "float32",
                      "--data_path_and_name_and_type", f"{td / 'wav.scp'},speech,sound",
                      "--asr_train_config", str(td / "asr" / "config.yaml"),
                      "--asr_model_file", str(td / "model.pth"), "--beam_size", str(beam),
                      "--ctc_weight", str(ctc_weight), "--nbest", str(nbest), "--lm_weight", "0.0",
                      "--batch_size", "1", "--num_workers", "0"])
        files = {}
        for f in sorted((td / "out").rglob("*")):
            if f.is_file():
                files[str(f.relative_to(td / "out"))] = f.read_text()
    np.savez_compressed(OUT / f"{name}.npz", config_yaml=np.array(cfg_text), vocab=np.array(vocab),
                        wseed=np.array(wseed), utts=np.array(json.dumps(utts)), beam=np.array(beam),
                        ctc_weight=np.array(ctc_weight), nbest=np.array(nbest),
                        state_shapes=np.array(json.dumps({k: list(v) for k, v in shapes.items()})),
                        melmat=s2t.asr_model.frontend.logmel.melmat.numpy(),
                        files=np.array(json.dumps(files)))
    print(f"[{name}] done in {time.time()-t0:.1f}s files={sorted(files)}")
    print(files.get("1best_recog/token_int", "")[:400])


def run_lm_search_case(name, conf, vocab, wseed, utt_id, n_samples, beam, ctc_weight, lm_weight, nbest,
                       lm_conf, lm_name="transformer"):
    """Speech2Text with a TransformerLM scorer (espnet2/bin/asr_inference.py:179-191, weight
    `lm_weight`; espnet2/lm/transformer_lm.py batch_score).  The LM config comes from an
    LMTask --dry_run (espnet2/tasks/lm.py), weights from the shared recipe (key prefix "lm.")."""
    from espnet2.bin.asr_inference import Speech2Text
    from espnet2.tasks.asr import ASRTask
    from espnet2.tasks.lm import LMTask

    t0 = time.time()
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        tok = td / "tokens.txt"
        tok.write_text("\n".join(token_list(vocab)) + "\n")
        (td / "train.yaml").write_text(yaml.safe_dump(conf))
        ASRTask.main(cmd=["--dry_run", "true", "--output_dir", str(td / "asr"), "--token_list", str(tok),
                          "--token_type", "word", "--config", str(td / "train.yaml")])
        (td / "lm.yaml").write_text(yaml.safe_dump(dict(lm=lm_name, lm_conf=lm_conf)))
        LMTask.main(cmd=["--dry_run", "true", "--output_dir", str(td / "lm"), "--token_list", str(tok),
                         "--token_type", "word", "--config", str(td / "lm.yaml")])
        s2t = Speech2Text(asr_train_config=str(td / "asr" / "config.yaml"), asr_model_file=None,
                          lm_train_config=str(td / "lm" / "config.yaml"), lm_file=None, device="cpu",
                          dtype="float32", beam_size=beam, ctc_weight=ctc_weight, lm_weight=lm_weight,
                          nbest=nbest, penalty=0.0, maxlenratio=0.0, minlenratio=0.0)
        cfg_text = (td / "asr" / "config.yaml").read_text()
    model = s2t.asr_model
    shapes = load_recipe(model, wseed)
    lm = s2t.beam_search.scorers["lm"]
    lm_shapes = {k: tuple(v.shape) for k, v in lm.state_dict().items()}
    lm_sd = recipe_state_dict({"lm." + k: v for k, v in lm_shapes.items()}, wseed, skip=())
    lm.load_state_dict({k[3:]: v for k, v in lm_sd.items()}, strict=True)
    lm.eval()
    wav = synth_waveform(utt_id, n_samples)
    enc, olens = model.encode(wav[None], torch.tensor([n_samples]))
    results = s2t(wav.numpy())
    L = max(len(h.yseq) for _, _, _, h in results)
    yseq = np.full((len(results), L), -1, dtype=np.int64)
    for i, (_, _, _, h) in enumerate(results):
        yseq[i, : len(h.yseq)] = h.yseq.numpy()
    keys = sorted(results[0][3].scores.keys())
    out = dict(config_yaml=np.array(cfg_text), vocab=np.array(vocab), wseed=np.array(wseed),
               utt_id=np.array(utt_id), n_samples=np.array(n_samples), beam=np.array(beam),
               ctc_weight=np.array(ctc_weight), lm_weight=np.array(lm_weight), nbest=np.array(nbest),
               lm_conf=np.array(json.dumps(lm_conf)), lm_name=np.array(lm_name),
               state_shapes=np.array(json.dumps({k: list(v) for k, v in shapes.items()})),
               lm_state_shapes=np.array(json.dumps({k: list(v) for k, v in lm_shapes.items()})),
               melmat=model.frontend.logmel.melmat.numpy(), enc_out=enc[0].numpy().copy(),
               enc_keep_every=np.array(1), yseq=yseq,
               yseq_lens=np.array([len(h.yseq) for _, _, _, h in results]),
               score=np.array([float(h.score) for _, _, _, h in results]),
               score_keys=np.array(json.dumps(keys)),
               scores=np.array([[float(h.scores[k]) for k in keys] for _, _, _, h in results]),
               token_int_best=np.array(results[0][2], dtype=np.int64))
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(f"[{name}] done in {time.time()-t0:.1f}s T={enc.shape[1]} best len={len(results[0][3].yseq)} "
          f"score={float(results[0][3].score):.4f} keys={keys}")


def stream_feats(utt_id, n_samples):
    """Deterministic encoder input for the streaming fixtures: log-mel of the synthetic waveform
    (oracle frontend, itself pinned against the reference) minus its per-utterance mean."""
    from oracle import conformer as oc
    from oracle.mel import slaney_mel_filterbank

    mel = torch.from_numpy(slaney_mel_filterbank(sr=16000, n_fft=512, n_mels=80, fmin=0, fmax=8000).T.copy())
    wav = synth_waveform(utt_id, n_samples)
    f, fl = oc.frontend_feats(wav[None], torch.tensor([n_samples]), mel, 512, 512, 128)
    return oc.utterance_mvn(f, fl)[0]


def run_streaming_case(name, enc_conf, wseed, utt_id, n_samples, chunk_frames, keep_every=1):
    """ContextualBlockConformerEncoder.forward_infer fed chunk by chunk
    (espnet2/asr/encoder/contextual_block_conformer_encoder.py:386-600), as
    Speech2TextStreaming.__call__ drives it (espnet2/bin/asr_inference_streaming.py:316-322)."""
    from espnet2.asr.encoder.contextual_block_conformer_encoder import ContextualBlockConformerEncoder

    t0 = time.time()
    enc = ContextualBlockConformerEncoder(input_size=80, **enc_conf)
    sd = enc.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    enc.load_state_dict(recipe_state_dict(shapes, wseed, skip=()), strict=True)
    enc.eval()
    feats = stream_feats(utt_id, n_samples)
    outs, lens, state = [], [], None
    pos = 0
    while pos < feats.size(0):
        nxt = min(feats.size(0), pos + chunk_frames)
        final = nxt == feats.size(0)
        y, _, state = enc(feats[None, pos:nxt], torch.tensor([nxt - pos]), state, is_final=final,
                          infer_mode=True)
        outs.append(y[0] if y.dim() == 3 else y)
        lens.append(int(outs[-1].size(0)))
        pos = nxt
    ys = torch.cat(outs, dim=0)
    # one-shot (is_final on the whole utterance) for the block-parallel path
    y1, _, _ = enc(feats[None], torch.tensor([feats.size(0)]), None, is_final=True, infer_mode=True)
    y1 = y1[0] if y1.dim() == 3 else y1
    out = dict(enc_conf=np.array(json.dumps(enc_conf)), wseed=np.array(wseed), utt_id=np.array(utt_id),
               n_samples=np.array(n_samples), chunk_frames=np.array(chunk_frames),
               state_shapes=np.array(json.dumps({k: list(v) for k, v in shapes.items()})),
               n_feat_frames=np.array(feats.size(0)), out_lens=np.array(lens),
               ys=ys[::keep_every].numpy().copy(), ys_oneshot=y1[::keep_every].numpy().copy(),
               keep_every=np.array(keep_every), ys_total=np.array(ys.size(0)),
               ys_oneshot_total=np.array(y1.size(0)))
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(f"[{name}] done in {time.time()-t0:.1f}s feats{tuple(feats.shape)} out lens {lens} "
          f"total {ys.size(0)} oneshot {y1.size(0)}")


def run_stream_search_case(name, vocab, wseed, utt_id, n_samples, chunk_samples, beam, ctc_weight, nbest,
                           tweaks=None, disable_repetition_detection=False, penalty=0.0, lm_conf=None,
                           lm_name="transformer", lm_weight=0.0, width=64, heads=1, peaked_seed=None,
                           qualify_noise=0.05):
    """Speech2TextStreaming end to end (espnet2/bin/asr_inference_streaming.py:293-336): waveform chunks
    -> apply_frontend -> ContextualBlockConformerEncoder.forward_infer -> BatchBeamSearchOnline.forward
    (espnet2/legacy/nets/batch_beam_search_online.py:155-534, block 40 / hop 16 / look-ahead 16).
    Stored per call: the encoder frames handed to the search and the n-best it returned."""
    import logging

    from espnet2.bin.asr_inference_streaming import Speech2TextStreaming

    t0 = time.time()
    conf = tiny(d=width, heads=heads, ff=128)
    conf["encoder"] = "contextual_block_conformer"
    conf["encoder_conf"] = dict(STREAM_TINY, output_size=width, attention_heads=heads)
    events = []

    class Grab(logging.Handler):
        def emit(self, rec):
            m = rec.getMessage()
            for key in ("Detected repetition", "reaching EOS in this block", "end detected at",
                        "no hypothesis. Finish", "adding <eos> in the last position"):
                if key in m:
                    events.append(key)

    with tempfile.TemporaryDirectory() as td:
        _, cfg_text = build_reference(conf, vocab, td, beam_size=1, ctc_weight=1.0)
        lm_cfg = None
        if lm_conf is not None:  # asr_inference_streaming.py:97-102 (scorers["lm"] = lm.lm)
            from espnet2.tasks.lm import LMTask

            (Path(td) / "lm.yaml").write_text(yaml.safe_dump(dict(lm=lm_name, lm_conf=lm_conf)))
            LMTask.main(cmd=["--dry_run", "true", "--output_dir", str(Path(td) / "lm"), "--token_list",
                             str(Path(td) / "tokens.txt"), "--token_type", "word", "--config", str(Path(td) / "lm.yaml")])
            lm_cfg = str(Path(td) / "lm" / "config.yaml")
        s2t = Speech2TextStreaming(asr_train_config=str(Path(td) / "asr" / "config.yaml"), asr_model_file=None,
                                   lm_train_config=lm_cfg, lm_file=None,
                                   device="cpu", dtype="float32", beam_size=beam, ctc_weight=ctc_weight,
                                   lm_weight=lm_weight, penalty=penalty, nbest=nbest,
                                   disable_repetition_detection=disable_repetition_detection)
    model = s2t.asr_model
    shapes = load_recipe(model, wseed, tweaks)
    lm_shapes = {}
    if lm_conf is not None:
        lm = s2t.beam_search.scorers["lm"]
        lm_shapes = {k: tuple(v.shape) for k, v in lm.state_dict().items()}
        lm_sd = recipe_state_dict({"lm." + k: v for k, v in lm_shapes.items()}, wseed, skip=())
        lm.load_state_dict({k[3:]: v for k, v in lm_sd.items()}, strict=True)
        lm.eval()
    h = Grab()
    logging.getLogger().addHandler(h)
    logging.getLogger().setLevel(logging.INFO)
    wav = synth_waveform(utt_id, n_samples)
    calls, enc_chunks = [], []
    orig_bs = s2t.beam_search.forward

    def spy(x, maxlenratio=0.0, minlenratio=0.0, is_final=True):
        enc_chunks.append(x.detach().clone())
        return orig_bs(x=x, maxlenratio=maxlenratio, minlenratio=minlenratio, is_final=is_final)

    s2t.beam_search.forward = spy

    def decode_stream():
        pos = 0
        while pos < n_samples:
            nxt = min(n_samples, pos + chunk_samples)
            n_before = len(enc_chunks)
            ev0 = len(events)
            res = s2t(wav[pos:nxt].numpy(), is_final=(nxt == n_samples))
            calls.append(dict(
                searched=len(enc_chunks) > n_before, events=events[ev0:],
                hyps=[dict(yseq=[int(v) for v in hy.yseq.tolist()], score=float(hy.score),
                           scores={k: float(v) for k, v in hy.scores.items()}) for _, _, _, hy in res]))
            pos = nxt

    decode_stream()
    extra = {}
    if peaked_seed is not None:
        # the encoder does not depend on the heads: take the frames of a first pass, fit both heads on them
        # (fit_peaked_search_heads: the decoder states are teacher-forced over the WHOLE memory; the online search
        # sees it block by block, so its costs are near, not at, the designed ones), decode the stream again
        enc_first = torch.cat(enc_chunks, 0)
        extra = fit_peaked_search_heads(model, enc_first[None], torch.tensor([enc_first.size(0)]), peaked_seed)
        calls.clear(), enc_chunks.clear(), events.clear()
        decode_stream()
        assert torch.equal(torch.cat(enc_chunks, 0), enc_first)
        final = calls[-1]["hyps"]
        sc = [hy["score"] for hy in final]
        print("  final n-best scores", [round(x, 3) for x in sc], "lens", [len(hy["yseq"]) for hy in final])
        # The block-synchronous search keeps what the offline one prunes (hypotheses that took a local <eos>, the
        # reference's duplicates at the end of a block), so the tail of its n-best is again decided among unfitted,
        # flat alternatives.  What is pinned is the SEPARATED HEAD of every call's list: the leading hypotheses up to
        # the first gap below 0.5 between different scores (exact duplicates count as one).
        def separated_head(hyps):
            n, i = 0, 0
            while i < len(hyps):
                j = i
                while j + 1 < len(hyps) and hyps[j + 1]["yseq"] == hyps[i]["yseq"]:
                    j += 1  # a group of exact duplicates
                if j + 1 < len(hyps) and hyps[i]["score"] - hyps[j + 1]["score"] < 0.5:
                    break
                n, i = j + 1, j + 1
            return n

        sep = [separated_head(c["hyps"]) for c in calls]
        print("  separated head per call:", sep)
        assert sep[-1] >= 1
        extra["sep_counts"] = np.array(sep)
        # qualification as for the offline fixture: the oracle's online search under noise returns the same separated
        # head at every call
        from oracle.beam_search_online import OnlineBeamSearchOracle

        sd_now = {k: v.detach().clone() for k, v in model.state_dict().items()}
        dcf = conf["decoder_conf"]
        for k in range(5):
            orc = OnlineBeamSearchOracle(sd_now, dcf["attention_heads"], dcf["num_blocks"], beam, ctc_weight,
                                         sos=vocab - 1, eos=vocab - 1, penalty=penalty,
                                         disable_repetition_detection=disable_repetition_detection,
                                         noise=qualify_noise if k else 0.0, noise_seed=k)
            p2 = 0
            for ci, (call, c) in enumerate(zip(calls, enc_chunks)):
                res = orc.forward(c, is_final=(ci == len(calls) - 1))[:nbest]
                got = [r["yseq"] for r in res][: sep[ci]]
                want = [hy["yseq"] for hy in call["hyps"]][: sep[ci]]
                assert got == want, f"online n-best moved under noise {qualify_noise if k else 0.0} (seed {k}, call {ci})"
        extra["noise_qualified"] = np.array(qualify_noise)
    logging.getLogger().removeHandler(h)
    enc_all = torch.cat(enc_chunks, 0) if enc_chunks else torch.zeros(0, 64)
    np.savez_compressed(
        OUT / f"{name}.npz", config_yaml=np.array(cfg_text), vocab=np.array(vocab), wseed=np.array(wseed),
        utt_id=np.array(utt_id), n_samples=np.array(n_samples), chunk_samples=np.array(chunk_samples),
        beam=np.array(beam), ctc_weight=np.array(ctc_weight), nbest=np.array(nbest), penalty=np.array(penalty),
        tweaks=np.array(json.dumps(tweaks or [])),
        disable_repetition_detection=np.array(disable_repetition_detection),
        state_shapes=np.array(json.dumps({k: list(v) for k, v in shapes.items()})),
        melmat=model.frontend.logmel.melmat.numpy(), enc_all=enc_all.numpy(),
        enc_lens=np.array([int(c.size(0)) for c in enc_chunks]), calls=np.array(json.dumps(calls)),
        lm_conf=np.array(json.dumps(lm_conf)), lm_name=np.array(lm_name), lm_weight=np.array(lm_weight),
        lm_state_shapes=np.array(json.dumps({k: list(v) for k, v in lm_shapes.items()})), **extra)
    print(f"[{name}] done in {time.time()-t0:.1f}s enc chunks {[int(c.size(0)) for c in enc_chunks]}")
    for k, c in enumerate(calls):
        print(f"   call {k}: searched={c['searched']} events={c['events']} n_hyps={len(c['hyps'])} "
              + (f"best len={len(c['hyps'][0]['yseq'])} score={c['hyps'][0]['score']:.3f}" if c["hyps"] else ""))


def run_stream_frontend_case(name, utt_id, n_samples, chunk_samples, use_global_mvn):
    """Speech2TextStreaming.apply_frontend (espnet2/bin/asr_inference_streaming.py:205-293) called
    as an unbound function on a stub carrying the attributes it reads, with the reference
    ESPnetASRModel (DefaultFrontend + UtteranceMVN or GlobalMVN)."""
    import types

    from espnet2.bin.asr_inference_streaming import Speech2TextStreaming

    t0 = time.time()
    conf = json.loads(json.dumps(tiny()))
    extra = {}
    with tempfile.TemporaryDirectory() as td:
        if use_global_mvn:
            g = torch.Generator().manual_seed(99)
            count = 1000.0
            mean = torch.randn(80, generator=g) * 2.0 - 8.0
            var = torch.rand(80, generator=g) * 4.0 + 1.0
            np.savez(Path(td) / "stats.npz", count=np.array(count), sum=(mean * count).numpy(),
                     sum_square=((var + mean * mean) * count).numpy())
            conf["normalize"] = "global_mvn"
            conf["normalize_conf"] = dict(stats_file=str(Path(td) / "stats.npz"))
        s2t, cfg_text = build_reference(conf, 50, td, beam_size=1, ctc_weight=1.0)
        model = s2t.asr_model
        if use_global_mvn:
            extra = dict(gmvn_mean=model.normalize.mean.numpy().copy(), gmvn_std=model.normalize.std.numpy().copy())
    fc = conf["frontend_conf"]
    stub = types.SimpleNamespace(asr_model=model, device="cpu", dtype="float32", n_fft=fc["n_fft"],
                                 hop_length=fc["hop_length"], win_length=fc["win_length"])
    wav = synth_waveform(utt_id, n_samples)
    feats, lens, state, pos = [], [], None, 0
    while pos < n_samples:
        nxt = min(n_samples, pos + chunk_samples)
        f, fl, state = Speech2TextStreaming.apply_frontend(stub, wav[pos:nxt], state, is_final=(nxt == n_samples))
        lens.append(-1 if f is None else int(f.size(1)))
        if f is not None:
            feats.append(f[0])
        pos = nxt
    out = dict(utt_id=np.array(utt_id), n_samples=np.array(n_samples), chunk_samples=np.array(chunk_samples),
               frontend_conf=np.array(json.dumps(fc)), use_global_mvn=np.array(use_global_mvn),
               melmat=model.frontend.logmel.melmat.numpy(), feat_lens=np.array(lens),
               feats=torch.cat(feats, 0).numpy(), **extra)
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print(f"[{name}] done in {time.time()-t0:.1f}s lens {lens}")


STREAM_SMALL = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12,
                    input_layer="conv2d", normalize_before=True, activation_type="swish",
                    macaron_style=True, use_cnn_module=True, cnn_module_kernel=15, block_size=40,
                    hop_size=16, look_ahead=16, init_average=True, ctx_pos_enc=True)
STREAM_TINY = dict(STREAM_SMALL, output_size=64, attention_heads=1, linear_units=128, num_blocks=2)

EBF_SMALL = dict(
    encoder="e_branchformer",
    encoder_conf=dict(output_size=256, attention_heads=4, attention_layer_type="rel_selfattn",
                      pos_enc_layer_type="rel_pos", rel_pos_type="latest", cgmlp_linear_units=1024,
                      cgmlp_conv_kernel=31, use_linear_after_conv=False, gate_activation="identity",
                      num_blocks=12, dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.1,
                      input_layer="conv2d", layer_drop_rate=0.0, linear_units=1024,
                      positionwise_layer_type="linear", use_ffn=True, macaron_ffn=True, merge_conv_kernel=31),
    decoder="transformer",
    decoder_conf=dict(attention_heads=4, linear_units=2048, num_blocks=6, dropout_rate=0.1,
                      positional_dropout_rate=0.1, self_attention_dropout_rate=0.1, src_attention_dropout_rate=0.1),
    model_conf=dict(ctc_weight=0.3, lsm_weight=0.1, length_normalized_loss=False),
    frontend_conf=dict(n_fft=512, win_length=400, hop_length=160),
)
EBF_TINY = json.loads(json.dumps(EBF_SMALL))
EBF_TINY["encoder_conf"].update(output_size=64, attention_heads=1, cgmlp_linear_units=128, linear_units=128,
                                num_blocks=2, cgmlp_conv_kernel=15, merge_conv_kernel=7)
EBF_TINY["decoder_conf"].update(attention_heads=1, linear_units=128, num_blocks=1)

BF_SMALL = json.loads(json.dumps(EBF_SMALL))
BF_SMALL["encoder"] = "branchformer"
BF_SMALL["encoder_conf"] = dict(output_size=256, use_attn=True, attention_heads=4, attention_layer_type="rel_selfattn",
                                pos_enc_layer_type="rel_pos", rel_pos_type="latest", use_cgmlp=True,
                                cgmlp_linear_units=1024, cgmlp_conv_kernel=31, use_linear_after_conv=False,
                                gate_activation="identity", merge_method="concat", num_blocks=6, dropout_rate=0.1,
                                positional_dropout_rate=0.1, attention_dropout_rate=0.1, input_layer="conv2d",
                                stochastic_depth_rate=0.0)
BF_TINY = json.loads(json.dumps(BF_SMALL))
BF_TINY["encoder_conf"].update(output_size=64, attention_heads=1, cgmlp_linear_units=128, num_blocks=2,
                               cgmlp_conv_kernel=7)
BF_TINY["decoder_conf"].update(attention_heads=1, linear_units=128, num_blocks=1)



def with_merge(conf, method, d=128, heads=2, blocks=2, **kw):
    """Branchformer config with another merge_method (branchformer_encoder.py:99-133)."""
    c = json.loads(json.dumps(conf))
    c["encoder_conf"].update(merge_method=method, output_size=d, attention_heads=heads, num_blocks=blocks,
                             cgmlp_linear_units=256, cgmlp_conv_kernel=15, **kw)
    return c


CASES = {
    # config 0/1 of BASELINE.json: Conformer-small, one 10 s utterance
    "small_10s": lambda: run_encode_case("small_10s", SMALL, 5000, 11, [0], [160000], keep_every=4),
    # ragged batch (padding quirks: zero-padded STFT tail, MVN over valid frames, unmasked conv)
    # the same weights and utterance with a ctc_lo FITTED to the reference's encoder output (peaked posteriors: the
    # bf16 path must return the reference's G1 tokens exactly)
    "small_10s_peaked": lambda: run_encode_case("small_10s_peaked", SMALL, 5000, 11, [0], [160000], keep_every=4,
                                                peaked_seed=5),
    # ... and with the fitted logits scaled down so that the reference's top-2 margins fill 0.05 .. 1.5: frames a
    # trained model finds hard, all above the bf16 flip bound
    "small_10s_midmargin": lambda: run_encode_case("small_10s_midmargin", SMALL, 5000, 11, [0], [160000], keep_every=4,
                                                   peaked_seed=5, peaked_level=(0.06, 1.5)),
    "small_ragged": lambda: run_encode_case("small_ragged", SMALL, 5000, 11, [1, 2, 3],
                                            [48000, 37123, 16000]),
    # tiny model with every block output pinned (oracle block-by-block check)
    "tiny_blocks": lambda: run_encode_case("tiny_blocks", tiny(), 50, 7, [4, 5], [16000, 9000],
                                           with_blocks=True),
    # large model encoder, 10 s
    "large_10s": lambda: run_encode_case("large_10s", LARGE, 5000, 13, [6], [160000], keep_every=8),
    # ... and with a ctc_lo FITTED to the reference's encoder output (peaked posteriors; 249 frames < 512 dimensions): the
    # 512-wide model's row-block launches + the CTC arg-max walk behind them must return the reference's ids exactly
    "large_10s_peaked": lambda: run_encode_case("large_10s_peaked", LARGE, 5000, 13, [6], [160000], keep_every=8,
                                                peaked_seed=9),
    # G2: width-1 CTC prefix search (decode_ctc_bs1.yaml), short audio
    "small_g2_3s": lambda: run_search_case("small_g2_3s", SMALL, 5000, 11, 7, 48000, 1, 1.0, 1,
                                           keep_every=2),
    # config 2/3: joint CTC/attention beam 10, ctc 0.3 (3 s and 10 s)
    "large_beam10_3s": lambda: run_search_case("large_beam10_3s", LARGE, 5000, 13, 8, 48000, 10,
                                               0.3, 10, keep_every=2),
    "large_beam10_10s": lambda: run_search_case("large_beam10_10s", LARGE, 5000, 13, 9, 160000,
                                                10, 0.3, 10, keep_every=8),
    # the same search with heads FITTED to a synthetic transcript (peaked posteriors, separated n-best: the bf16 search
    # must return the reference's n-best token sequences exactly)
    "large_beam10_3s_peaked": lambda: run_search_case("large_beam10_3s_peaked", LARGE, 5000, 13, 8, 48000, 10,
                                                      0.3, 10, keep_every=2, peaked_seed=3),
    # tiny joint search (fast unit-level fixture for the search restatement)
    "tiny_beam5": lambda: run_search_case("tiny_beam5", tiny(d=64, heads=2, ff=128), 50, 7, 10,
                                          24000, 5, 0.3, 5),
    "tiny_beam3_attn_only": lambda: run_search_case("tiny_beam3_attn_only",
                                                    tiny(d=64, heads=2, ff=128), 50, 7, 11,
                                                    16000, 3, 0.0, 3),
    # hypotheses that END EARLY (<eos> logit biased in decoder.output_layer): exercises the
    # ended-list bookkeeping, end_detect (e2e_asr_common.py:14-44) and a shrinking running beam;
    # plus length_bonus (penalty != 0) and a minlenratio bound
    "tiny_beam4_early_eos": lambda: run_search_case(
        "tiny_beam4_early_eos", tiny(d=64, heads=2, ff=128), 50, 7, 12, 32000, 4, 0.3, 4,
        penalty=0.3, tweaks=[["decoder.output_layer.bias", 49, 3.5]]),
    "tiny_beam4_minlen": lambda: run_search_case(
        "tiny_beam4_minlen", tiny(d=64, heads=2, ff=128), 50, 7, 13, 32000, 4, 0.1, 8,
        minlenratio=0.2, maxlenratio=0.6, tweaks=[["decoder.output_layer.bias", 49, 6.0]]),
    # config 5: contextual-block streaming Conformer (aishell recipe shape), 640 ms chunks
    "stream_small_6s": lambda: run_streaming_case("stream_small_6s", STREAM_SMALL, 17, 20, 96000, 80,
                                                  keep_every=4),
    # tiny: every frame kept; odd chunk size exercises both carry buffers; 0.3 s = short-utterance path
    "stream_tiny_4s": lambda: run_streaming_case("stream_tiny_4s", STREAM_TINY, 18, 21, 64000, 37),
    # Speech2TextStreaming.apply_frontend: 640 ms chunks with GlobalMVN, odd chunk size with UtteranceMVN
    "stream_frontend_gmvn": lambda: run_stream_frontend_case("stream_frontend_gmvn", 23, 48000, 10240, True),
    "stream_frontend_umvn": lambda: run_stream_frontend_case("stream_frontend_umvn", 24, 30000, 7001, False),
    # LM scorer fusion: joint CTC/attention + TransformerLM, lm_weight 0.6
    "tiny_beam5_lm": lambda: run_lm_search_case(
        "tiny_beam5_lm", tiny(d=64, heads=2, ff=128), 50, 7, 14, 24000, 5, 0.3, 0.6, 5,
        dict(pos_enc=None, embed_unit=32, att_unit=64, head=2, unit=128, layer=2)),
    "tiny_beam4_lm_posenc": lambda: run_lm_search_case(
        "tiny_beam4_lm_posenc", tiny(d=64, heads=2, ff=128), 50, 7, 15, 24000, 4, 0.5, 1.0, 4,
        dict(pos_enc="sinusoidal", embed_unit=64, att_unit=128, head=2, unit=128, layer=1)),
    # the LibriSpeech decode shape (egs2/librispeech/asr1/conf/decode_asr.yaml: beam 60, ctc 0.3, lm 0.6)
    # on a tiny model with a vocabulary wider than the pre-beam (int(1.5 * 60) = 90 < 300)
    "tiny_beam60_lm_v300": lambda: run_lm_search_case(
        "tiny_beam60_lm_v300", tiny(d=64, heads=2, ff=128), 300, 9, 16, 32000, 60, 0.3, 0.6, 20,
        dict(pos_enc=None, embed_unit=64, att_unit=64, head=2, unit=128, layer=2)),
    # end-to-end shape for the HIP encoder (d_k = 64): Speech2Text(lm_train_config, lm_file, lm_weight)
    "e2e_beam5_lm": lambda: run_lm_search_case(
        "e2e_beam5_lm", tiny(d=128, heads=2, ff=128), 50, 11, 17, 40000, 5, 0.3, 0.6, 5,
        dict(pos_enc=None, embed_unit=64, att_unit=128, head=2, unit=128, layer=2)),
    # decode CLI: ragged wav.scp incl. one utterance below the subsampling limit
    "cli_decode": lambda: run_cli_case(
        "cli_decode", tiny(d=128, heads=2, ff=128), 50, 13,
        [["uttA", 31, 28000], ["uttB", 32, 9000], ["uttC", 33, 500], ["uttD", 34, 41000],
         ["uttE", 35, 16000], ["uttF", 36, 23456], ["uttG", 37, 33000]], 3, 0.3, 2),
    # streaming decode: Speech2TextStreaming + BatchBeamSearchOnline, 640 ms chunks
    "stream_search_a": lambda: run_stream_search_case(
        "stream_search_a", 50, 21, 41, 80000, 10240, 4, 0.3, 4, tweaks=[["decoder.output_layer.bias", 49, 2.5]]),
    "stream_search_b": lambda: run_stream_search_case(
        "stream_search_b", 50, 22, 42, 64000, 10240, 5, 0.5, 3, tweaks=[["decoder.output_layer.bias", 49, 1.0]],
        disable_repetition_detection=True),
    "stream_search_c": lambda: run_stream_search_case(
        "stream_search_c", 50, 23, 43, 48000, 7000, 3, 0.3, 3, penalty=0.2),
    # E-Branchformer: attention + cgMLP branches, depthwise-conv merge
    "ebf_tiny_blocks": lambda: run_encode_case("ebf_tiny_blocks", EBF_TINY, 50, 31, [51, 52], [32000, 21000],
                                               with_blocks=True),
    "ebf_small_5s": lambda: run_encode_case("ebf_small_5s", EBF_SMALL, 5000, 32, [53, 54], [80000, 48000],
                                            keep_every=4),
    # SequentialRNNLM (LSTM) as the LM scorer (espnet2/lm/seq_rnn_lm.py; the LMTask default `lm: seq_rnn`)
    "tiny_beam5_rnnlm": lambda: run_lm_search_case(
        "tiny_beam5_rnnlm", tiny(d=64, heads=2, ff=128), 50, 7, 18, 28000, 5, 0.3, 0.7, 5,
        dict(unit=64, nlayers=2), lm_name="seq_rnn"),
    "tiny_beam4_rnnlm_nhid": lambda: run_lm_search_case(
        "tiny_beam4_rnnlm_nhid", tiny(d=64, heads=2, ff=128), 50, 7, 19, 24000, 4, 0.5, 1.0, 4,
        dict(unit=64, nhid=128, nlayers=1), lm_name="seq_rnn"),
    # the other rnn_type choices of SequentialRNNLM (seq_rnn_lm.py:40-58): GRU, tanh / relu RNN
    "tiny_beam5_gru": lambda: run_lm_search_case(
        "tiny_beam5_gru", tiny(d=64, heads=2, ff=128), 50, 7, 20, 28000, 5, 0.3, 0.7, 5,
        dict(unit=64, nlayers=2, rnn_type="gru"), lm_name="seq_rnn"),
    "tiny_beam4_gru_nhid": lambda: run_lm_search_case(
        "tiny_beam4_gru_nhid", tiny(d=64, heads=2, ff=128), 50, 7, 21, 24000, 4, 0.5, 1.0, 4,
        dict(unit=64, nhid=96, nlayers=1, rnn_type="gru"), lm_name="seq_rnn"),
    "tiny_beam4_rnn_tanh": lambda: run_lm_search_case(
        "tiny_beam4_rnn_tanh", tiny(d=64, heads=2, ff=128), 50, 7, 22, 24000, 4, 0.4, 0.8, 4,
        dict(unit=64, nlayers=2, rnn_type="rnn_tanh"), lm_name="seq_rnn"),
    "tiny_beam4_rnn_relu": lambda: run_lm_search_case(
        "tiny_beam4_rnn_relu", tiny(d=64, heads=2, ff=128), 50, 7, 23, 24000, 4, 0.4, 0.8, 4,
        dict(unit=64, nhid=128, nlayers=1, rnn_type="rnn_relu"), lm_name="seq_rnn"),
    # Branchformer (merge_method concat): the E-Branchformer layer without FFNs and merge conv
    "bf_tiny_blocks": lambda: run_encode_case("bf_tiny_blocks", BF_TINY, 50, 33, [55, 56], [30000, 17000],
                                              with_blocks=True),
    "bf_small_4s": lambda: run_encode_case("bf_small_4s", BF_SMALL, 5000, 34, [57, 58], [64000, 40000], keep_every=4),
    # Branchformer merge_method learned_ave (attention-pooled branch weights) and fixed_ave (per-block cgmlp_weight)
    "bf_learned_ave_4s": lambda: run_encode_case("bf_learned_ave_4s", with_merge(BF_SMALL, "learned_ave"), 50, 47,
                                                 [72, 73], [64000, 30000], with_blocks=True),
    "bf_fixed_ave_4s": lambda: run_encode_case("bf_fixed_ave_4s",
                                               with_merge(BF_SMALL, "fixed_ave", cgmlp_weight=[0.3, 0.65]), 50, 48,
                                               [74, 75], [64000, 35000], with_blocks=True),
    # other input layers: Conv2dSubsampling6 (5x5 stride-3 second conv) and Conv2dSubsampling8 (third conv)
    "sub6_small_6s": lambda: run_encode_case("sub6_small_6s", with_input_layer(tiny(d=128, heads=2, ff=128), "conv2d6"),
                                             50, 41, [61, 62], [96000, 50000], with_blocks=True),
    "sub8_small_6s": lambda: run_encode_case("sub8_small_6s", with_input_layer(tiny(d=128, heads=2, ff=128), "conv2d8"),
                                             50, 42, [63, 64], [96000, 41000], with_blocks=True),
    "ebf_sub6_4s": lambda: run_encode_case("ebf_sub6_4s", with_input_layer(EBF_SMALL, "conv2d6", blocks=2), 50, 43,
                                           [65, 66], [64000, 33000]),
    # rel_pos_type legacy (LegacyRelPositionMultiHeadedAttention + LegacyRelPositionalEncoding: older checkpoints)
    "legacy_small_5s": lambda: run_encode_case("legacy_small_5s", with_legacy(tiny(d=128, heads=2, ff=128)), 50, 44,
                                               [67, 68], [80000, 37000], with_blocks=True),
    "legacy_small_12s": lambda: run_encode_case("legacy_small_12s", with_legacy(tiny(d=128, heads=2, ff=128)), 50, 45,
                                                [69], [192000], keep_every=4),
    "ebf_legacy_4s": lambda: run_encode_case("ebf_legacy_4s", with_legacy(with_input_layer(EBF_SMALL, "conv2d", blocks=2)),
                                             50, 46, [70, 71], [64000, 29000]),
    # streaming decode with an LM scorer (asr_inference_streaming.py:97-102): TransformerLM and LSTM LM
    "stream_search_lm": lambda: run_stream_search_case(
        "stream_search_lm", 50, 24, 44, 64000, 10240, 4, 0.3, 3, tweaks=[["decoder.output_layer.bias", 49, 1.5]],
        disable_repetition_detection=True,
        lm_conf=dict(pos_enc=None, embed_unit=64, att_unit=64, head=1, unit=128, layer=2), lm_weight=0.5),
    "stream_search_rnnlm": lambda: run_stream_search_case(
        "stream_search_rnnlm", 50, 25, 45, 56000, 10240, 3, 0.4, 3, tweaks=[["decoder.output_layer.bias", 49, 2.0]],
        lm_conf=dict(unit=64, nlayers=1), lm_name="seq_rnn", lm_weight=0.6),
    "stream_search_gru": lambda: run_stream_search_case(
        "stream_search_gru", 50, 26, 46, 56000, 10240, 3, 0.4, 3, tweaks=[["decoder.output_layer.bias", 49, 2.0]],
        lm_conf=dict(unit=64, nlayers=2, rnn_type="gru"), lm_name="seq_rnn", lm_weight=0.6),
    # streaming decode on PEAKED posteriors (heads fitted to a transcript, as large_beam10_3s_peaked): the bf16 online
    # search must return the reference's n-best at every call
    "stream_search_peaked": lambda: run_stream_search_case(
        "stream_search_peaked", 300, 27, 47, 48000, 10240, 5, 0.3, 5, width=128, heads=2, peaked_seed=4),
    "stream_tiny_short": lambda: run_streaming_case("stream_tiny_short", STREAM_TINY, 18, 22, 4800, 1000),
}

def dump_cli_options(module: str, out_name: str):
    """Option table (dest -> default / required / choices) of a reference decode CLI's `get_parser()`."""
    import importlib

    parser = importlib.import_module(module).get_parser()
    table = {a.dest: dict(default=a.default, required=a.required, choices=list(a.choices) if a.choices else None)
             for a in parser._actions if a.dest != "help"}
    (HERE / out_name).write_text(json.dumps(table, indent=1, sort_keys=True, default=str) + "\n")
    print(f"[{out_name}] {len(table)} options")


API_TARGETS = [  # (reference dotted path, attribute) -> espnet_amd counterpart is listed in tests/test_cpu_host.py
    ("espnet2.bin.asr_inference", "Speech2Text.__init__"), ("espnet2.bin.asr_inference", "Speech2Text.__call__"),
    ("espnet2.bin.asr_inference", "inference"),
    ("espnet2.bin.asr_inference_streaming", "Speech2TextStreaming.__init__"),
    ("espnet2.bin.asr_inference_streaming", "Speech2TextStreaming.__call__"),
    ("espnet2.bin.asr_inference_streaming", "inference"),
    ("espnet2.asr.espnet_model", "ESPnetASRModel.__init__"), ("espnet2.asr.espnet_model", "ESPnetASRModel.encode"),
    ("espnet2.asr.frontend.default", "DefaultFrontend.__init__"),
    ("espnet2.asr.encoder.conformer_encoder", "ConformerEncoder.__init__"),
    ("espnet2.asr.encoder.e_branchformer_encoder", "EBranchformerEncoder.__init__"),
    ("espnet2.asr.encoder.branchformer_encoder", "BranchformerEncoder.__init__"),
    ("espnet2.asr.encoder.contextual_block_conformer_encoder", "ContextualBlockConformerEncoder.__init__"),
    ("espnet2.asr.decoder.transformer_decoder", "TransformerDecoder.__init__"),
    ("espnet2.asr.ctc", "CTC.__init__"),
    ("espnet2.lm.transformer_lm", "TransformerLM.__init__"), ("espnet2.lm.seq_rnn_lm", "SequentialRNNLM.__init__"),
    ("espnet2.legacy.nets.beam_search", "BeamSearch.__init__"),
    ("espnet2.legacy.nets.batch_beam_search_online", "BatchBeamSearchOnline.__init__"),
]


def dump_api_signatures():
    """Parameter names and (JSON-representable) defaults of the reference callables the drop-in mirrors."""
    import importlib
    import inspect

    table = {}
    for mod, attr in API_TARGETS:
        obj = importlib.import_module(mod)
        for part in attr.split("."):
            obj = getattr(obj, part)
        obj = inspect.unwrap(obj)
        params = {}
        for name, p in inspect.signature(obj).parameters.items():
            if name == "self" or p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD):
                continue
            d = p.default
            if d is inspect.Parameter.empty:
                params[name] = {"required": True}
            elif d is None or isinstance(d, (bool, int, float, str)) or (isinstance(d, (list, tuple, dict)) and not d):
                params[name] = {"required": False, "default": d if not isinstance(d, tuple) else list(d)}
            else:
                params[name] = {"required": False, "default_repr": repr(d)}
        table[f"{mod}:{attr}"] = params
    (HERE / "api_signatures.json").write_text(json.dumps(table, indent=1, sort_keys=True) + "\n")
    print(f"[api_signatures.json] {len(table)} callables")


CASES["api_signatures"] = dump_api_signatures
CASES["cli_options"] = lambda: dump_cli_options("espnet2.bin.asr_inference", "asr_inference_cli_options.json")
CASES["cli_options_streaming"] = lambda: dump_cli_options("espnet2.bin.asr_inference_streaming",
                                                          "asr_inference_streaming_cli_options.json")

if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 1)
    names = sys.argv[1:] or list(CASES)
    for n in names:
        CASES[n]()
