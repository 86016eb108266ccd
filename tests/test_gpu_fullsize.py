"""Size-independent properties at BASELINE.json's FULL bench sizes (the oracle takes minutes there, so these
sizes are not compared element by element): configs[1] Conformer-small, B = 32 x 10 s, greedy CTC and
configs[2] Conformer-large + 6-layer decoder, beam 10, B = 16 x 10 s — the same synthetic batches bench.py times.
Every check goes through the C-ABI product path (encode_device / greedy_ctc_device / search_batch)."""
import itertools

import pytest
import torch

pytestmark = pytest.mark.gpu

import bench  # noqa: E402


def _model(name, dtype):
    from espnet_amd.tasks.asr import ASRTask

    torch.manual_seed(0)
    return ASRTask.build_model(bench.model_config(name, dtype)).cuda().eval()


def test_greedy_b32_full_size_properties():
    B, N = 32, bench.N_SAMPLES
    wav = bench.synth_batch(0, B)
    model = _model("small", "float32")
    st = model.encode_device(wav.cuda(), [N] * B)
    ids, tokens, tlens = model.greedy_ctc_device(st)
    T = st.enc_out.size(1)
    assert T == 249 and st.olens == [T] * B and st.enc_out.shape == (B, T, 256)  # T_f 1001 -> T 249
    assert torch.isfinite(st.enc_out).all()
    ids_h, tok_h, tl_h = ids.cpu(), tokens.cpu(), tlens.cpu()
    V = model.vocab_size
    assert int(ids_h.min()) >= 0 and int(ids_h.max()) < V
    # (1) G1 definition (asr_inference.py:574-575) restated on the host over the device's frame arg-max:
    #     groupby, then drop blank (and <sos/eos>) -> must be the device's collapsed tokens, bit for bit
    for b in range(B):
        want = [k for k, _ in itertools.groupby(ids_h[b].tolist()) if k not in (model.blank_id, model.sos)]
        n = int(tl_h[b])
        assert tok_h[b, :n].tolist() == want and bool((tok_h[b, n:] == -1).all())
    # (2) determinism: the same batch again gives identical bits
    st2 = model.encode_device(wav.cuda(), [N] * B)
    assert torch.equal(st2.enc_out, st.enc_out)
    assert torch.equal(model.greedy_ctc_device(st2)[1], tokens)
    # (3) utterances are independent: a row of the batch == that utterance in a batch of one / of three
    #     (other GEMM tile shapes and row counts, same math: fp32 round-off only) and the same frame arg-max
    #     wherever the top-2 margin is not round-off
    logits = model.ctc.logits_device(st.enc_act).view(B, T, V)
    top2 = logits.topk(2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1]).cpu()
    for rows in ([5], [31, 0, 17]):
        sub = model.encode_device(wav[rows].cuda(), [N] * len(rows))
        sub_ids = model.greedy_ctc_device(sub)[0].cpu()
        for k, b in enumerate(rows):
            err = (sub.enc_out[k] - st.enc_out[b]).abs().max().item()
            assert err < 5e-4, (rows, b, err)
            diff = sub_ids[k] != ids_h[b]
            assert bool((margin[b][diff] < 1e-3).all()), (rows, b, int(diff.sum()))
    # (4) permutation equivariance of the batch dimension
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3))
    stp = model.encode_device(wav[perm].cuda(), [N] * B)
    assert (stp.enc_out - st.enc_out[perm.cuda()]).abs().max().item() < 5e-4
    # (5) bf16 MFMA mode against the fp32 mode of the same weights: relative error of the encoder output
    model.set_compute_dtype("bfloat16")
    stb = model.encode_device(wav.cuda(), [N] * B)
    rel = ((stb.enc_out - st.enc_out).norm() / st.enc_out.norm()).item()
    assert rel < 3e-2, rel
    tb = model.greedy_ctc_device(stb)
    assert int(tb[2].min()) >= 0 and int(tb[2].max()) <= T


def test_beam10_b16_full_size_properties():
    from espnet_amd.nets.batch_beam_search import build_beam_search

    B, N, W = 16, bench.N_SAMPLES, 10
    wav = bench.synth_batch(0, B)
    model = _model("large", "float32")
    bs = build_beam_search(model, beam_size=W, ctc_weight=0.3, penalty=0.0, token_list=model.token_list)
    st = model.encode_device(wav.cuda(), [N] * B)
    T = st.enc_out.size(1)
    assert T == 249 and st.enc_out.size(2) == 512
    nbest = bs.search_batch(st.enc_act, st.olens)
    assert len(nbest) == B
    eos, V = model.eos, model.vocab_size
    for hyps in nbest:
        assert 1 <= len(hyps)
        sc = [float(h.score) for h in hyps]
        assert sc == sorted(sc, reverse=True)  # n-best is sorted
        for h in hyps:
            y = h.yseq.tolist()
            assert y[0] == eos and y[-1] == eos and len(y) <= T + 2  # maxlenratio 0 -> maxlen = T (+ sos, eos)
            # <eos> only at the end; a hypothesis that picks <eos> at the very last step gets the forced one
            # appended as well (batch_beam_search.py:393-410), so the last two may both be <eos>
            assert all(0 <= t < V for t in y) and eos not in y[1:-2]
            tot = sum(bs.weights.get(k, 0.0) * float(v) for k, v in h.scores.items())  # score = sum of weighted scorers
            assert abs(tot - float(h.score)) < 1e-2 + 1e-4 * abs(tot)
            assert {"decoder", "ctc"} <= set(h.scores)
    # determinism
    again = bs.search_batch(st.enc_act, st.olens)
    assert [[h.yseq.tolist() for h in u] for u in again] == [[h.yseq.tolist() for h in u] for u in nbest]
    # utterances are independent inside the batched search: utterance b searched alone (10 rows instead of
    # 160: other GEMM kernels, fp32 round-off) ends at the same best score, and at the same tokens unless
    # the batch's own top-2 are close
    for b in (0, 9):
        alone = bs.search_batch(st.enc_act[b : b + 1].contiguous(), [T])[0]
        best = float(nbest[b][0].score)
        gap = best - float(nbest[b][1].score) if len(nbest[b]) > 1 else 10.0
        assert abs(float(alone[0].score) - best) < 0.5, (b, float(alone[0].score), best)
        if gap > 1.0 and abs(float(alone[0].score) - best) < 1e-2:
            assert alone[0].yseq.tolist() == nbest[b][0].yseq.tolist()
