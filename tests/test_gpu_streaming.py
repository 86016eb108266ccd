"""GPU parity of the streaming (contextual block) Conformer encoder step (SURVEY.md §8(a) A16,
BASELINE config 5) against fixtures produced by the reference class fed chunk by chunk."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import load_stream_golden, stream_feats  # noqa: E402


def build(g, dtype):
    from espnet_amd.asr.encoder.contextual_block_conformer_encoder import ContextualBlockConformerEncoder

    enc = ContextualBlockConformerEncoder(input_size=80, compute_dtype=dtype, **g["conf"])
    enc.load_state_dict(g["sd"], strict=True)
    return enc.cuda().eval()


def run_chunks(enc, feats, cf):
    outs, lens, state, pos = [], [], None, 0
    while pos < feats.size(0):
        nxt = min(feats.size(0), pos + cf)
        y, _, state = enc(feats[None, pos:nxt].cuda(), torch.tensor([nxt - pos]), state,
                          is_final=(nxt == feats.size(0)), infer_mode=True)
        outs.append(y[0])
        lens.append(int(y.size(1)))
        pos = nxt
    return torch.cat(outs, 0), lens


@pytest.mark.parametrize("name", ["stream_tiny_4s", "stream_tiny_short", "stream_small_6s"])
def test_streaming_encoder_f32_matches_reference(name):
    g = load_stream_golden(name)
    feats = stream_feats(int(g["utt_id"]), int(g["n_samples"]))
    enc = build(g, "float32")
    ys, lens = run_chunks(enc, feats, int(g["chunk_frames"]))
    assert lens == g["out_lens"].tolist()  # same frames emitted by every call
    ke = int(g["keep_every"])
    # f32 MFMA vs CPU fp32: summation-order round-off through 12 layers of O(1) activations
    np.testing.assert_allclose(ys.cpu()[::ke].numpy(), g["ys"], atol=2e-3, rtol=0)
    y1, _, _ = enc(feats[None].cuda(), torch.tensor([feats.size(0)]), None, is_final=True, infer_mode=True)
    np.testing.assert_allclose(y1[0].cpu()[::ke].numpy(), g["ys_oneshot"], atol=2e-3, rtol=0)


def test_streaming_encoder_bf16_within_tolerance():
    g = load_stream_golden("stream_small_6s")
    feats = stream_feats(int(g["utt_id"]), int(g["n_samples"]))
    enc = build(g, "bfloat16")
    ys, lens = run_chunks(enc, feats, int(g["chunk_frames"]))
    assert lens == g["out_lens"].tolist()
    err = np.abs(ys.cpu()[:: int(g["keep_every"])].numpy() - g["ys"])
    # LayerNorm'ed outputs are O(1); bf16 operands through 12 layers
    assert err.max() < 0.2 and err.mean() < 0.02, (err.max(), err.mean())


def test_streaming_chunking_invariance():
    """Size-independent property: the frames a streaming run emits do not depend on how the audio
    is cut into calls once a block is complete -- chunked == one-shot (f32, same kernels)."""
    g = load_stream_golden("stream_tiny_4s")
    feats = stream_feats(int(g["utt_id"]), int(g["n_samples"]))
    enc = build(g, "float32")
    a, _ = run_chunks(enc, feats, 37)
    b, _ = run_chunks(enc, feats, 64)
    c, _ = run_chunks(enc, feats, 10 ** 6)
    assert a.shape == b.shape == c.shape
    assert (a - b).abs().max().item() < 1e-4
    assert (a - c).abs().max().item() < 1e-4


def test_block_mha_kernel():
    from espnet_amd import lib as L

    lib = L.load()
    torch.manual_seed(0)
    n_blk, Lb, d, h = 3, 42, 256, 4
    dk = d // h
    qkv = torch.randn(n_blk * Lb, 3 * d)
    for dt, tdt, tol in ((L.EM_F32, torch.float32, 1e-5), (L.EM_BF16, torch.bfloat16, 2e-2)):
        qd = qkv.to(tdt).cuda()
        ctx = torch.empty(n_blk * Lb, d, dtype=tdt, device="cuda")
        for mode in (0, 1):
            L.check(lib.em_block_mha(dt, L.ptr(qd), n_blk, Lb, d, h, mode, L.ptr(ctx), None), "mha")
            torch.cuda.synchronize()
            q, k, v = qd.float().cpu().view(n_blk, Lb, 3, h, dk).permute(2, 0, 3, 1, 4)
            sc = q @ k.transpose(-1, -2) / dk ** 0.5
            if mode:
                m = torch.zeros(Lb, Lb, dtype=torch.bool)
                m[1:, : Lb - 1] = True
                sc = sc.masked_fill(~m, torch.finfo(torch.float32).min)
                ref = (torch.softmax(sc, -1).masked_fill(~m, 0.0) @ v)
            else:
                ref = torch.softmax(sc, -1) @ v
            ref = ref.permute(0, 2, 1, 3).reshape(n_blk * Lb, d)
            assert (ctx.float().cpu() - ref).abs().max().item() < tol, (dt, mode)


def test_streaming_step_hipgraph_replay_equals_eager():
    """BASELINE config 5: the steady-state chunk step replayed from a hipGraph gives exactly the
    eager results (same kernels, same data), including across the eager -> graph -> final hand-over."""
    from espnet_amd.asr.encoder.contextual_block_conformer_encoder import StreamingStepGraph

    g = load_stream_golden("stream_small_6s")
    feats = stream_feats(int(g["utt_id"]), int(g["n_samples"])).cuda()
    enc = build(g, "bfloat16")
    cf = 64  # 640 ms of 10 ms frames -> 16 encoder frames = one block per call
    eager, state, pos = [], None, 0
    while pos < feats.size(0):
        nxt = min(feats.size(0), pos + cf)
        y, _, state = enc.forward_infer(feats[None, pos:nxt], torch.tensor([nxt - pos]), state,
                                        nxt == feats.size(0))
        eager.append(y[0].clone())
        pos = nxt
    runner = StreamingStepGraph(enc, cf)
    for rep in range(2):  # second utterance reuses the captured graph
        got, pos = [], 0
        runner.reset()
        while pos < feats.size(0):
            nxt = min(feats.size(0), pos + cf)
            got.append(runner(feats[pos:nxt], is_final=(nxt == feats.size(0))).clone())
            pos = nxt
        assert runner.n_replays > 0
        assert [t.size(0) for t in got] == [t.size(0) for t in eager]
        for a, b in zip(got, eager):
            assert torch.equal(a, b)
