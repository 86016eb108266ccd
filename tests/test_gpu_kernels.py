"""GPU parity tests, kernel by kernel: every HIP kernel (called through the C ABI) against the
oracle's CPU-fp32 restatement on the same seeded inputs.

Tolerances (stated per SURVEY.md §8(c)):
  * EM_F32 (exact-f32 MFMA): only the summation order differs from the CPU -> atol/rtol 1e-4 class.
  * EM_BF16: operands rounded to bf16 (8 mantissa bits), f32 accumulate.  The reference value is
    computed in f32 FROM THE SAME bf16-ROUNDED OPERANDS, so what is checked is the kernel, not
    the rounding: 2e-2 relative to the output scale (output itself is rounded to bf16).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from espnet_amd import lib as L
from oracle import conformer as oc

pytestmark = pytest.mark.gpu
DT = {"f32": (L.EM_F32, torch.float32), "bf16": (L.EM_BF16, torch.bfloat16)}


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return L.load()


_KEEP = []


def dev(t):
    """Copy to the GPU and keep the tensor alive until the test ends: `L.ptr(dev(x))` hands a raw
    pointer to the C ABI, and a temporary would be freed (and its block re-used by the caching
    allocator) before the kernel runs."""
    t = t.contiguous().cuda()
    _KEEP.append(t)
    return t


@pytest.fixture(autouse=True)
def _release_kept_tensors():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def sptr():
    return L.current_stream_ptr()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(t, tdt):
    """Round to the activation dtype and come back to f32 (what the kernel actually sees)."""
    return t.to(tdt).to(torch.float32)


def assert_close(got, ref, tol, what=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


def gemm(lib, dt, epi, A, W, C, bias, M, N, K, lda, ldc, scale=1.0, amode=L.EM_A_PLAIN, conv=None):
    a = L.EmGemmArgs(A=A.data_ptr(), W=W.data_ptr(), C=C.data_ptr(),
                     bias=0 if bias is None else bias.data_ptr(), M=M, N=N, K=K, lda=lda, ldc=ldc,
                     scale=scale)
    if conv:
        a.T1, a.F1, a.T2, a.F2, a.d = conv
    L.check(lib.em_gemm(dt, epi, amode, a, sptr()), "em_gemm")


# --------------------------------------------------------------------------- MFMA layout
@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_gemm_identity_asymmetric(lib, prec):
    """A = I against an asymmetric W catches operand / output transposes (C must equal W^T)."""
    dt, tdt = DT[prec]
    n = 128
    A = torch.eye(n)
    W = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251) / 16.0  # exact in bf16? keep small ints/16
    W = q(W, tdt)
    C = torch.zeros(n, n, dtype=torch.float32, device="cuda")
    gemm(lib, dt, L.EM_EPI_STORE_F32, dev(A.to(tdt)), dev(W.to(tdt)), C, None, n, n, n, n, n)
    torch.cuda.synchronize()
    assert torch.equal(C.cpu(), W.t().contiguous())


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 256), (497, 768, 256), (1000, 5000, 256),
                                   (77, 256, 1024)])
def test_gemm_store_f32(lib, prec, M, N, K):
    dt, tdt = DT[prec]
    A, W, b = q(rnd(M, K, seed=1), tdt), q(rnd(N, K, seed=2, scale=K ** -0.5), tdt), rnd(N, seed=3)
    C = torch.full((M, N), 7.0, dtype=torch.float32, device="cuda")
    gemm(lib, dt, L.EM_EPI_STORE_F32, dev(A.to(tdt)), dev(W.to(tdt)), C, dev(b), M, N, K, K, N)
    ref = A @ W.t() + b
    assert_close(C, ref, 2e-5 if prec == "f32" else 2e-5, f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(300, 256, 128),  # tiled kernel
                                   (5, 64, 64), (42, 256, 2048), (48, 96, 256), (160, 1536, 512),  # skinny (M <= 48) / tiled
                                   (160, 512, 2048), (160, 512, 512), (640, 512, 2048), (77, 192, 192),  # gemm_mid.hip
                                   (50, 64, 64), (1000, 640, 256)])
def test_gemm_epilogues(lib, prec, M, N, K):
    dt, tdt = DT[prec]
    tol = 1e-5 if prec == "f32" else 1e-2
    A, W, b = q(rnd(M, K, seed=4), tdt), q(rnd(N, K, seed=5, scale=K ** -0.5), tdt), rnd(N, seed=6)
    Ad, Wd, bd = dev(A.to(tdt)), dev(W.to(tdt)), dev(b)
    lin = A @ W.t() + b
    for epi, fn in [(L.EM_EPI_STORE, lambda v: v), (L.EM_EPI_SWISH, oc.swish), (L.EM_EPI_RELU, F.relu)]:
        C = torch.zeros(M, N, dtype=tdt, device="cuda")
        gemm(lib, dt, epi, Ad, Wd, C, bd, M, N, K, K, N)
        assert_close(C, fn(lin), tol, f"epi {epi}")
    x0 = rnd(M, N, seed=7)
    C = dev(x0.clone())
    gemm(lib, dt, L.EM_EPI_RESID_F32, Ad, Wd, C, bd, M, N, K, K, N, scale=0.5)
    assert_close(C, x0 + 0.5 * lin, 1e-5, "resid")
    C = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    gemm(lib, dt, L.EM_EPI_SCALE_F32, Ad, Wd, C, bd, M, N, K, K, N, scale=16.0)
    assert_close(C, 16.0 * lin, 1e-5, "scale")
    # GLU: rows interleaved in 16-row granules [v | g]
    d = N // 2
    perm = torch.arange(N).reshape(2, d // 16, 16).permute(1, 0, 2).reshape(-1)
    C = torch.zeros(M, d, dtype=tdt, device="cuda")
    gemm(lib, dt, L.EM_EPI_GLU, Ad, dev(W[perm].to(tdt)), C, dev(b[perm]), M, N, K, K, d)
    assert_close(C, F.glu(lin, dim=1), tol, "glu")


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_conv2_implicit_gemm(lib, prec):
    dt, tdt = DT[prec]
    B, T1, F1, d = 2, 23, 39, 64
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    x = q(rnd(B, d, T1, F1, seed=8), tdt)           # NCHW for the reference
    w = q(rnd(d, d, 3, 3, seed=9, scale=(9 * d) ** -0.5), tdt)
    b = rnd(d, seed=10)
    ref = F.relu(F.conv2d(x, w, b, stride=2))       # (B, d, T2, F2)
    xcl = dev(x.permute(0, 2, 3, 1).to(tdt))        # (B, T1, F1, d) channel-last
    wp = dev(w.permute(0, 2, 3, 1).reshape(d, 9 * d).to(tdt))
    C = torch.zeros(B * T2 * F2, d, dtype=tdt, device="cuda")
    gemm(lib, dt, L.EM_EPI_RELU, xcl, wp, C, dev(b), B * T2 * F2, d, 9 * d, 0, d, amode=L.EM_A_CONV2,
         conv=(T1, F1, T2, F2, d))
    got = C.float().cpu().reshape(B, T2, F2, d).permute(0, 3, 1, 2)
    assert_close(got, ref, 2e-5 if prec == "f32" else 1e-2, "conv2")


# --------------------------------------------------------------------------- frontend
def _frontend(lib, speech, lens, melmat, win_length, hop):
    from espnet_amd.asr.frontend.default import DefaultFrontend

    fe = DefaultFrontend(n_fft=512, win_length=win_length, hop_length=hop)
    fe.logmel.melmat.copy_(melmat)
    fe.cuda()
    return fe(dev(speech), lens)


@pytest.mark.parametrize("win_length,hop", [(400, 160), (None, 160), (512, 128)])
def test_frontend_logmel(lib, win_length, hop):
    from oracle.mel import slaney_mel_filterbank
    from oracle.weights import synth_waveform

    melmat = torch.from_numpy(slaney_mel_filterbank(16000, 512, 80, 0, 8000).T.copy())
    lens = torch.tensor([16000, 12345, 5000])
    speech = torch.zeros(3, 16000)
    for i, n in enumerate(lens.tolist()):
        speech[i, :n] = synth_waveform(20 + i, n)
    ref, rlens = oc.frontend_feats(speech, lens, melmat, 512, win_length, hop)
    got, glens = _frontend(lib, speech, lens, melmat, win_length, hop)
    assert glens.tolist() == rlens.tolist()
    # log-mel of N(0, 0.1^2) noise: abs 1e-4; the log amplifies the relative f32
    # FFT error of low-power bins, hence absolute tolerance on the log value.
    err = (got.cpu() - ref).abs().max().item()
    assert err < 2e-4, err


@pytest.mark.parametrize("n_mels,hop,isolate", [(80, 160, False), (80, 128, True), (40, 160, False), (23, 160, True), (80, 97, False)])
def test_frontend_v2_equals_v1(lib, n_mels, hop, isolate):
    """Round 5: the frontend kernel that walks eight frames per wave (window, twiddles and the mel bands in registers,
    wave-private padded LDS tiles) must give the bits of the one-frame-per-wave kernel of rounds 1-4
    (ESPNET_AMD_FRONTEND_V1=1): same operations in the same order.  Ragged lengths (reflection at the utterance's own end
    with `isolate`), mel banks with other band widths, an odd hop (the unaligned load path)."""
    import os

    from oracle.mel import slaney_mel_filterbank
    from oracle.weights import synth_waveform

    melmat = torch.from_numpy(slaney_mel_filterbank(16000, 512, n_mels, 0, 8000).T.copy())
    lens = [16000, 12345, 5000, 700]
    nmax = 16000 + (1 if hop == 97 else 0)  # (an odd row pitch: every other utterance starts on an odd sample)
    speech = torch.zeros(len(lens), nmax)
    for i, n in enumerate(lens):
        speech[i, :n] = synth_waveform(40 + i, n)
    from espnet_amd.asr.frontend.default import DefaultFrontend

    fe = DefaultFrontend(fs=16000, n_fft=512, win_length=400, hop_length=hop, n_mels=n_mels)
    fe.logmel.melmat.copy_(melmat)
    fe.cuda()
    sp = dev(speech)
    flens = dev(torch.tensor(fe.feature_lengths(lens), dtype=torch.int32))
    wlens = dev(torch.tensor(lens, dtype=torch.int32)) if isolate else None
    outs = []
    for v1 in (False, True):
        if v1:
            os.environ["ESPNET_AMD_FRONTEND_V1"] = "1"
        L.load().em_dev_switches_reload()  # (the library reads its developer switches once; this test flips one in-process)
        try:
            outs.append(fe.forward_device(sp, flens, wlens).clone())
            torch.cuda.synchronize()
        finally:
            os.environ.pop("ESPNET_AMD_FRONTEND_V1", None)
            L.load().em_dev_switches_reload()
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]), (outs[0] - outs[1]).abs().max().item()


def test_utt_mvn_and_conv1(lib):
    B, T_f, D, d = 3, 61, 80, 64
    feats = rnd(B, T_f, D, seed=11) * 2 - 8
    flens = torch.tensor([61, 40, 7])
    feats = feats.masked_fill(oc.make_pad_mask(flens, T_f)[:, :, None], 0.0)
    fl = dev(flens.to(torch.int32))
    partial = torch.empty(B, 8, D, device="cuda")
    L.check(lib.em_utt_mvn_partial_f32(L.ptr(dev(feats)), L.ptr(fl), B, T_f, D, L.ptr(partial), sptr()))
    mean = partial.sum(1).cpu() / flens[:, None].float()
    ref_mean = feats.sum(1) / flens[:, None].float()
    assert_close(mean, ref_mean, 1e-6, "mvn mean")
    w1, b1 = rnd(d, 1, 3, 3, seed=12, scale=1 / 3), rnd(d, seed=13, scale=0.1)
    normed = oc.utterance_mvn(feats, flens)
    ref = F.relu(F.conv2d(normed.unsqueeze(1), w1, b1, stride=2)).permute(0, 2, 3, 1)  # (B,T1,F1,d)
    T1, F1 = ref.shape[1], ref.shape[2]
    for prec, tol in (("f32", 1e-5), ("bf16", 1e-2)):
        dt, tdt = DT[prec]
        out = torch.zeros(B, T1, F1, d, dtype=tdt, device="cuda")
        L.check(lib.em_conv2d_sub1(dt, L.ptr(dev(feats)), L.ptr(partial), L.ptr(fl), B, T_f, D,
                                   L.ptr(dev(w1.reshape(d, 9))), L.ptr(dev(b1)), d, L.ptr(out), sptr()))
        assert_close(out, ref, tol, f"conv1 {prec}")


@pytest.mark.parametrize("d", [256, 512])
@pytest.mark.parametrize("T_f,D,with_mvn", [(61, 80, True), (133, 80, False), (47, 40, True), (7, 80, True)])
def test_conv2d_sub12_fused(lib, T_f, D, with_mvn, d):
    """Conv2dSubsampling's two convolutions in one kernel (csrc/subsample2.hip; conv1 on the matrix cores with split-bf16
    operands, the conv1 map only in LDS) against torch's f32 convolutions.

    What is asserted: (1) the usual bf16 bound against the reference whose conv1 output is rounded to bf16 (what conv2
    sees in either device path); (2) against the EXACT f32 reference (no intermediate rounding) the fused kernel is as
    accurate as the two-launch device path (em_conv2d_sub1 + implicit-GEMM conv2): its split-operand conv1 carries ~17
    bits, so ~0.2 % of the conv1 values land on the other side of a bf16 rounding boundary than the f32 conv1's do -
    one more bf16 ulp on those, nothing against the half ulp every value gets from the rounding itself (measured:
    profiles/r03k_sub12_accuracy.txt)."""
    from espnet_amd.asr.encoder.conformer_encoder import pack_conv1_frags, pack_conv2_frags

    B = 3  # d = 512 (round 4): one launch per 256 output channels, 16 chunks of input channels
    flens = torch.tensor([T_f, max(7, T_f - 9), 7])
    feats = (rnd(B, T_f, D, seed=31) * 2 - 8).masked_fill(oc.make_pad_mask(flens, T_f)[:, :, None], 0.0)
    fl = dev(flens.to(torch.int32))
    fd = dev(feats)
    partial = torch.empty(B, 8, D, device="cuda")
    L.check(lib.em_utt_mvn_partial_f32(L.ptr(fd), L.ptr(fl), B, T_f, D, L.ptr(partial), sptr()))
    pp = L.ptr(partial) if with_mvn else None
    w1, b1 = rnd(d, 1, 3, 3, seed=32, scale=1 / 3), rnd(d, seed=33, scale=0.1)
    w2 = q(rnd(d, d, 3, 3, seed=34, scale=(9 * d) ** -0.5), torch.bfloat16)
    b2 = rnd(d, seed=35, scale=0.1)
    x = oc.utterance_mvn(feats, flens) if with_mvn else feats
    c1 = F.relu(F.conv2d(x.unsqueeze(1), w1, b1, stride=2))
    exact = F.relu(F.conv2d(c1, w2, b2, stride=2)).permute(0, 2, 3, 1)                              # (B, T2, F2, d)
    ref = F.relu(F.conv2d(q(c1, torch.bfloat16), w2, b2, stride=2)).permute(0, 2, 3, 1)             # bf16 conv1 map
    T1, F1, T2, F2 = c1.shape[2], c1.shape[3], ref.shape[1], ref.shape[2]
    out = torch.full((B, T2, F2, d), 7.0, dtype=torch.bfloat16, device="cuda")
    w1f = dev(pack_conv1_frags(w1.reshape(d, 9), b1).to(torch.bfloat16))
    w2p = w2.permute(0, 2, 3, 1).reshape(d, 9 * d)
    w2f = dev(pack_conv2_frags(w2p).to(torch.bfloat16))
    b2d = dev(b2)
    L.check(lib.em_conv2d_sub12_bf16(L.ptr(fd), pp, L.ptr(fl), B, T_f, D, L.ptr(w1f), L.ptr(w2f), L.ptr(b2d), d,
                                     L.ptr(out), sptr()), "sub12")
    # the two-launch device path on the same operands
    c1d = torch.zeros(B, T1, F1, d, dtype=torch.bfloat16, device="cuda")
    L.check(lib.em_conv2d_sub1(L.EM_BF16, L.ptr(fd), pp, L.ptr(fl), B, T_f, D, L.ptr(dev(w1.reshape(d, 9))),
                               L.ptr(dev(b1)), d, L.ptr(c1d), sptr()))
    out2 = torch.zeros(B * T2 * F2, d, dtype=torch.bfloat16, device="cuda")
    gemm(lib, L.EM_BF16, L.EM_EPI_RELU, c1d, dev(w2p.to(torch.bfloat16)), out2, b2d, B * T2 * F2, d, 9 * d, 0, d,
         amode=L.EM_A_CONV2, conv=(T1, F1, T2, F2, d))
    torch.cuda.synchronize()
    assert_close(out, ref, 1e-2, "fused conv1 + conv2")
    o1, o2 = out.float().cpu(), out2.float().cpu().reshape(B, T2, F2, d)
    rms1 = (o1 - exact).pow(2).mean().sqrt().item()
    rms2 = (o2 - exact).pow(2).mean().sqrt().item()
    print(f"rms error against the exact f32 result: fused {rms1:.4e}, two launches {rms2:.4e}")
    assert rms1 <= 1.05 * rms2 + 1e-6, (rms1, rms2)
    # nothing was written past the valid rows (the last tile overhangs T2)
    assert torch.isfinite(o1).all()


# --------------------------------------------------------------------------- norm / conv / attention
@pytest.mark.parametrize("d", [64, 256, 512])
def test_layernorm(lib, d):
    M = 77
    x = rnd(M, d, seed=14) * 3 + 1
    g1, b1, g2, b2 = 1 + 0.1 * rnd(d, seed=15), 0.1 * rnd(d, seed=16), 1 + 0.1 * rnd(d, seed=17), 0.1 * rnd(d, seed=18)
    ref1 = F.layer_norm(x, (d,), g1, b1, 1e-12)
    ref2 = F.layer_norm(ref1, (d,), g2, b2, 1e-12)
    for prec, tol in (("f32", 2e-6), ("bf16", 1e-2)):
        dt, tdt = DT[prec]
        out = torch.zeros(M, d, dtype=tdt, device="cuda")
        of = torch.zeros(M, d, device="cuda")
        L.check(lib.em_layernorm(dt, L.ptr(dev(x)), L.ptr(dev(g1)), L.ptr(dev(b1)), M, d, 1e-12,
                                 L.ptr(out), L.ptr(of), sptr()))
        assert_close(out, ref1, tol, "ln")
        assert_close(of, ref1, 2e-6, "ln f32 copy")
        xd = dev(x.clone())
        L.check(lib.em_layernorm2(dt, L.ptr(xd), L.ptr(dev(g1)), L.ptr(dev(b1)), L.ptr(dev(g2)),
                                  L.ptr(dev(b2)), M, d, 1e-12, L.ptr(out), L.ptr(of), sptr()))
        assert_close(xd, ref1, 2e-6, "ln2 in-place")
        assert_close(out, ref2, tol, "ln2 out")
        assert_close(of, ref2, 4e-6, "ln2 f32")


@pytest.mark.parametrize("k", [31, 15])
def test_dwconv_bn_swish(lib, k):
    B, T, d = 2, 50, 128
    x = rnd(B, T, d, seed=19)
    w, b = rnd(d, 1, k, seed=20, scale=k ** -0.5), rnd(d, seed=21, scale=0.1)
    gam, bet = 1 + 0.1 * rnd(d, seed=22), 0.1 * rnd(d, seed=23)
    mean, var = 0.1 * rnd(d, seed=24), torch.rand(d, generator=torch.Generator().manual_seed(25)) + 0.5
    scale = gam / torch.sqrt(var + 1e-5)
    wf, bf = (w.reshape(d, k) * scale[:, None]), (b - mean) * scale + bet
    for prec, tol in (("f32", 2e-6), ("bf16", 1e-2)):
        dt, tdt = DT[prec]
        xq = q(x, tdt)
        y = F.conv1d(xq.transpose(1, 2), w, b, padding=(k - 1) // 2, groups=d)
        y = F.batch_norm(y, mean, var, gam, bet, False, 0.0, 1e-5)
        ref = oc.swish(y).transpose(1, 2)
        out = torch.zeros(B, T, d, dtype=tdt, device="cuda")
        L.check(lib.em_dwconv_bn_swish(dt, L.ptr(dev(xq.to(tdt))), L.ptr(dev(wf.t().contiguous())), L.ptr(dev(bf)), None, B, T,
                                       d, k, L.ptr(out), sptr()))
        assert_close(out, ref, tol, f"dwconv {prec}")
        # with lengths: frames t >= tlens[b] read as zero (the utterance decoded alone)
        tl = [T, max(1, T // 3)][:B] + [T] * max(0, B - 2)
        xm = xq.clone()
        for bi, n in enumerate(tl):
            xm[bi, n:] = 0
        ym = F.conv1d(xm.transpose(1, 2), w, b, padding=(k - 1) // 2, groups=d)
        ym = oc.swish(F.batch_norm(ym, mean, var, gam, bet, False, 0.0, 1e-5)).transpose(1, 2)
        out2 = torch.zeros(B, T, d, dtype=tdt, device="cuda")
        L.check(lib.em_dwconv_bn_swish(dt, L.ptr(dev(xq.to(tdt))), L.ptr(dev(wf.t().contiguous())), L.ptr(dev(bf)),
                                       L.ptr(torch.tensor(tl, dtype=torch.int32, device="cuda")), B, T, d, k,
                                       L.ptr(out2), sptr()))
        for bi, n in enumerate(tl):
            assert_close(out2[bi, :n], ym[bi, :n], tol, f"dwconv masked {prec}")


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("T,klens", [(74, [74, 59, 26]), (249, [249, 100]), (130, [130, 1])])
def test_relpos_attention(lib, prec, T, klens):
    dt, tdt = DT[prec]
    B, h, dk = len(klens), 2, 64
    d = h * dk
    qkv = q(rnd(B, T, 3 * d, seed=26), tdt)
    p = q(rnd(2 * T - 1, d, seed=27), tdt)
    u, v = rnd(h, dk, seed=28, scale=0.3), rnd(h, dk, seed=29, scale=0.3)
    qq = qkv[..., :d].reshape(B, T, h, dk)
    kk = qkv[..., d:2 * d].reshape(B, T, h, dk).transpose(1, 2)
    vv = qkv[..., 2 * d:].reshape(B, T, h, dk).transpose(1, 2)
    pp = p.reshape(1, 2 * T - 1, h, dk).transpose(1, 2)
    q_u = q(qq + u, tdt).transpose(1, 2)
    q_v = q(qq + v, tdt).transpose(1, 2)
    ac = q_u @ kk.transpose(-2, -1)
    bd = oc.rel_shift(q_v @ pp.transpose(-2, -1))
    scores = (ac + bd) / math.sqrt(dk)
    valid = ~oc.make_pad_mask(torch.tensor(klens), T)
    mask = ~valid[:, None, None, :]
    attn = torch.softmax(scores.masked_fill(mask, torch.finfo(torch.float32).min), -1).masked_fill(mask, 0.0)
    ref = (attn @ vv).transpose(1, 2).reshape(B, T, d)
    ctx = torch.zeros(B, T, d, dtype=tdt, device="cuda")
    L.check(lib.em_relpos_attention(dt, L.ptr(dev(qkv.to(tdt))), L.ptr(dev(p.to(tdt))), d, L.ptr(dev(u)),
                                    L.ptr(dev(v)), L.ptr(dev(torch.tensor(klens, dtype=torch.int32))),
                                    B, T, h, dk, L.ptr(ctx), sptr()))
    assert_close(ctx, ref, 2e-5 if prec == "f32" else 2e-2, f"attention {prec} T={T}")


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("T,klens", [(49, [49, 20]), (64, [64]), (65, [65, 64]), (249, [249, 130, 1]), (300, [300])])
def test_legacy_relpos_attention(lib, prec, T, klens):
    """LegacyRelPositionMultiHeadedAttention core (attention.py:318-360) incl. the wrapped upper part of its
    square rel_shift, against the literal pad-and-reshape on the same (rounded) operands."""
    dt, tdt = DT[prec]
    B, h, dk = len(klens), 2, 64
    d = h * dk
    qkv = q(rnd(B, T, 3 * d, seed=36), tdt)
    p = q(rnd(T, d, seed=37), tdt)
    u, v = rnd(h, dk, seed=38, scale=0.3), rnd(h, dk, seed=39, scale=0.3)
    qq = qkv[..., :d].reshape(B, T, h, dk)
    kk = qkv[..., d:2 * d].reshape(B, T, h, dk).transpose(1, 2)
    vv = qkv[..., 2 * d:].reshape(B, T, h, dk).transpose(1, 2)
    pp = p.reshape(1, T, h, dk).transpose(1, 2)
    q_u = q(qq + u, tdt).transpose(1, 2)
    q_v = q(qq + v, tdt).transpose(1, 2)
    ac = q_u @ kk.transpose(-2, -1)
    bd = oc.legacy_rel_shift(q_v @ pp.transpose(-2, -1))
    scores = (ac + bd) / math.sqrt(dk)
    valid = ~oc.make_pad_mask(torch.tensor(klens), T)
    mask = ~valid[:, None, None, :]
    attn = torch.softmax(scores.masked_fill(mask, torch.finfo(torch.float32).min), -1).masked_fill(mask, 0.0)
    ref = (attn @ vv).transpose(1, 2).reshape(B, T, d)
    ctx = torch.zeros(B, T, d, dtype=tdt, device="cuda")
    L.check(lib.em_legacy_relpos_attention(dt, L.ptr(dev(qkv.to(tdt))), L.ptr(dev(p.to(tdt))), d, L.ptr(dev(u)),
                                           L.ptr(dev(v)), L.ptr(dev(torch.tensor(klens, dtype=torch.int32))),
                                           B, T, h, dk, L.ptr(ctx), sptr()))
    assert_close(ctx, ref, 2e-5 if prec == "f32" else 2e-2, f"legacy attention {prec} T={T}")


# --------------------------------------------------------------------------- CTC head
def test_argmax_logsoftmax_collapse(lib):
    M, V = 333, 5000
    x = rnd(M, V, seed=30)
    x[5, 17] = x[5, 4000] = 9.0  # tie -> lowest index
    xd = dev(x)
    ids = torch.zeros(M, dtype=torch.int32, device="cuda")
    L.check(lib.em_argmax_rows_f32(L.ptr(xd), M, V, L.ptr(ids), sptr()))
    assert torch.equal(ids.cpu().long(), torch.argmax(x, dim=1))
    assert ids[5].item() == 17
    L.check(lib.em_log_softmax_rows_f32(L.ptr(xd), M, V, sptr()))
    assert_close(xd, F.log_softmax(x, dim=1), 1e-6, "log_softmax")
    B, T = 4, 150
    g = torch.Generator().manual_seed(31)
    idm = torch.randint(0, 4, (B, T), generator=g, dtype=torch.int32)
    idm[idm == 3] = 49
    olens = torch.tensor([150, 64, 65, 1], dtype=torch.int32)
    tok = torch.zeros(B, T, dtype=torch.int32, device="cuda")
    tl = torch.zeros(B, dtype=torch.int32, device="cuda")
    L.check(lib.em_ctc_collapse(L.ptr(dev(idm)), L.ptr(dev(olens)), B, T, 0, 49, L.ptr(tok), L.ptr(tl), sptr()))
    for b in range(B):
        ref = oc.g1_collapse(idm[b, : olens[b]].tolist(), (0, 49))
        assert tl[b].item() == len(ref)
        assert tok[b, : len(ref)].cpu().tolist() == ref
        assert (tok[b, len(ref):] == -1).all()


def test_layernorm_inplace(lib):
    M, d = 300, 256
    x = rnd(M, d, seed=60)
    g, b = 1 + 0.1 * rnd(d, seed=61), 0.1 * rnd(d, seed=62)
    xd = dev(x.clone())
    L.check(lib.em_layernorm_inplace_f32(L.ptr(xd), L.ptr(dev(g)), L.ptr(dev(b)), M, d, 1e-12, sptr()))
    assert_close(xd, F.layer_norm(x, (d,), g, b, 1e-12), 2e-6, "ln inplace")


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(37, 200, 64), (160, 1536, 512), (5, 5000, 256), (640, 96, 1024)])
def test_ln_gemm_small_m(lib, prec, M, N, K):
    """LayerNorm fused into the consuming projection (decoder / LM step): against torch fp32 on the
    operand-dtype-rounded LayerNorm output, for the three epilogues the search uses."""
    dt, tdt = DT[prec]
    x = rnd(M, K, seed=31) * 2 + 0.3
    g, be = 1 + 0.1 * rnd(K, seed=32), 0.1 * rnd(K, seed=33)
    w = q(rnd(N, K, seed=34, scale=K ** -0.5), tdt)
    bias = 0.1 * rnd(N, seed=35)
    xn = q(F.layer_norm(x, (K,), g, be, 1e-12), tdt)
    ref = F.linear(xn, w, bias)
    tol = 3e-5 if prec == "f32" else 3e-2
    xd, gd, bd, wd, biasd = dev(x), dev(g), dev(be), dev(w.to(tdt)), dev(bias)
    for epi, r, odt in ((L.EM_EPI_STORE, ref, tdt), (L.EM_EPI_RELU, torch.relu(ref), tdt),
                        (L.EM_EPI_STORE_F32, ref, torch.float32)):
        out = torch.zeros(M, N, dtype=odt, device="cuda")
        L.check(lib.em_ln_gemm(dt, epi, L.ptr(xd), L.ptr(gd), L.ptr(bd), 1e-12, L.ptr(wd), L.ptr(biasd), L.ptr(out),
                               M, N, K, N, sptr()), "em_ln_gemm")
        assert_close(out, r, tol, f"ln_gemm {prec} epi {epi}")


@pytest.mark.parametrize("n,d,ff,hs", [(640, 512, 2048, 0), (160, 512, 2048, 0), (160, 512, 2048, 256), (320, 512, 2048, 512),
                                       (144, 256, 2048, 0), (640, 256, 2048, 512), (96, 256, 1024, 128), (32, 512, 2048, 128),
                                       (16, 512, 2048, 0), (336, 512, 2048, 0), (336, 256, 1024, 256)])
def test_dec_ffn_fragment_major(lib, n, d, ff, hs):
    """em_dec_ffn (norm3 + feed_forward + residual of a decoder label step on fragment-major operands: LayerNorm in the
    prologue of the first projection, which writes the hidden activation fragment-major; mid_gemm on 1 KiB operand loads):
    against torch fp32 on bf16-rounded operands with the hidden activation rounded to bf16, against the three-launch form on
    the row-major matrices (em_layernorm + em_gemm x 2: the same LayerNorm bits, another summation order), bit-repeatable;
    hs forces the hidden units per workgroup of the first launch (developer switch).  From 320 rows a workgroup owns TWO row
    fragments (n = 336: 21 fragments - the last workgroup's second one is past the end and must not be written)."""
    import os
    x = rnd(n, d, seed=71) * 2 + 0.3
    g, be = 1 + 0.1 * rnd(d, seed=72), 0.1 * rnd(d, seed=73)
    w1 = q(rnd(ff, d, seed=74, scale=d ** -0.5), torch.bfloat16)
    w2 = q(rnd(d, ff, seed=75, scale=ff ** -0.5), torch.bfloat16)
    b1, b2 = 0.1 * rnd(ff, seed=76), 0.1 * rnd(d, seed=77)
    xn = q(F.layer_norm(x, (d,), g, be, 1e-12), torch.bfloat16)
    hid = q(torch.relu(F.linear(xn, w1, b1)), torch.bfloat16)
    ref = x + F.linear(hid, w2, b2)
    gd, bd, w1d, w2d, b1d, b2d = dev(g), dev(be), dev(w1.to(torch.bfloat16)), dev(w2.to(torch.bfloat16)), dev(b1), dev(b2)
    w1f, w2f = dev(L.pack_frag16(w1.to(torch.bfloat16))), dev(L.pack_frag16(w2.to(torch.bfloat16)))
    if hs:
        os.environ["ESPNET_AMD_DEC_FFN_SPLIT"] = str(hs)
        lib.em_dev_switches_reload()
    try:
        assert lib.em_dec_ffn_split(n, d, ff) == (hs or {640: 256, 160: 128, 144: 128, 16: 128, 336: 128}[n])  # (from 320 rows 32-row workgroups: 640 rows = 20 row blocks)
        outs = []
        for _ in range(3):
            xd = dev(x.clone())
            hb = torch.full((n, ff), float("nan"), dtype=torch.bfloat16, device="cuda")
            L.check(lib.em_dec_ffn(L.EM_BF16, L.ptr(xd), L.ptr(gd), L.ptr(bd), 1e-12, L.ptr(w1f), L.ptr(b1d), L.ptr(w2f),
                                   L.ptr(b2d), n, d, ff, L.ptr(hb), sptr()), "em_dec_ffn")
            torch.cuda.synchronize()
            outs.append(xd.clone())
            # the hidden activation, read back out of its fragment-major layout: exactly relu(W1 LN(x) + b1) in bf16 up to
            # the summation order
            hrow = hb.view(n // 16, ff // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(n, ff)
            assert_close(hrow, hid, 2e-2, "dec_ffn hidden")
    finally:
        if hs:
            del os.environ["ESPNET_AMD_DEC_FFN_SPLIT"]
            lib.em_dev_switches_reload()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert_close(outs[0], ref, 2e-3, "dec_ffn vs torch")
    # the three-launch form on the same inputs
    x3 = dev(x.clone())
    hb3 = torch.zeros(n, ff, dtype=torch.bfloat16, device="cuda")
    xs = torch.zeros(n, d, dtype=torch.bfloat16, device="cuda")
    L.check(lib.em_layernorm(L.EM_BF16, L.ptr(x3), L.ptr(gd), L.ptr(bd), n, d, 1e-12, L.ptr(xs), None, sptr()))
    a1 = L.EmGemmArgs(A=xs.data_ptr(), W=w1d.data_ptr(), C=hb3.data_ptr(), bias=b1d.data_ptr(), M=n, N=ff, K=d, lda=d, ldc=ff, scale=1.0)
    L.check(lib.em_gemm(L.EM_BF16, L.EM_EPI_RELU, L.EM_A_PLAIN, a1, sptr()))
    a2 = L.EmGemmArgs(A=hb3.data_ptr(), W=w2d.data_ptr(), C=x3.data_ptr(), bias=b2d.data_ptr(), M=n, N=d, K=ff, lda=ff, ldc=d, scale=1.0)
    L.check(lib.em_gemm(L.EM_BF16, L.EM_EPI_RESID_F32, L.EM_A_PLAIN, a2, sptr()))
    assert_close(outs[0], x3, 2e-3, "dec_ffn vs three launches")
    # shapes outside the kernels are refused, never computed on another path
    assert lib.em_dec_ffn(L.EM_F32, L.ptr(x3), L.ptr(gd), L.ptr(bd), 1e-12, L.ptr(w1f), L.ptr(b1d), L.ptr(w2f), L.ptr(b2d),
                          n, d, ff, L.ptr(hb3), sptr()) == L.EM_ERR_UNSUPPORTED
    assert lib.em_dec_ffn_split(n + 3, d, ff) == 0 and lib.em_dec_ffn_split(n, 384, ff) == 0


@pytest.mark.parametrize("n,N,d", [(640, 1536, 512), (160, 1536, 512), (160, 5000, 512), (640, 5000, 512), (150, 768, 256),
                                   (97, 5000, 256), (3, 260, 512), (333, 1536, 512), (321, 5000, 256)])
def test_ln_gemm_frag_rows(lib, n, N, d):
    """em_ln_gemm_frag with row-major outputs (the label step's norm1 + q|k|v projection and after_norm + output_layer on a
    fragment-major weight whose rows are zero-padded to 512): against torch fp32 on the bf16-rounded LayerNorm output and
    against em_ln_gemm on the row-major matrix (the same LayerNorm bits); ragged n and N, nothing written past either."""
    x = rnd(n, d, seed=81) * 2 + 0.3
    g, be = 1 + 0.1 * rnd(d, seed=82), 0.1 * rnd(d, seed=83)
    w = q(rnd(N, d, seed=84, scale=d ** -0.5), torch.bfloat16)
    bias = 0.1 * rnd(N, seed=85)
    xn = q(F.layer_norm(x, (d,), g, be, 1e-12), torch.bfloat16)
    ref = F.linear(xn, w, bias)
    xd, gd, bd, wd, biasd = dev(x), dev(g), dev(be), dev(w.to(torch.bfloat16)), dev(bias)
    wf = dev(L.pack_frag16(w.to(torch.bfloat16), pad_rows=512))
    for mode, odt, epi, tol in ((L.EM_LNF_STORE, torch.bfloat16, L.EM_EPI_STORE, 2e-2), (L.EM_LNF_STORE_F32, torch.float32, L.EM_EPI_STORE_F32, 2e-3)):
        out = torch.full((n + 1, N), 5.0, dtype=odt, device="cuda")
        L.check(lib.em_ln_gemm_frag(mode, L.ptr(xd), L.ptr(gd), L.ptr(bd), 1e-12, L.ptr(wf), L.ptr(biasd), L.ptr(out), n, N, d,
                                    sptr()), "em_ln_gemm_frag")
        torch.cuda.synchronize()
        assert (out[n] == 5.0).all()
        assert_close(out[:n], ref, tol, f"ln_gemm_frag mode {mode}")
        old = torch.zeros(n, N, dtype=odt, device="cuda")
        L.check(lib.em_ln_gemm(L.EM_BF16, epi, L.ptr(xd), L.ptr(gd), L.ptr(bd), 1e-12, L.ptr(wd), L.ptr(biasd), L.ptr(old),
                               n, N, d, N, sptr()), "em_ln_gemm")
        assert_close(out[:n], old, tol, f"ln_gemm_frag vs ln_gemm mode {mode}")
    assert lib.em_ln_gemm_frag(L.EM_LNF_STORE, L.ptr(xd), L.ptr(gd), L.ptr(bd), 1e-12, L.ptr(wf), L.ptr(biasd), L.ptr(out), n, N + 2,
                               d, sptr()) == L.EM_ERR_UNSUPPORTED


@pytest.mark.parametrize("epi", ["STORE", "SWISH", "RELU", "GELU", "RESID_F32", "SCALE_F32", "STORE_F32"])
def test_gemm_large_m_bf16_128x128_tile(lib, epi):
    """Shapes big enough for the 128x128 tile (>= 384 workgroups; the other GEMM tests run the 64-row tile):
    every LDS-transposed epilogue against torch fp32 on the rounded operands, incl. ragged M / N edges."""
    M, N, K = 4001, 1160, 512
    a = q(rnd(M, K, seed=41), torch.bfloat16)
    w = q(rnd(N, K, seed=42, scale=K ** -0.5), torch.bfloat16)
    bias = 0.1 * rnd(N, seed=43)
    acc = a @ w.t() + bias
    c0 = rnd(M, N, seed=44)
    code = getattr(L, "EM_EPI_" + epi)
    f32out = epi in ("RESID_F32", "SCALE_F32", "STORE_F32")
    ref = {"STORE": acc, "SWISH": oc.swish(acc), "RELU": torch.relu(acc), "GELU": F.gelu(acc),
           "RESID_F32": c0 + 0.5 * acc, "SCALE_F32": 0.5 * acc, "STORE_F32": acc}[epi]
    out = dev(c0.clone()) if f32out else torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ad, wd, bd = dev(a.to(torch.bfloat16)), dev(w.to(torch.bfloat16)), dev(bias)
    args = L.EmGemmArgs(A=ad.data_ptr(), W=wd.data_ptr(), C=out.data_ptr(), bias=bd.data_ptr(), M=M, N=N, K=K, lda=K,
                        ldc=N, scale=0.5 if epi in ("RESID_F32", "SCALE_F32") else 1.0)
    L.check(lib.em_gemm(L.EM_BF16, code, L.EM_A_PLAIN, args, sptr()), "em_gemm large")
    assert_close(out, ref, 2e-4 if f32out else 2e-2, f"large-M gemm {epi}")



@pytest.mark.parametrize("B,T,h", [(3, 249, 8), (2, 31, 4), (5, 64, 2)])
def test_gemm_head_layout_epilogues(lib, B, T, h):
    """Round 4: the per-head operands of the LDS-resident attention written by the projection GEMMs themselves -
    EM_EPI_QK_HEADS (q | k as [B][heads][Tpad][64]) and EM_EPI_VT_HEADS (V^T [B][heads][64][Tpad] from the swapped product
    W_v . x^T with a per-row bias); frames >= T of a slab are left untouched.  T odd, utterance boundaries inside a tile."""
    d, K, Tpad = 64 * h, 64 * h, 256
    M = B * T
    x = q(rnd(M, K, seed=1), torch.bfloat16)
    w = q(rnd(3 * d, K, seed=2, scale=K ** -0.5), torch.bfloat16)
    bias = rnd(3 * d, seed=3)
    xd, wd, bd = dev(x.to(torch.bfloat16)), dev(w.to(torch.bfloat16)), dev(bias)
    qk = torch.full((2, B, h, Tpad, 64), 7.0, dtype=torch.bfloat16, device="cuda")
    vt = torch.full((B, h, 64, Tpad), 7.0, dtype=torch.bfloat16, device="cuda")
    a = L.EmGemmArgs(A=xd.data_ptr(), W=wd.data_ptr(), C=qk.data_ptr(), bias=bd.data_ptr(), M=M, N=2 * d, K=K, lda=K,
                     ldc=64, scale=1.0, T1=T, T2=Tpad, F1=h, d=d)
    L.check(lib.em_gemm(L.EM_BF16, L.EM_EPI_QK_HEADS, L.EM_A_PLAIN, a, sptr()), "em_gemm(QK_HEADS)")
    a = L.EmGemmArgs(A=wd.data_ptr() + 2 * d * K * 2, W=xd.data_ptr(), C=vt.data_ptr(), bias=bd.data_ptr() + 2 * d * 4,
                     M=d, N=M, K=K, lda=K, ldc=Tpad, scale=1.0, T1=T, T2=Tpad, F1=h, d=d)
    L.check(lib.em_gemm(L.EM_BF16, L.EM_EPI_VT_HEADS, L.EM_A_PLAIN, a, sptr()), "em_gemm(VT_HEADS)")
    ref = x @ w.t() + bias  # (M, 3d)
    for which in range(2):
        want = ref[:, which * d : (which + 1) * d].reshape(B, T, h, 64).permute(0, 2, 1, 3)
        assert_close(qk[which, :, :, :T], want, 2e-2, f"head layout which={which}")
        assert bool((qk[which, :, :, T:].float() == 7.0).all())
    want_v = ref[:, 2 * d :].reshape(B, T, h, 64).permute(0, 2, 3, 1)
    assert_close(vt[:, :, :, :T], want_v, 2e-2, "V^T head layout")
    assert bool((vt[:, :, :, T:].float() == 7.0).all())


@pytest.mark.parametrize("M,ff,ln_mode,pre", [(200, 256, 1, False), (1000, 2048, 1, False), (1000, 2048, 2, False),
                                              (64, 1024, 2, False), (1000, 2048, 2, True), (333, 512, 1, True),
                                              (17000, 256, 1, True)])  # (266 workgroups: a second round on 256 CUs)
def test_ffn_rows_fused(lib, M, ff, ln_mode, pre):
    """Round 4: the 512-wide model's feed-forward module as ONE row-block launch (csrc/ffn_rows.hip: w_1 + Swish + w_2 +
    residual + the LayerNorm(s) that follow, positionwise_feed_forward.py:30-32 inside encoder_layer.py:111-121 / :160-171)
    against torch f32 on the same bf16-rounded operands (the hidden activation rounded to bf16 as the kernel rounds it).
    Ragged M (the last workgroup overhangs), the minimal two chunks, both LayerNorm modes, rows past M untouched.
    pre: the launch starts with a projection of its own, x += W_pre . pre_in + b_pre, and LayerNorm(pre_g, pre_be) of the
    result is the module's input (pointwise_conv2 + residual + norm_ff, convolution.py:78-79, encoder_layer.py:158-161)."""
    from espnet_amd.asr.encoder.conformer_encoder import pack_ffn_rows_w1, pack_ffn_rows_w2, pack_rows_proj

    d = 512
    x = rnd(M, d, seed=51) * 2 + 0.3
    g0, b0 = 1 + 0.1 * rnd(d, seed=52), 0.1 * rnd(d, seed=53)
    w1 = q(rnd(ff, d, seed=54, scale=d ** -0.5), torch.bfloat16)
    w2 = q(rnd(d, ff, seed=55, scale=ff ** -0.5), torch.bfloat16)
    b1, b2 = 0.1 * rnd(ff, seed=56), 0.1 * rnd(d, seed=57)
    g1, be1, g2, be2 = 1 + 0.1 * rnd(d, seed=58), 0.1 * rnd(d, seed=59), 1 + 0.1 * rnd(d, seed=60), 0.1 * rnd(d, seed=61)
    if pre:
        pin = q(rnd(M, d, seed=62), torch.bfloat16)
        wp = q(rnd(d, d, seed=63, scale=d ** -0.5), torch.bfloat16)
        bp = 0.1 * rnd(d, seed=64)
        x0 = x + (pin @ wp.t() + bp)  # the residual stream the module sees
    else:
        x0 = x
    xn = q(F.layer_norm(x0, (d,), g0, b0, 1e-12), torch.bfloat16)
    h = q(oc.swish(xn @ w1.t() + b1), torch.bfloat16)
    x1 = x0 + 0.5 * (h @ w2.t() + b2)
    if ln_mode == 1:
        ref_x, ref_n = x1, F.layer_norm(x1, (d,), g1, be1, 1e-12)
    else:
        ref_x = F.layer_norm(x1, (d,), g1, be1, 1e-12)
        ref_n = F.layer_norm(ref_x, (d,), g2, be2, 1e-12)
    pad = 70  # rows past M: must stay untouched
    xd = dev(torch.cat([x, torch.full((pad, d), 7.0)]))
    xnd = dev(torch.cat([xn, torch.full((pad, d), 7.0)]).to(torch.bfloat16))
    out_n = torch.full((M + pad, d), 7.0, dtype=torch.bfloat16, device="cuda")
    out_f = torch.full((M + pad, d), 7.0, device="cuda")
    a = L.EmFfnRowsArgs(xn_in=0 if pre else xnd.data_ptr(), x=xd.data_ptr(),
                        w1p=dev(pack_ffn_rows_w1(w1).to(torch.bfloat16)).data_ptr(),
                        w2p=dev(pack_ffn_rows_w2(w2).to(torch.bfloat16)).data_ptr(),
                        b1=dev(b1).data_ptr(), b2=dev(b2).data_ptr(), g1=dev(g1).data_ptr(), be1=dev(be1).data_ptr(),
                        g2=dev(g2).data_ptr(), be2=dev(be2).data_ptr(), xn_out=out_n.data_ptr(),
                        out_f32=out_f.data_ptr() if ln_mode == 2 else 0, M=M, d=d, ff=ff, ln_mode=ln_mode, scale=0.5,
                        eps=1e-12)
    if pre:
        pind = dev(torch.cat([pin, torch.full((pad, d), 7.0)]).to(torch.bfloat16))
        a.pre_in, a.pre_w = pind.data_ptr(), dev(pack_rows_proj(wp).to(torch.bfloat16)).data_ptr()
        a.pre_b, a.pre_g, a.pre_be = dev(bp).data_ptr(), dev(g0).data_ptr(), dev(b0).data_ptr()
    L.check(lib.em_ffn_rows_fused(a, sptr()), "em_ffn_rows_fused")
    torch.cuda.synchronize()
    assert_close(xd[:M], ref_x, 2e-3, f"ffn_rows x (mode {ln_mode})")  # f32 out; the bf16 rounding of H is in the reference
    assert_close(out_n[:M], ref_n, 1e-2, f"ffn_rows LN out (mode {ln_mode})")
    if ln_mode == 2:
        assert_close(out_f[:M], ref_n, 2e-3, "ffn_rows f32 copy")
    assert bool((xd[M:] == 7.0).all()) and bool((out_n[M:].float() == 7.0).all()) and bool((out_f[M:] == 7.0).all())
    # in place over its own input (the encoder's use): same result
    a.xn_out = pind.data_ptr() if pre else xnd.data_ptr()
    xd2 = dev(torch.cat([x, torch.full((pad, d), 7.0)]))
    a.x = xd2.data_ptr()
    L.check(lib.em_ffn_rows_fused(a, sptr()), "em_ffn_rows_fused (aliased)")
    torch.cuda.synchronize()
    assert torch.equal((pind if pre else xnd)[:M], out_n[:M]) and torch.equal(xd2[:M], xd[:M])
    with pytest.raises(NotImplementedError):
        a.d = 256
        L.check(lib.em_ffn_rows_fused(a, sptr()), "em_ffn_rows_fused d=256")


@pytest.mark.parametrize("M,V,pre", [(1000, 5000, True), (130, 300, False)])
def test_ffn_rows_ctc_argmax_walk(lib, M, V, pre):
    """Round 4: the CTC head's arg-max (asr/ctc.py:207-215) as a walk behind the last row-block launch of the 512-wide stack
    (EmFfnRowsArgs.post_*): the ids must be arg-maxima of ctc_lo applied to the launch's OWN bf16 output (what the stand-alone
    arg-max GEMM reads) - equal to torch.argmax except on near-ties (the summation order differs), where the chosen label's
    logit is within 1e-3 of the maximum; rows past M untouched; V not a multiple of 128 (padded rows never win)."""
    from espnet_amd.asr.encoder.conformer_encoder import pack_ffn_rows_w1, pack_ffn_rows_w2, pack_rows_proj

    d, ff = 512, 1024
    x = rnd(M, d, seed=81) * 2 + 0.3
    g0, b0 = 1 + 0.1 * rnd(d, seed=82), 0.1 * rnd(d, seed=83)
    w1 = q(rnd(ff, d, seed=84, scale=d ** -0.5), torch.bfloat16)
    w2 = q(rnd(d, ff, seed=85, scale=ff ** -0.5), torch.bfloat16)
    b1, b2 = 0.1 * rnd(ff, seed=86), 0.1 * rnd(d, seed=87)
    g1, be1, g2, be2 = 1 + 0.1 * rnd(d, seed=88), 0.1 * rnd(d, seed=89), 1 + 0.1 * rnd(d, seed=90), 0.1 * rnd(d, seed=91)
    wc = q(rnd(V, d, seed=92, scale=d ** -0.5), torch.bfloat16)
    bc = 0.1 * rnd(V, seed=93)
    chunks = (V + 127) // 128
    wcp = torch.zeros(chunks * 128, d)
    wcp[:V] = wc
    bcp = torch.full((chunks * 128,), -3.0e38)
    bcp[:V] = bc
    pad = 70
    xd = dev(torch.cat([x, torch.full((pad, d), 7.0)]))
    out_n = torch.full((M + pad, d), 7.0, dtype=torch.bfloat16, device="cuda")
    out_f = torch.full((M + pad, d), 7.0, device="cuda")
    ids = torch.full((M + pad,), -7, dtype=torch.int32, device="cuda")
    a = L.EmFfnRowsArgs(x=xd.data_ptr(), w1p=dev(pack_ffn_rows_w1(w1).to(torch.bfloat16)).data_ptr(),
                        w2p=dev(pack_ffn_rows_w2(w2).to(torch.bfloat16)).data_ptr(), b1=dev(b1).data_ptr(), b2=dev(b2).data_ptr(),
                        g1=dev(g1).data_ptr(), be1=dev(be1).data_ptr(), g2=dev(g2).data_ptr(), be2=dev(be2).data_ptr(),
                        xn_out=out_n.data_ptr(), out_f32=out_f.data_ptr(), M=M, d=d, ff=ff, ln_mode=2, scale=0.5, eps=1e-12,
                        post_w=dev(pack_ffn_rows_w1(wcp).to(torch.bfloat16)).data_ptr(), post_b=dev(bcp).data_ptr(),
                        post_ids=ids.data_ptr(), post_chunks=chunks, post_vocab=V)
    if pre:
        pin = q(rnd(M, d, seed=94), torch.bfloat16)
        wp = q(rnd(d, d, seed=95, scale=d ** -0.5), torch.bfloat16)
        pind = dev(torch.cat([pin, torch.full((pad, d), 7.0)]).to(torch.bfloat16))
        a.pre_in, a.pre_w = pind.data_ptr(), dev(pack_rows_proj(wp).to(torch.bfloat16)).data_ptr()
        a.pre_b, a.pre_g, a.pre_be = dev(0.1 * rnd(d, seed=96)).data_ptr(), dev(g0).data_ptr(), dev(b0).data_ptr()
    else:
        xnd = dev(q(F.layer_norm(x, (d,), g0, b0, 1e-12), torch.bfloat16).to(torch.bfloat16))
        a.xn_in = xnd.data_ptr()
    L.check(lib.em_ffn_rows_fused(a, sptr()), "em_ffn_rows_fused(post)")
    torch.cuda.synchronize()
    logits = out_n[:M].float().cpu() @ wc.t() + bc  # from the launch's own bf16 output
    got = ids[:M].cpu().long()
    assert int(got.min()) >= 0 and int(got.max()) < V
    ref = logits.argmax(-1)
    chosen = logits.gather(1, got[:, None])[:, 0]
    assert bool((chosen >= logits.max(-1).values - 1e-3).all()), (chosen - logits.max(-1).values).min().item()
    print(f"[ctc walk] {int((got != ref).sum())} of {M} rows differ from torch.argmax (near-ties)")
    assert int((got != ref).sum()) <= max(1, M // 200)
    assert bool((ids[M:] == -7).all())


@pytest.mark.parametrize("M", [1000, 77])
def test_rows_glu_fused(lib, M):
    """Round 4: attention output -> linear_out + residual -> norm_conv -> pointwise_conv1 + GLU as ONE row-block launch
    (csrc/ffn_rows.hip, EM_ROWS_GLU; attention.py:149-151, encoder_layer.py:147-151, convolution.py:62-66) against torch f32 on
    the same bf16-rounded operands (LN(x) rounded to bf16 as the kernel rounds it).  Ragged M, rows past M untouched."""
    from espnet_amd.asr.encoder.conformer_encoder import glu_chunk_order, pack_rows_glu, pack_rows_proj

    d = 512
    x = rnd(M, d, seed=71) * 2 + 0.3
    ctx = q(rnd(M, d, seed=72), torch.bfloat16)
    wout = q(rnd(d, d, seed=73, scale=d ** -0.5), torch.bfloat16)
    bout = 0.1 * rnd(d, seed=74)
    g, be = 1 + 0.1 * rnd(d, seed=75), 0.1 * rnd(d, seed=76)
    pw1 = q(rnd(2 * d, d, seed=77, scale=d ** -0.5), torch.bfloat16)
    pb = 0.1 * rnd(2 * d, seed=78)
    x1 = x + (ctx @ wout.t() + bout)
    xn = q(F.layer_norm(x1, (d,), g, be, 1e-12), torch.bfloat16)
    pw = xn @ pw1.t() + pb
    ref = pw[:, :d] * torch.sigmoid(pw[:, d:])
    pad = 70
    xd = dev(torch.cat([x, torch.full((pad, d), 7.0)]))
    cd = dev(torch.cat([ctx, torch.full((pad, d), 7.0)]).to(torch.bfloat16))
    out = torch.full((M + pad, d), 7.0, dtype=torch.bfloat16, device="cuda")
    a = L.EmFfnRowsArgs(x=xd.data_ptr(), w1p=dev(pack_rows_glu(pw1).to(torch.bfloat16)).data_ptr(),
                        b1=dev(pb[glu_chunk_order(d)]).data_ptr(), xn_out=out.data_ptr(), M=M, d=d, ff=2 * d, ln_mode=1,
                        scale=1.0, eps=1e-12, pre_in=cd.data_ptr(), pre_w=dev(pack_rows_proj(wout).to(torch.bfloat16)).data_ptr(),
                        pre_b=dev(bout).data_ptr(), pre_g=dev(g).data_ptr(), pre_be=dev(be).data_ptr(), main=L.EM_ROWS_GLU)
    L.check(lib.em_ffn_rows_fused(a, sptr()), "em_ffn_rows_fused(GLU)")
    torch.cuda.synchronize()
    assert_close(xd[:M], x1, 2e-3, "rows_glu x")
    assert_close(out[:M], ref, 1e-2, "rows_glu out")
    assert bool((xd[M:] == 7.0).all()) and bool((out[M:].float() == 7.0).all())
