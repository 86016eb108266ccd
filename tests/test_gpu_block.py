"""GPU parity tests of the fused per-block path (csrc/block.hip, csrc/attention2.hip), through the C ABI.

Every fused kernel is checked against a torch fp32 restatement of the reference operators it replaces
(conformer/encoder_layer.py:79-179, conformer/convolution.py:56-79, attention.py:391-459), computed FROM THE
SAME bf16-ROUNDED operands and with the same bf16 rounding points the unfused bf16 path has (LayerNorm output,
FFN hidden activation, GLU output, depthwise-conv output, q / k / v), so what is checked is the kernel, not the
rounding.  Tolerances: 4e-3 of the output scale for the f32 residual stream (same class as
test_ffn_fused_bf16), 2e-2 for bf16 outputs (one bf16 ulp is 0.8 %).  Then the whole encoder: fused vs the
per-operator launch sequence on the reference goldens.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from espnet_amd import lib as L
from espnet_amd.asr.encoder.conformer_encoder import pack_k_units, pack_w1, pack_w2
from oracle import conformer as oc

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
D, H, G = 256, 4, L.EM_BLOCK_PARAM_GROUP


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return L.load()


_KEEP = []


def dev(t):
    t = t.contiguous().cuda()
    _KEEP.append(t)
    return t


@pytest.fixture(autouse=True)
def _release_kept_tensors():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(t):
    return t.to(BF).to(torch.float32)


def assert_close(got, ref, tol, what=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


def ln(x, g, b):
    return F.layer_norm(x, (D,), g, b, 1e-12)


def group(*vecs):
    v = torch.cat([t.reshape(-1).float() for t in vecs])
    return F.pad(v, (0, G - v.numel()))


def pad_ff(b):
    return F.pad(b, (0, 1024 - b.numel()))


def tpad(T):
    return (T + 255) // 256 * 256


# ------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("T,klens", [(49, [49, 20]), (128, [128]), (249, [249, 130, 1]), (256, [256, 255]),
                                     (300, [300, 257]), (700, [700, 513])])
def test_relpos_attention2(lib, T, klens):
    """em_relpos_attention2_bf16 against the literal rel_shift formulation (the same reference as
    test_relpos_attention), on per-head operand layouts; T > 256 exercises the super-tile loop."""
    B, dk = len(klens), 64
    d = H * dk
    Tp = tpad(T)
    qq = q(rnd(B, T, H, dk, seed=26))
    kk = q(rnd(B, T, H, dk, seed=27))
    vv = q(rnd(B, T, H, dk, seed=28))
    Lb = 3  # the table holds linear_pos of several blocks side by side (ldp = Lb * d); use the middle one
    pall = q(rnd(2 * T - 1, Lb * d, seed=29))
    p = pall[:, d:2 * d]
    u, v = rnd(H, dk, seed=30, scale=0.3), rnd(H, dk, seed=31, scale=0.3)
    q_u = q(qq + u).transpose(1, 2)
    q_v = q(qq + v).transpose(1, 2)
    pp = p.reshape(1, 2 * T - 1, H, dk).transpose(1, 2)
    ac = q_u @ kk.transpose(1, 2).transpose(-2, -1)
    bd = oc.rel_shift(q_v @ pp.transpose(-2, -1))
    scores = (ac + bd) / math.sqrt(dk)
    mask = oc.make_pad_mask(torch.tensor(klens), T)[:, None, None, :]
    attn = torch.softmax(scores.masked_fill(mask, torch.finfo(torch.float32).min), -1).masked_fill(mask, 0.0)
    ref = (attn @ vv.transpose(1, 2)).transpose(1, 2).reshape(B, T, d)

    qh = torch.full((B, H, Tp, dk), float("nan"), dtype=BF)  # rows >= T: never read into a stored output
    kh = torch.full((B, H, Tp, dk), float("nan"), dtype=BF)  # keys >= klen: masked by select, never by arithmetic
    vt = torch.zeros(B, H, dk, Tp, dtype=BF)                 # V^T padding must be finite (0 * x)
    qh[:, :, :T] = qq.transpose(1, 2).to(BF)
    kh[:, :, :T] = kk.transpose(1, 2).to(BF)
    vt[:, :, :, :T] = vv.permute(0, 2, 3, 1).to(BF)
    ctx = torch.zeros(B, T, d, dtype=BF, device="cuda")
    palld = dev(pall.to(BF))
    L.check(lib.em_relpos_attention2_bf16(L.ptr(dev(qh)), L.ptr(dev(kh)), L.ptr(dev(vt)),
                                          palld.data_ptr() + d * 2, Lb * d, L.ptr(dev(u)), L.ptr(dev(v)),
                                          L.ptr(dev(torch.tensor(klens, dtype=torch.int32))), B, T, Tp, H,
                                          L.ptr(ctx), L.current_stream_ptr()), "attention2")
    assert_close(ctx, ref, 2e-2, f"attention2 T={T}")


# ------------------------------------------------------------------------------------------ block kernels
class Layer:
    """Random weights of one EncoderLayer (bf16-rounded matrices, f32 vectors) + its host packing."""

    def __init__(self, seed, ff):
        s = iter(range(seed, seed + 100))
        self.ff = ff
        w = lambda n, k: q(rnd(n, k, seed=next(s), scale=k ** -0.5))
        vb = lambda n: rnd(n, seed=next(s), scale=0.1)
        lng = lambda: (1 + 0.1 * rnd(D, seed=next(s)), 0.1 * rnd(D, seed=next(s)))
        self.ln_mac, self.ln_mha, self.ln_conv, self.ln_ff, self.ln_final = lng(), lng(), lng(), lng(), lng()
        self.ffm_w1, self.ffm_b1, self.ffm_w2, self.ffm_b2 = w(ff, D), vb(ff), w(D, ff), vb(D)
        self.ff_w1, self.ff_b1, self.ff_w2, self.ff_b2 = w(ff, D), vb(ff), w(D, ff), vb(D)
        self.wqkv, self.bqkv = w(3 * D, D), vb(3 * D)
        self.wout, self.bout = w(D, D), vb(D)
        self.pw1, self.pw1_b = w(2 * D, D), vb(2 * D)
        self.dw_w, self.dw_b = rnd(31, D, seed=next(s), scale=0.25), vb(D)  # [k][d], BatchNorm already folded
        self.pw2, self.pw2_b = w(D, D), vb(D)

    # torch fp32 restatements with the bf16 rounding points of the bf16 path
    def part_a(self, x):
        xn = q(ln(x, *self.ln_mac))
        x = x + 0.5 * (q(oc.swish(xn @ self.ffm_w1.t() + self.ffm_b1)) @ self.ffm_w2.t() + self.ffm_b2)
        qkv = q(ln(x, *self.ln_mha)) @ self.wqkv.t() + self.bqkv
        return x, qkv

    def part_c(self, x, ctx):
        x = x + ctx @ self.wout.t() + self.bout
        y = q(ln(x, *self.ln_conv)) @ self.pw1.t() + self.pw1_b
        return x, y[:, :D] * torch.sigmoid(y[:, D:])

    def part_d(self, x, glu, B, T, tlens=None):
        g3 = q(glu).reshape(B, T, D).clone()
        if tlens is not None:
            for b, n in enumerate(tlens):
                g3[b, n:] = 0
        conv = F.conv1d(g3.transpose(1, 2), self.dw_w.t().reshape(D, 1, 31), self.dw_b, padding=15, groups=D)
        c = q(oc.swish(conv.transpose(1, 2).reshape(B * T, D)))
        x = x + c @ self.pw2.t() + self.pw2_b
        xn = q(ln(x, *self.ln_ff))
        x = x + 0.5 * (q(oc.swish(xn @ self.ff_w1.t() + self.ff_b1)) @ self.ff_w2.t() + self.ff_b2)
        return ln(x, *self.ln_final)

    def a_groups(self):
        return [group(*self.ln_mac, pad_ff(self.ffm_b1), self.ffm_b2), group(*self.ln_mha, self.bqkv)]

    def d_groups(self):
        return [group(self.pw2_b, *self.ln_ff), group(pad_ff(self.ff_b1), self.ff_b2, *self.ln_final)]

    def perm(self):
        return torch.cat([torch.cat([torch.arange(64 * j, 64 * j + 64), torch.arange(D + 64 * j, D + 64 * j + 64)])
                          for j in range(4)])

    def c_group(self):
        return group(self.bout, *self.ln_conv, self.pw1_b[self.perm()])


def block_args(B, T, ff, **ptrs):
    a = L.EmBlockArgs(B=B, T=T, Tpad=tpad(T), d=D, ff=ff, kernel=31, eps=1e-12)
    for k, v in ptrs.items():
        setattr(a, k, v.data_ptr() if v is not None else None)
    return a


def split_heads(buf, B, T):
    """[B][H][Tpad][64] -> [B*T][256]"""
    return buf[:, :, :T].permute(0, 2, 1, 3).reshape(B * T, D)


def _k_frag_index(Tp):
    """key index of K'[jt][n][ks][lane][e] (EmBlockArgs.kv_frag): [Tp/64, 4, 1, 64, 1] broadcastable."""
    jt = torch.arange(Tp // 64).reshape(-1, 1, 1, 1, 1)
    n = torch.arange(4).reshape(1, -1, 1, 1, 1)
    lane = torch.arange(64).reshape(1, 1, 1, -1, 1)
    lr = lane % 16
    return 64 * jt + 32 * (n // 2) + 8 * (lr // 4) + 4 * (n % 2) + (lr % 4)


def k_to_frag(kh):
    """[B][H][Tp][64] row-major -> the fragment-major K of EmBlockArgs.kv_frag (same shape, other order)."""
    B, Hh, Tp, _ = kh.shape
    key = _k_frag_index(Tp).expand(Tp // 64, 4, 2, 64, 8)
    ks = torch.arange(2).reshape(1, 1, -1, 1, 1)
    lg = (torch.arange(64) // 16).reshape(1, 1, 1, -1, 1)
    col = (32 * ks + 8 * lg + torch.arange(8).reshape(1, 1, 1, 1, -1)).expand(Tp // 64, 4, 2, 64, 8)
    return kh[:, :, key, col].reshape(B, Hh, Tp, 64)


def k_from_frag(kf):
    B, Hh, Tp, _ = kf.shape
    key = _k_frag_index(Tp).expand(Tp // 64, 4, 2, 64, 8)
    ks = torch.arange(2).reshape(1, 1, -1, 1, 1)
    lg = (torch.arange(64) // 16).reshape(1, 1, 1, -1, 1)
    col = (32 * ks + 8 * lg + torch.arange(8).reshape(1, 1, 1, 1, -1)).expand(Tp // 64, 4, 2, 64, 8)
    out = torch.zeros_like(kf)
    out[:, :, key, col] = kf.reshape(B, Hh, Tp // 64, 4, 2, 64, 8)
    return out


def _vt_frag_index(Tp):
    jt = torch.arange(Tp // 64).reshape(-1, 1, 1, 1, 1)
    f = torch.arange(4).reshape(1, -1, 1, 1, 1)
    jp = torch.arange(2).reshape(1, 1, -1, 1, 1)
    lane = torch.arange(64).reshape(1, 1, 1, -1, 1)
    e = torch.arange(8).reshape(1, 1, 1, 1, -1)
    row = (16 * f + lane % 16).expand(Tp // 64, 4, 2, 64, 8)
    key = (64 * jt + 32 * jp + 8 * (lane // 16) + e).expand(Tp // 64, 4, 2, 64, 8)
    return row, key


def vt_to_frag(vt):
    """[B][H][64][Tp] (V transposed, row-major) -> the fragment-major V^T of EmBlockArgs.kv_frag."""
    B, Hh, _, Tp = vt.shape
    row, key = _vt_frag_index(Tp)
    return vt[:, :, row, key].reshape(B, Hh, 64, Tp)


def vt_from_frag(vf):
    B, Hh, _, Tp = vf.shape
    row, key = _vt_frag_index(Tp)
    out = torch.zeros_like(vf)
    out[:, :, row, key] = vf.reshape(B, Hh, Tp // 64, 4, 2, 64, 8)
    return out


SHAPES = [(2, 249, 1024), (3, 70, 256), (1, 32, 1024), (2, 31, 64), (1, 40, 192), (2, 100, 576)]  # ff / 64 even, 1, odd


@pytest.mark.parametrize("B,T,ff", SHAPES)
@pytest.mark.parametrize("kv_frag", [0, 1])
def test_block_a(lib, B, T, ff, kv_frag):
    """kv_frag = 1 (round 6): K and V^T written fragment-major for block<ATT|C>; unpacked here and held to the same reference."""
    ly = Layer(100, ff)
    x0 = rnd(B * T, D, seed=1)
    x_ref, qkv_ref = ly.part_a(x0)
    Tp = tpad(T)
    xd = dev(x0.clone())
    qh, kh = (torch.zeros(B, H, Tp, 64, dtype=BF, device="cuda") for _ in range(2))
    vt = torch.zeros(B, H, 64, Tp, dtype=BF, device="cuda")
    a = block_args(B, T, ff, x=xd, qh=qh, kh=kh, vt=vt, ffm_w1=dev(pack_w1(ly.ffm_w1).to(BF)), ffm_w2=dev(pack_w2(ly.ffm_w2).to(BF)),
                   wqkv=dev(pack_k_units(ly.wqkv).to(BF)), params=dev(torch.cat(ly.a_groups())))
    a.kv_frag = kv_frag
    L.check(lib.em_conformer_block_fused(L.EM_BLOCK_A, a, L.current_stream_ptr()), "block<A>")
    if kv_frag:
        kh, vt = k_from_frag(kh.cpu()), vt_from_frag(vt.cpu())
    assert_close(xd, x_ref, 4e-3, "block<A> x")
    assert_close(split_heads(qh, B, T), qkv_ref[:, :D], 2e-2, "block<A> q")
    assert_close(split_heads(kh, B, T), qkv_ref[:, D:2 * D], 2e-2, "block<A> k")
    v_got = vt[:, :, :, :T].permute(0, 3, 1, 2).reshape(B * T, D)
    assert_close(v_got, qkv_ref[:, 2 * D:], 2e-2, "block<A> v")
    assert torch.isfinite(vt.float()).all()


@pytest.mark.parametrize("B,T,ff", SHAPES)
def test_block_c(lib, B, T, ff):
    ly = Layer(200, ff)
    x0, ctx = rnd(B * T, D, seed=2), q(rnd(B * T, D, seed=3))
    x_ref, glu_ref = ly.part_c(x0, ctx)
    xd = dev(x0.clone())
    glu = torch.zeros(B * T, D, dtype=BF, device="cuda")
    a = block_args(B, T, ff, x=xd, ctx=dev(ctx.to(BF)), glu=glu, wout=dev(pack_k_units(ly.wout).to(BF)),
                   pw1f=dev(pack_k_units(ly.pw1[ly.perm()]).to(BF)), params=dev(ly.c_group()))
    L.check(lib.em_conformer_block_fused(L.EM_BLOCK_C, a, L.current_stream_ptr()), "block<C>")
    assert_close(xd, x_ref, 4e-3, "block<C> x")
    assert_close(glu, glu_ref, 2e-2, "block<C> glu")


def _attention_inputs(B, T, klens, seed):
    """Per-head operands of the attention as block<A> writes them, and the torch restatement of the context."""
    dk = 64
    qq = q(rnd(B, T, H, dk, seed=seed))
    kk = q(rnd(B, T, H, dk, seed=seed + 1))
    vv = q(rnd(B, T, H, dk, seed=seed + 2))
    Lb = 3  # linear_pos of several blocks side by side (ldp = Lb * d); the middle one is this block's
    pall = q(rnd(2 * T - 1, Lb * D, seed=seed + 3))
    p = pall[:, D:2 * D]
    u, v = rnd(H, dk, seed=seed + 4, scale=0.3), rnd(H, dk, seed=seed + 5, scale=0.3)
    q_u = q((qq + u) * 0.125).transpose(1, 2)  # (1 / sqrt(d_k) rides in the operands: a power of two, exact)
    q_v = q((qq + v) * 0.125).transpose(1, 2)
    pp = p.reshape(1, 2 * T - 1, H, dk).transpose(1, 2)
    scores = q_u @ kk.transpose(1, 2).transpose(-2, -1) + oc.rel_shift(q_v @ pp.transpose(-2, -1))
    mask = oc.make_pad_mask(torch.tensor(klens), T)[:, None, None, :]
    attn = torch.softmax(scores.masked_fill(mask, torch.finfo(torch.float32).min), -1).masked_fill(mask, 0.0)
    ctx = (attn @ vv.transpose(1, 2)).transpose(1, 2).reshape(B * T, D)
    Tp = tpad(T)
    qh = torch.full((B, H, Tp, dk), float("nan"), dtype=BF)  # rows >= T: never read into a stored output
    kh = torch.full((B, H, Tp, dk), float("nan"), dtype=BF)  # keys >= klen: masked by select, never by arithmetic
    vt = torch.zeros(B, H, dk, Tp, dtype=BF)                 # V^T padding must be finite (0 * x)
    qh[:, :, :T] = qq.transpose(1, 2).to(BF)
    kh[:, :, :T] = kk.transpose(1, 2).to(BF)
    vt[:, :, :, :T] = vv.permute(0, 2, 3, 1).to(BF)
    return dict(qh=qh, kh=kh, vt=vt, pall=pall, u=u, v=v, ctx=ctx, Lb=Lb)


def packed_pos(lib, palld, T, Lb):
    """em_relpos_pack_pos_bf16 of a [2T-1][Lb * 256] table, checked against its definition (include/espnet_amd.h)."""
    npg = lib.em_relpos_pos_fragments(T)
    assert npg == 2 * ((T + 31) // 32) + 4 * ((T + 63) // 64) + 2
    out = torch.full((Lb, 4, npg, 2, 64, 8), float("nan"), dtype=BF, device="cuda")
    L.check(lib.em_relpos_pack_pos_bf16(L.ptr(palld), Lb * D, T, Lb, L.ptr(out), L.current_stream_ptr()), "pack_pos")
    gx = (T + 31) // 32
    g = torch.arange(npg).reshape(-1, 1, 1, 1)
    lane = torch.arange(64).reshape(1, 1, -1, 1)
    row = (16 * g + T - 32 * gx - 32 + lane % 16).clamp(0, 2 * T - 2).expand(npg, 2, 64, 8)
    col = (32 * torch.arange(2).reshape(1, -1, 1, 1) + 8 * (lane // 16) + torch.arange(8).reshape(1, 1, 1, -1)).expand(npg, 2, 64, 8)
    p = palld.cpu()
    for l in range(Lb):
        for h in range(4):
            assert torch.equal(out[l, h].cpu(), p[row, l * D + h * 64 + col]), (l, h)
    _KEEP.append(out)
    return out, npg


ATT_SHAPES = [(2, 249, [249, 130]), (3, 70, [70, 1, 33]), (1, 32, [32]), (2, 31, [31, 7]), (1, 128, [128]), (2, 256, [256, 255]),
              (1, 300, [257]), (2, 700, [700, 513]), (9, 40, [40, 39, 38, 37, 3, 2, 1, 40, 17]), (16, 65, list(range(50, 66)))]


@pytest.mark.parametrize("B,T,klens", ATT_SHAPES)
def test_block_att_c(lib, B, T, klens):
    """Round 6, block<ATT|C>: relative-position attention of a workgroup's 32 queries + linear_out + residual + norm_conv +
    pointwise_conv1 + GLU in ONE launch, against the torch restatement of attention.py:416-459 followed by part_c, and
    against the two-launch form (em_relpos_attention2_bf16 + block<C>) on the same operands.  Shapes: key counts that end
    inside / on a 64-key tile, odd and even tile counts, more than 256 keys, utterances of one frame, batches that do and
    do not fill groups of 8 utterances (the workgroup -> (utterance, query block) map differs)."""
    ly = Layer(1100, 256)
    at = _attention_inputs(B, T, klens, 40)
    x0 = rnd(B * T, D, seed=2)
    x_ref, glu_ref = ly.part_c(x0, q(at["ctx"]))
    Tp = tpad(T)
    palld = dev(at["pall"].to(BF))
    qh, kh, vt = dev(at["qh"]), dev(at["kh"]), dev(at["vt"])
    khf, vtf = dev(k_to_frag(at["kh"])), dev(vt_to_frag(at["vt"]))  # the layouts the A part writes with kv_frag = 1
    ppk, npg = packed_pos(lib, palld, T, at["Lb"])
    u, v, kl = dev(at["u"]), dev(at["v"]), dev(torch.tensor(klens, dtype=torch.int32))
    wout, pw1f, par = dev(pack_k_units(ly.wout).to(BF)), dev(pack_k_units(ly.pw1[ly.perm()]).to(BF)), dev(ly.c_group())
    xd = dev(x0.clone())
    glu = torch.zeros(B * T, D, dtype=BF, device="cuda")
    a = block_args(B, T, 256, x=xd, glu=glu, qh=qh, kh=khf, vt=vtf, wout=wout, pw1f=pw1f, params=par, pos_u=u, pos_v=v, klens=kl)
    a.pos, a.ldp, a.kv_frag = ppk.data_ptr() + 4 * npg * 2048, npg, 1  # (the middle block of the table)
    L.check(lib.em_conformer_block_fused(L.EM_BLOCK_ATT | L.EM_BLOCK_C, a, L.current_stream_ptr()), "block<ATT|C>")
    # rows of an utterance past its key count attend over the valid keys like any other row (the reference masks keys only)
    assert_close(xd, x_ref, 4e-3, "block<ATT|C> x")
    assert_close(glu, glu_ref, 2e-2, "block<ATT|C> glu")
    # the two-launch form on the same operands: same rounding points, so the two agree far inside the tolerance above
    ctx2 = torch.zeros(B * T, D, dtype=BF, device="cuda")
    L.check(lib.em_relpos_attention2_bf16(L.ptr(qh), L.ptr(kh), L.ptr(vt), palld.data_ptr() + D * 2, at["Lb"] * D, L.ptr(u),
                                          L.ptr(v), L.ptr(kl), B, T, Tp, H, L.ptr(ctx2), L.current_stream_ptr()), "attention2")
    x2 = dev(x0.clone())
    glu2 = torch.zeros(B * T, D, dtype=BF, device="cuda")
    a2 = block_args(B, T, 256, x=x2, ctx=ctx2, glu=glu2, wout=wout, pw1f=pw1f, params=par)
    L.check(lib.em_conformer_block_fused(L.EM_BLOCK_C, a2, L.current_stream_ptr()), "block<C>")
    assert_close(xd, x2, 1e-3, "block<ATT|C> vs attention2 + block<C>: x")
    assert_close(glu, glu2, 1.6e-2, "block<ATT|C> vs attention2 + block<C>: glu")  # (a bf16 ulp where a rounding falls the other way)
    # missing operands are refused
    a.klens = None
    assert lib.em_conformer_block_fused(L.EM_BLOCK_ATT | L.EM_BLOCK_C, a, L.current_stream_ptr()) == L.EM_ERR_BAD_ARG


def test_block_att_c_repeatable_at_bench_shape(lib):
    """block<ATT|C> at B = 32 x T = 249 twenty times back to back: bit-identical outputs (operands are requested a tile ahead on
    counted waits; a read that raced its request would show up as run-to-run differences)."""
    B, T = 32, 249
    ly = Layer(1200, 256)
    at = _attention_inputs(B, T, [T - (b % 5) * 11 for b in range(B)], 60)
    x0 = rnd(B * T, D, seed=3)
    palld = dev(at["pall"].to(BF))
    x0d, xd = dev(x0), dev(x0.clone())
    glu = torch.zeros(B * T, D, dtype=BF, device="cuda")
    ppk, npg = packed_pos(lib, palld, T, at["Lb"])
    a = block_args(B, T, 256, x=xd, glu=glu, qh=dev(at["qh"]), kh=dev(k_to_frag(at["kh"])), vt=dev(vt_to_frag(at["vt"])),
                   wout=dev(pack_k_units(ly.wout).to(BF)),
                   pw1f=dev(pack_k_units(ly.pw1[ly.perm()]).to(BF)), params=dev(ly.c_group()), pos_u=dev(at["u"]), pos_v=dev(at["v"]),
                   klens=dev(torch.tensor([T - (b % 5) * 11 for b in range(B)], dtype=torch.int32)))
    a.pos, a.ldp, a.kv_frag = ppk.data_ptr() + 4 * npg * 2048, npg, 1
    first = None
    for it in range(20):
        xd.copy_(x0d)
        L.check(lib.em_conformer_block_fused(L.EM_BLOCK_ATT | L.EM_BLOCK_C, a, L.current_stream_ptr()), "block<ATT|C>")
        got = (xd.clone(), glu.clone())
        if first is None:
            first = got
            x_ref, glu_ref = ly.part_c(x0, q(at["ctx"]))
            assert_close(xd, x_ref, 4e-3, "block<ATT|C> x at B=32")
            assert_close(glu, glu_ref, 2e-2, "block<ATT|C> glu at B=32")
        else:
            for g0, g1 in zip(first, got):
                assert torch.equal(g0, g1), f"run {it} differs from run 0"


@pytest.mark.parametrize("B,T,ff", SHAPES)
@pytest.mark.parametrize("masked", [False, True])
def test_block_d_final(lib, B, T, ff, masked):
    ly = Layer(300, ff)
    x0, glu = rnd(B * T, D, seed=4), q(rnd(B * T, D, seed=5))
    tl = [max(1, T - 7 * b - 3) for b in range(B)] if masked else None
    ag, ab = 1 + 0.1 * rnd(D, seed=6), 0.1 * rnd(D, seed=7)
    xf = ly.part_d(x0, glu, B, T, tl)
    ref = ln(xf, ag, ab)
    xd = dev(x0.clone())
    out = torch.zeros(B * T, D, dtype=torch.float32, device="cuda")
    act = torch.zeros(B * T, D, dtype=BF, device="cuda")
    a = block_args(B, T, ff, x=xd, glu=dev(glu.to(BF)), enc_out=out, enc_act=act, pw2=dev(pack_k_units(ly.pw2).to(BF)),
                   ff_w1=dev(pack_w1(ly.ff_w1).to(BF)), ff_w2=dev(pack_w2(ly.ff_w2).to(BF)), dw_w=dev(ly.dw_w), dw_b=dev(ly.dw_b),
                   tlens=dev(torch.tensor(tl, dtype=torch.int32)) if masked else None,
                   params=dev(torch.cat(ly.d_groups() + [group(ag, ab), torch.zeros(G)])))
    L.check(lib.em_conformer_block_fused(L.EM_BLOCK_D | L.EM_BLOCK_FINAL, a, L.current_stream_ptr()), "block<D|F>")
    assert_close(out, ref, 4e-3, "block<D|FINAL> enc_out")
    assert_close(act, ref, 2e-2, "block<D|FINAL> enc_act")


@pytest.mark.parametrize("B,T,ff", SHAPES)
def test_block_da(lib, B, T, ff):
    l0, l1 = Layer(400, ff), Layer(500, ff)
    x0, glu = rnd(B * T, D, seed=8), q(rnd(B * T, D, seed=9))
    x_ref, qkv_ref = l1.part_a(l0.part_d(x0, glu, B, T))
    Tp = tpad(T)
    xd = dev(x0.clone())
    qh, kh = (torch.zeros(B, H, Tp, 64, dtype=BF, device="cuda") for _ in range(2))
    vt = torch.zeros(B, H, 64, Tp, dtype=BF, device="cuda")
    a = block_args(B, T, ff, x=xd, glu=dev(glu.to(BF)), qh=qh, kh=kh, vt=vt, pw2=dev(pack_k_units(l0.pw2).to(BF)),
                   ff_w1=dev(pack_w1(l0.ff_w1).to(BF)), ff_w2=dev(pack_w2(l0.ff_w2).to(BF)), dw_w=dev(l0.dw_w), dw_b=dev(l0.dw_b),
                   ffm_w1=dev(pack_w1(l1.ffm_w1).to(BF)), ffm_w2=dev(pack_w2(l1.ffm_w2).to(BF)), wqkv=dev(pack_k_units(l1.wqkv).to(BF)),
                   params=dev(torch.cat(l0.d_groups() + l1.a_groups())))
    L.check(lib.em_conformer_block_fused(L.EM_BLOCK_D | L.EM_BLOCK_A, a, L.current_stream_ptr()), "block<D|A>")
    assert_close(xd, x_ref, 6e-3, "block<D|A> x")
    assert_close(split_heads(qh, B, T), qkv_ref[:, :D], 3e-2, "block<D|A> q")
    assert_close(split_heads(kh, B, T), qkv_ref[:, D:2 * D], 3e-2, "block<D|A> k")
    assert_close(vt[:, :, :, :T].permute(0, 3, 1, 2).reshape(B * T, D), qkv_ref[:, 2 * D:], 3e-2, "block<D|A> v")


@pytest.mark.parametrize("B,T,ff", SHAPES)
@pytest.mark.parametrize("masked", [False, True])
def test_block_cda_folded(lib, B, T, ff, masked):
    """Round 4: block<C|D|A> - the C part computed inside the launch that consumes it, for the workgroup's 32 frames and
    the depthwise conv's halo either side; the residual stream goes x -> x_out.  Same restatement as the three-launch
    sequence (part_c -> part_d -> next layer's part_a), utterance boundaries and a masked tail (tlens) included."""
    l0, l1 = Layer(800, ff), Layer(900, ff)
    x0, ctx = rnd(B * T, D, seed=12), q(rnd(B * T, D, seed=13))
    tl = [max(1, T - 7 * b - 3) for b in range(B)] if masked else None
    xc, glu_ref = l0.part_c(x0, ctx)
    x_ref, qkv_ref = l1.part_a(l0.part_d(xc, glu_ref, B, T, tl))
    Tp = tpad(T)
    xd, xo = dev(x0.clone()), torch.full((B * T, D), float("nan"), device="cuda")
    qh, kh = (torch.zeros(B, H, Tp, 64, dtype=BF, device="cuda") for _ in range(2))
    vt = torch.zeros(B, H, 64, Tp, dtype=BF, device="cuda")
    a = block_args(B, T, ff, x=xd, x_out=xo, ctx=dev(ctx.to(BF)), qh=qh, kh=kh, vt=vt,
                   wout=dev(pack_k_units(l0.wout).to(BF)), pw1f=dev(pack_k_units(l0.pw1[l0.perm()]).to(BF)),
                   params_c=dev(l0.c_group()), pw2=dev(pack_k_units(l0.pw2).to(BF)),
                   ff_w1=dev(pack_w1(l0.ff_w1).to(BF)), ff_w2=dev(pack_w2(l0.ff_w2).to(BF)), dw_w=dev(l0.dw_w), dw_b=dev(l0.dw_b),
                   ffm_w1=dev(pack_w1(l1.ffm_w1).to(BF)), ffm_w2=dev(pack_w2(l1.ffm_w2).to(BF)), wqkv=dev(pack_k_units(l1.wqkv).to(BF)),
                   tlens=dev(torch.tensor(tl, dtype=torch.int32)) if masked else None,
                   params=dev(torch.cat(l0.d_groups() + l1.a_groups())))
    mode = L.EM_BLOCK_C | L.EM_BLOCK_D | L.EM_BLOCK_A
    L.check(lib.em_conformer_block_fused(mode, a, L.current_stream_ptr()), "block<C|D|A>")
    assert torch.equal(xd.cpu(), x0), "the folded kernel must not write its input residual"
    assert_close(xo, x_ref, 8e-3, "block<C|D|A> x_out")
    assert_close(split_heads(qh, B, T), qkv_ref[:, :D], 3e-2, "block<C|D|A> q")
    assert_close(split_heads(kh, B, T), qkv_ref[:, D:2 * D], 3e-2, "block<C|D|A> k")
    assert_close(vt[:, :, :, :T].permute(0, 3, 1, 2).reshape(B * T, D), qkv_ref[:, 2 * D:], 3e-2, "block<C|D|A> v")
    # in place is refused: a neighbour workgroup reads these rows as its halo
    a.x_out = a.x
    assert lib.em_conformer_block_fused(mode, a, L.current_stream_ptr()) == L.EM_ERR_BAD_ARG


@pytest.mark.parametrize("B,T,ff", [(2, 249, 1024), (3, 70, 256), (2, 31, 64)])
def test_block_cd_final_folded(lib, B, T, ff):
    ly = Layer(1000, ff)
    x0, ctx = rnd(B * T, D, seed=14), q(rnd(B * T, D, seed=15))
    ag, ab = 1 + 0.1 * rnd(D, seed=16), 0.1 * rnd(D, seed=17)
    ref = ln(ly.part_d(*ly.part_c(x0, ctx), B, T), ag, ab)
    out = torch.zeros(B * T, D, dtype=torch.float32, device="cuda")
    act = torch.zeros(B * T, D, dtype=BF, device="cuda")
    a = block_args(B, T, ff, x=dev(x0.clone()), ctx=dev(ctx.to(BF)), enc_out=out, enc_act=act,
                   wout=dev(pack_k_units(ly.wout).to(BF)), pw1f=dev(pack_k_units(ly.pw1[ly.perm()]).to(BF)),
                   params_c=dev(ly.c_group()), pw2=dev(pack_k_units(ly.pw2).to(BF)),
                   ff_w1=dev(pack_w1(ly.ff_w1).to(BF)), ff_w2=dev(pack_w2(ly.ff_w2).to(BF)), dw_w=dev(ly.dw_w), dw_b=dev(ly.dw_b),
                   params=dev(torch.cat(ly.d_groups() + [group(ag, ab), torch.zeros(G)])))
    L.check(lib.em_conformer_block_fused(L.EM_BLOCK_C | L.EM_BLOCK_D | L.EM_BLOCK_FINAL, a, L.current_stream_ptr()),
            "block<C|D|FINAL>")
    assert_close(out, ref, 6e-3, "block<C|D|FINAL> enc_out")
    assert_close(act, ref, 2e-2, "block<C|D|FINAL> enc_act")


def test_block_fused_rejects_outside_its_shapes(lib):
    x = torch.zeros(64, D, device="cuda")
    par = torch.zeros(4 * G, device="cuda")
    a = block_args(1, 64, 2048, x=x, params=par)
    assert lib.em_conformer_block_fused(L.EM_BLOCK_A, a, L.current_stream_ptr()) == L.EM_ERR_UNSUPPORTED
    a = block_args(1, 64, 1024, x=x, params=par)  # missing operands
    assert lib.em_conformer_block_fused(L.EM_BLOCK_A, a, L.current_stream_ptr()) == L.EM_ERR_BAD_ARG
    assert lib.em_conformer_block_fused(3, a, L.current_stream_ptr()) == L.EM_ERR_BAD_ARG


def test_block_repeatable_under_load(lib):
    """The weight stream runs three units ahead of the MFMAs on counted vmcnt waits, the H tiles and LayerNorm partials
    change hands between waves at barriers, and the L2 warm-up's share depends on arrival order: run the D|A kernel
    many times back to back at the bench shape and require bit-identical outputs (a read that raced its producer
    would show up as run-to-run differences)."""
    B, T, ff = 32, 249, 1024
    l0, l1 = Layer(600, ff), Layer(700, ff)
    x0, glu = rnd(B * T, D, seed=10), dev(q(rnd(B * T, D, seed=11)).to(BF))
    Tp = tpad(T)
    qh, kh = (torch.zeros(B, H, Tp, 64, dtype=BF, device="cuda") for _ in range(2))
    vt = torch.zeros(B, H, 64, Tp, dtype=BF, device="cuda")
    xd = dev(x0.clone())
    x0d = dev(x0)
    a = block_args(B, T, ff, x=xd, glu=glu, qh=qh, kh=kh, vt=vt, pw2=dev(pack_k_units(l0.pw2).to(BF)),
                   ff_w1=dev(pack_w1(l0.ff_w1).to(BF)), ff_w2=dev(pack_w2(l0.ff_w2).to(BF)), dw_w=dev(l0.dw_w), dw_b=dev(l0.dw_b),
                   ffm_w1=dev(pack_w1(l1.ffm_w1).to(BF)), ffm_w2=dev(pack_w2(l1.ffm_w2).to(BF)), wqkv=dev(pack_k_units(l1.wqkv).to(BF)),
                   params=dev(torch.cat(l0.d_groups() + l1.a_groups())))
    first = None
    for it in range(20):
        xd.copy_(x0d)
        L.check(lib.em_conformer_block_fused(L.EM_BLOCK_D | L.EM_BLOCK_A, a, L.current_stream_ptr()), "block<D|A>")
        got = (xd.clone(), qh.clone(), kh.clone(), vt.clone())
        if first is None:
            first = got
            x_ref, _ = l1.part_a(l0.part_d(x0, glu.float().cpu(), B, T))
            assert_close(xd, x_ref, 6e-3, "block<D|A> x at B=32")
        else:
            for g0, g1 in zip(first, got):
                assert torch.equal(g0, g1), f"run {it} differs from run 0"
