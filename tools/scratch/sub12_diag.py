import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from espnet_amd import lib as L
from oracle import conformer as oc
from espnet_amd.asr.encoder.conformer_encoder import pack_conv1_frags, pack_conv2_frags
lib = L.load()
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed); return torch.randn(*shape, generator=g) * scale
def q(t): return t.to(torch.bfloat16).float()
for (T_f, D, with_mvn) in [(61, 80, True), (133, 80, False), (61, 80, False)]:
    B, d = 3, 256
    flens = torch.tensor([T_f, max(7, T_f - 9), 7])
    feats = (rnd(B, T_f, D, seed=31) * 2 - 8).masked_fill(oc.make_pad_mask(flens, T_f)[:, :, None], 0.0)
    fl = flens.to(torch.int32).cuda()
    partial = torch.empty(B, 8, D, device="cuda")
    fd = feats.cuda()
    L.check(lib.em_utt_mvn_partial_f32(L.ptr(fd), L.ptr(fl), B, T_f, D, L.ptr(partial), None))
    w1, b1 = rnd(d, 1, 3, 3, seed=32, scale=1 / 3), rnd(d, seed=33, scale=0.1)
    w2 = q(rnd(d, d, 3, 3, seed=34, scale=(9 * d) ** -0.5)); b2 = rnd(d, seed=35, scale=0.1)
    x = oc.utterance_mvn(feats, flens) if with_mvn else feats
    c1f = F.relu(F.conv2d(x.unsqueeze(1), w1, b1, stride=2))
    c1 = q(c1f)
    ref = F.relu(F.conv2d(c1, w2, b2, stride=2)).permute(0, 2, 3, 1)
    T2, F2 = ref.shape[1], ref.shape[2]
    out = torch.full((B, T2, F2, d), 7.0, dtype=torch.bfloat16, device="cuda")
    w1f = pack_conv1_frags(w1.reshape(d, 9), b1).to(torch.bfloat16).cuda()
    w2f = pack_conv2_frags(w2.permute(0, 2, 3, 1).reshape(d, 9 * d)).to(torch.bfloat16).cuda()
    b2d = b2.cuda()
    L.check(lib.em_conv2d_sub12_bf16(L.ptr(fd), L.ptr(partial) if with_mvn else None, L.ptr(fl), B, T_f, D, L.ptr(w1f), L.ptr(w2f), L.ptr(b2d), d, L.ptr(out), None))
    # unfused device path
    T1, F1 = c1.shape[2], c1.shape[3]
    c1d = torch.zeros(B, T1, F1, d, dtype=torch.bfloat16, device="cuda")
    w1d, b1d = w1.reshape(d, 9).contiguous().cuda(), b1.cuda()
    L.check(lib.em_conv2d_sub1(L.EM_BF16, L.ptr(fd), L.ptr(partial) if with_mvn else None, L.ptr(fl), B, T_f, D, L.ptr(w1d), L.ptr(b1d), d, L.ptr(c1d), None))
    out2 = torch.zeros(B * T2 * F2, d, dtype=torch.bfloat16, device="cuda")
    wp = w2.permute(0, 2, 3, 1).reshape(d, 9 * d).to(torch.bfloat16).cuda()
    a = L.EmGemmArgs(A=c1d.data_ptr(), W=wp.data_ptr(), C=out2.data_ptr(), bias=b2d.data_ptr(), M=B * T2 * F2, N=d, K=9 * d, lda=0, ldc=d, scale=1.0)
    a.T1, a.F1, a.T2, a.F2, a.d = T1, F1, T2, F2, d
    L.check(lib.em_gemm(L.EM_BF16, L.EM_EPI_RELU, L.EM_A_CONV2, a, None))
    torch.cuda.synchronize()
    o1, o2 = out.float().cpu(), out2.float().cpu().reshape(B, T2, F2, d)
    c1dev = c1d.float().cpu().permute(0, 3, 1, 2)
    print(f"== T_f {T_f} D {D} mvn {with_mvn}: ref scale {ref.abs().max():.3f}  c1 dev-vs-ref mismatches {(c1dev != c1).float().mean():.5f}")
    for name, o in (("fused", o1), ("unfused", o2)):
        err = (o - ref).abs(); tol = 2.0 ** -7 * ref.abs().clamp_min(0.05)
        bad = err > tol
        print(f"  {name}: max err {err.max():.4e} frac>tol {bad.float().mean():.5f} per-utt {[round(bad[b].float().mean().item(),5) for b in range(B)]} per-t2 {[round(bad[:,t].float().mean().item(),4) for t in range(T2)]}")
        print(f"     per 32-ch chunk of out {[round(bad[...,c*32:(c+1)*32].float().mean().item(),4) for c in range(8)]} per f2 {[round(bad[:,:,f].float().mean().item(),4) for f in range(F2)]}")
    print("  fused vs unfused max diff", (o1 - o2).abs().max().item())
