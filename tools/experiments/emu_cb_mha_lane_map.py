"""Numpy replay of cb_mha_heads_mfma_kernel's index arithmetic (csrc/streaming.hip) under the 16x16x32 MFMA operand layout (A: lane (lr, lg)
holds A[row lr][k = 8 lg .. 8 lg + 7]; B: B[k = 8 lg ..][col lr]; C: C[row 4 lg + r][col lr]) against plain attention, slots past the block
poisoned with NaN: run before the kernel first saw a GPU (round 5).  python tools/experiments/emu_cb_mha_lane_map.py"""
import numpy as np
rng = np.random.default_rng(0)
def mfma(A, B, C):
    # A[lane][8], B[lane][8], C[lane][4]; lane=(lg*16+lr)
    Am = np.zeros((16,32)); Bm = np.zeros((32,16))
    for lane in range(64):
        lr, lg = lane & 15, lane >> 4
        for e in range(8):
            Am[lr, 8*lg+e] = A[lane][e]
            Bm[8*lg+e, lr] = B[lane][e]
    Cm = Am @ Bm
    out = np.array(C, dtype=float).copy()
    for lane in range(64):
        lr, lg = lane & 15, lane >> 4
        for r in range(4):
            out[lane][r] += Cm[4*lg+r, lr]
    return out
def run(L, Tpad, H, mask_mode, n_blk=2):
    qh = rng.standard_normal((n_blk,H,Tpad,64)); kh = rng.standard_normal((n_blk,H,Tpad,64)); vt = rng.standard_normal((n_blk,H,64,Tpad))
    # poison rows >= L
    qh[:,:,L:] = np.nan; kh[:,:,L:] = np.nan; vt[:,:,:,L:] = np.nan
    ctx = np.full((n_blk*L, H*64), 123.0)
    nkeys = L-1 if mask_mode else L
    for blk in range(n_blk):
      for h in range(H):
        qb = qh[blk,h].reshape(-1); kb = kh[blk,h].reshape(-1); vb = vt[blk,h].reshape(-1)
        for qt in range(4):
            if 16*qt >= L: continue
            qf = np.zeros((2,64,8)); kf = np.zeros((4,2,64,8)); vraw = np.zeros((4,2,2,64,4))
            for lane in range(64):
                lr, lg = lane & 15, lane >> 4
                for ks in range(2):
                    o = (16*qt+lr)*64 + ks*32 + lg*8
                    qf[ks,lane] = qb[o:o+8]*0.125
                    for nf in range(4):
                        o2 = (16*nf+lr)*64 + ks*32 + lg*8
                        kf[nf,ks,lane] = kb[o2:o2+8]
                for f in range(4):
                    for jp in range(2):
                        for hf in range(2):
                            o = (16*f+lr)*Tpad + 32*jp + 16*hf + 4*lg
                            vraw[f,jp,hf,lane] = vb[o:o+4]
            sc = np.zeros((4,64,4))
            for ks in range(2):
                for nf in range(4):
                    sc[nf] = mfma(kf[nf,ks], qf[ks], sc[nf])
            tm = np.full(64, -np.inf)
            for lane in range(64):
                lr, lg = lane & 15, lane >> 4
                for nf in range(4):
                    for r in range(4):
                        if not (16*nf+4*lg+r < nkeys): sc[nf,lane,r] = -np.inf
                        tm[lane] = np.fmax(tm[lane], sc[nf,lane,r])
            tm2 = tm.copy()
            for lane in range(64):
                lr = lane & 15
                tm2[lane] = max(tm[lr+16*g] for g in range(4))  # fmax ignoring nan: emulate
            pb = np.zeros((2,64,8))
            for lane in range(64):
                for nf in range(4):
                    for hh in range(2):
                        for i in range(2):
                            e = np.exp(sc[nf,lane,2*hh+i]-tm2[lane])
                            slot = ((nf&1)*2+hh)*2 + i
                            pb[nf>>1,lane,slot] = e
            acc_o = np.zeros((4,64,4)); acc_l = np.zeros((64,4))
            ones = np.ones((64,8))
            for jp in range(2):
                acc_l = mfma(ones, pb[jp], acc_l)
                for f in range(4):
                    vf = np.zeros((64,8))
                    for lane in range(64):
                        lg = lane >> 4
                        for hf in range(2):
                            nv = nkeys - (32*jp+16*hf+4*lg)
                            for i in range(4):
                                vf[lane, 4*hf+i] = vraw[f,jp,hf,lane,i] if i < nv else 0.0
                    acc_o[f] = mfma(vf, pb[jp], acc_o[f])
            for lane in range(64):
                lr, lg = lane & 15, lane >> 4
                q = 16*qt+lr
                if q < L:
                    none = mask_mode and q == 0
                    for f in range(4):
                        for r in range(4):
                            ctx[blk*L+q, h*64+16*f+4*lg+r] = 0.0 if none else acc_o[f,lane,r]/acc_l[lane,0]
    # reference
    ref = np.zeros_like(ctx)
    for blk in range(n_blk):
        for h in range(H):
            q = qh[blk,h,:L]; k = kh[blk,h,:nkeys]; v = vt[blk,h,:,:nkeys].T
            s = q @ k.T / 8.0
            p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
            o = p @ v
            if mask_mode: o[0] = 0
            ref[blk*L:(blk+1)*L, h*64:(h+1)*64] = o
    return np.abs(ctx-ref).max()
for L, mm in ((42,1),(42,0),(24,0),(64,1),(17,1),(3,1),(33,0)):
    print(L, mm, run(L, 64, 2, mm))
