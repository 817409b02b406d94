"""Lane-level emulation (NumPy, float64, no GPU) of the index arithmetic of colconv_deconv1_fused_kernel (colconv_wreg.hip)
and deconv1_mfma_kernel: the packed weight fragments, both MFMA stages with the operand / accumulator lane layouts of
v_mfma_f32_16x16x32, the per-wave LDS shift-add (P pieces, carry), runs with a recomputed block and the output-row edges,
against the direct formula  G = conv2^T(hid),  y[4 x + u] += G[:, :, x] W1[:, u].  Run:  python scripts/emu_fused_decoder.py [W F runs]
It checks the algorithm the kernels implement, not the kernels (tests/ does that on the GPU)."""
import sys
import numpy as np
rs = np.random.RandomState(0)
KH, H = 20, 11
HO, PH = H + KH - 1, KH - 1
nf = 30
W = int(sys.argv[1]) if len(sys.argv) > 1 else 40; F = int(sys.argv[2]) if len(sys.argv) > 2 else 4 * W + 29
n_xb = (W + 15) // 16
hid = rs.randn(nf, H, W)                      # [co][h][x]
W2t = rs.randn(KH, nf, nf)                    # Wcol_t_h[u][out ci][in co]
W1p = np.zeros((nf, 32)); W1p[:, :30] = rs.randn(nf, 30)
bias = np.zeros(32)

# direct
G = np.zeros((nf, HO, W))
for t in range(HO):
    for h in range(H):
        u = h - t + PH
        if 0 <= u < KH:
            G[:, t, :] += W2t[u] @ hid[:, h, :]
ydir = np.zeros((HO, F + 64))
for x in range(W):
    for tap in range(32):
        ydir[:, 4 * x + tap] += (G[:, :, x] * W1p[:, tap][:, None]).sum(0)
ydir = ydir[:, :F]

# packs
Wh = np.zeros((KH, 32, 40)); Wh[:, :nf, :nf] = W2t
Wq = np.zeros((KH, 2, 64, 8))
for u in range(KH):
    for half in range(2):
        for lane in range(64):
            for j in range(8):
                Wq[u, half, lane, j] = Wh[u, (lane & 15) + 16 * half, (lane >> 4) * 8 + j]
Wq1 = np.zeros((2, 64, 8))
for mh in range(2):
    for lane in range(64):
        for j in range(8):
            fi, kg = lane & 15, lane >> 4
            ci = 4 * kg + j if j < 4 else 16 + 4 * kg + (j - 4)
            tap = 4 * ((fi >> 2) + 4 * mh) + (fi & 3)
            Wq1[mh, lane, j] = W1p[ci, tap] if ci < nf else 0.0

def mma(a, b, c):
    """a[64][8]: lane (i = lane&15, kg) -> A[i][8kg+j]; b[64][8]: lane (n, kg) -> B[8kg+j][n]; c[64][4]: C[4(lane>>4)+e][lane&15]"""
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for lane in range(64):
        i, kg = lane & 15, lane >> 4
        A[i, 8 * kg:8 * kg + 8] = a[lane]
        B[8 * kg:8 * kg + 8, i] = b[lane]
    C = A @ B
    out = c.copy()
    for lane in range(64):
        for e in range(4):
            out[lane, e] += C[4 * (lane >> 4) + e, lane & 15]
    return out

out = np.full((HO, F), np.nan)
rpi = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for rr in range(rpi):
    Pb = np.zeros((2, 8, 32, 4)); Cb = np.zeros((HO, 8, 4))
    b_lo, b_hi = rr * n_xb // rpi, (rr + 1) * n_xb // rpi
    b_first = b_lo - 1 if b_lo > 0 else 0
    for b in range(b_first, b_hi):
        a = np.zeros((H, 64, 8))
        for lane in range(64):
            fi, kq = lane & 15, lane >> 4
            xl = b * 16 + fi; xc = min(xl, W - 1)
            for j in range(8):
                c = min(kq * 8 + j, nf - 1)
                a[:, lane, j] = hid[c, :, xc]
        keep = b >= b_lo
        for y in range(0, HO, 2):
            for t in range(2):
                acc = [np.zeros((64, 4)), np.zeros((64, 4))]
                for h in range(H):
                    u = h - (y + t) + PH
                    if 0 <= u < KH:
                        acc[0] = mma(Wq[u, 0], a[h], acc[0]); acc[1] = mma(Wq[u, 1], a[h], acc[1])
                gv = np.zeros((64, 8))
                for lane in range(64):
                    kq = lane >> 4
                    for e in range(4):
                        gv[lane, e] = acc[0][lane, e] + bias[4 * kq + e]
                        gv[lane, 4 + e] = acc[1][lane, e] + bias[16 + 4 * kq + e]
                for mh in range(2):
                    P = mma(Wq1[mh], gv, np.zeros((64, 4)))
                    for lane in range(64):
                        fi, kq = lane & 15, lane >> 4
                        x_ok = b * 16 + fi < W
                        flat = Pb.reshape(-1, 4)
                        flat[kq * 32 + 8 + fi + t * 256 + mh * 128] = P[lane] if x_ok else 0
            flat = Pb.reshape(-1, 4)
            newC = {}
            for lane in range(64):
                rt, rq = lane >> 5, lane & 31
                base = rt * 256 + (rq if rq < 23 else 22) + 8
                ssum = np.zeros(4)
                for mm in range(8):
                    ssum = ssum + flat[base + mm * 32 - mm]
                cin = Cb[y + rt, rq & 7].copy()
                if rq < 8: ssum = ssum + cin
                if 16 <= rq < 24: newC[(y + rt, rq - 16)] = ssum if rq < 23 else np.zeros(4)
                if keep and rq < 16:
                    f0 = 4 * (b * 16 + rq)
                    for e in range(4):
                        if f0 + e < F:
                            assert np.isnan(out[y + rt, f0 + e]), "double write"
                            out[y + rt, f0 + e] = ssum[e]
            for k, v in newC.items(): Cb[k] = v
    if b_hi == n_xb:
        for i in range((HO + 7) // 8):
            for lane in range(64):
                t = (lane >> 3) + 8 * i; f = 4 * (16 * n_xb + (lane & 7))
                if t < HO:
                    for e in range(4):
                        if f + e < F:
                            assert np.isnan(out[t, f + e]); out[t, f + e] = Cb[t, lane & 7, e]
print("nan left:", int(np.isnan(out).sum()), " max |err|:", float(np.nanmax(np.abs(out - ydir))), " scale", float(np.abs(ydir).max()))
