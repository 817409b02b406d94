"""CPU checks of the one-batch kernels' index arithmetic (csrc/dsd_lat.hip): the weight re-layouts come from the library's
own packers (host-only entry points), the kernels' lane arithmetic is emulated in NumPy -- v_mfma_f32_16x16x4_f32
semantics: lane (fi = lane & 15, kq = lane >> 4) supplies A[fi][kq] and B[kq][fi], the block is D[i][j] += sum_kq
A[i][kq] B[kq][j] -- and compared with the plain formula of the layer.  No GPU."""
import ctypes

import numpy as np
import pytest

from deepconvsep_amd import _lib


def _pack_b(B, K, n_cb, slice_len, n_slices):
    lib = _lib.load()
    B = np.ascontiguousarray(B, dtype=np.float32)
    n = lib.dcs_lat_pack_b_host(B.ctypes.data_as(ctypes.c_void_p), B.shape[1], K, n_cb, slice_len, n_slices, None, 0)
    assert n > 0
    out = np.empty(n, dtype=np.float32)
    assert lib.dcs_lat_pack_b_host(B.ctypes.data_as(ctypes.c_void_p), B.shape[1], K, n_cb, slice_len, n_slices,
                                   out.ctypes.data_as(ctypes.c_void_p), n) == n
    J = (slice_len + 15) // 16
    return out.reshape(n_slices, n_cb, J, 64, 4), J


def _emu_gemm(A_flat, row_stride, M, K, Bp, J, slice_len, n_slices, n_cb):
    """lat_gemm_kernel, lane by lane."""
    C = np.zeros((M, n_cb * 16), dtype=np.float64)
    lanes = np.arange(64)
    fi, kq = lanes & 15, lanes >> 4
    for rb in range((M + 15) // 16):
        rows = rb * 16 + fi
        row_ok = rows < M
        for cb in range(n_cb):
            blk = np.zeros((16, 16))
            for s in range(n_slices):
                for j in range(J):
                    kl = 16 * j + 4 * kq
                    ok = row_ok & (kl < slice_len) & (s * slice_len + kl < K)
                    for e in range(4):
                        a = np.zeros(64)
                        idx = np.where(ok, rows, 0) * row_stride + s * slice_len + kl + e
                        a[ok] = A_flat[idx[ok]]
                        b = Bp[s, cb, j, :, e].astype(np.float64)
                        Am = np.zeros((16, 4)); Bm = np.zeros((4, 16))
                        Am[fi, kq] = a
                        Bm[kq, fi] = b
                        blk += Am @ Bm
            r0 = rb * 16
            nr = min(16, M - r0)
            C[r0:r0 + nr, cb * 16:(cb + 1) * 16] = blk[:nr]
    return C


@pytest.mark.parametrize("case", ["conv1_1025", "conv1_513", "conv2", "fc", "fc1x"])
def test_sliced_gemm_index_arithmetic(case):
    rng = np.random.RandomState(3)
    if case.startswith("conv1"):
        F = int(case.split("_")[1]); ld = (F + 3) // 4 * 4
        M, K, n_cb, n_slices = 37, ld, 4, 16
        slice_len = ((K + 15) // 16 + 3) // 4 * 4
        A = rng.randn(M, ld); A[:, F:] = 0
        B = np.zeros(((K + 127) // 128 * 128, 64)); B[:F, :50] = rng.randn(F, 50)
        A_flat, stride = A.reshape(-1), ld
        want = A @ B[:K]
    elif case == "conv2":      # position p: 15 consecutive H1 rows of 52 floats, one slice per tap
        rows1, CI = 40, 52
        H1 = rng.randn(rows1, CI)
        M, K, n_cb, n_slices, slice_len = rows1 - 14, 15 * CI, 4, 15, CI
        B = np.zeros((896, 64)); B[:K, :50] = rng.randn(K, 50)
        A_flat, stride = H1.reshape(-1), CI
        want = np.stack([H1[p:p + 15].reshape(-1) for p in range(M)]) @ B[:K]
    elif case == "fc":         # tile k: 16 consecutive C2 rows from row 5 k
        n, CP, st = 19, 52, 5
        C2 = rng.randn((n - 1) * st + 16, CP)
        M, K, n_cb, n_slices, slice_len = n, 16 * CP, 8, 16, CP
        B = np.zeros((896, 128)); B[:K] = rng.randn(K, 128)
        A_flat, stride = C2.reshape(-1), st * CP
        want = np.stack([C2[k * st:k * st + 16].reshape(-1) for k in range(n)]) @ B[:K]
    else:
        n = 21
        Z = rng.randn(n, 128)
        M, K, n_cb, n_slices, slice_len = n, 128, 6, 4, 32
        B = rng.randn(128, 96)
        A_flat, stride = Z.reshape(-1), 128
        want = Z @ B
    Bp, J = _pack_b(B, K, n_cb, slice_len, n_slices)
    got = _emu_gemm(A_flat.astype(np.float32).astype(np.float64), stride, M, K, Bp, J, slice_len, n_slices, n_cb)
    # B went through float32 in the packer
    Bq = np.asarray(B, dtype=np.float32).astype(np.float64)
    if case.startswith("conv1"):
        want = A.astype(np.float32).astype(np.float64) @ Bq[:K]
    elif case == "conv2":
        H = H1.astype(np.float32).astype(np.float64)
        want = np.stack([H[p:p + 15].reshape(-1) for p in range(M)]) @ Bq[:K]
    elif case == "fc":
        Cq = C2.astype(np.float32).astype(np.float64)
        want = np.stack([Cq[k * st:k * st + 16].reshape(-1) for k in range(n)]) @ Bq[:K]
    else:
        want = Z.astype(np.float32).astype(np.float64) @ Bq
    np.testing.assert_allclose(got, want[:, :n_cb * 16], rtol=0, atol=1e-9)


def test_transposed_conv2_gemm_col2im_index_arithmetic():
    """lat_deconv2_kernel: P[t'][dt] per channel from the packed weights, skewed store Ps[t' + dt][dt], row sum over the
    taps with the validity select -- against G[t, ci] = sum_{dt, co} D[t - dt][co] W2c[co][ci][dt]."""
    lib = _lib.load()
    rng = np.random.RandomState(5)
    CI8, CP, kh, H2, tc = 56, 52, 15, 16, 30
    W2c = np.zeros((CP, CI8, 16)); W2c[:50, :50, :kh] = rng.randn(50, 50, kh)        # [co][ci][dt]
    Bw2s = np.ascontiguousarray(W2c.transpose(1, 2, 0), dtype=np.float32)             # [ci][dt][co]
    n = lib.dcs_lat_pack_deconv2_host(Bw2s.ctypes.data_as(ctypes.c_void_p), CI8, None, 0)
    Wp = np.empty(n, dtype=np.float32)
    assert lib.dcs_lat_pack_deconv2_host(Bw2s.ctypes.data_as(ctypes.c_void_p), CI8, Wp.ctypes.data_as(ctypes.c_void_p), n) == n
    Wp = Wp.reshape(CI8, 4, 64, 4)
    D = rng.randn(H2, CP).astype(np.float32).astype(np.float64); D[:, 50:] = 0
    lanes = np.arange(64); fi, kq = lanes & 15, lanes >> 4
    W2q = Bw2s.astype(np.float64).transpose(2, 0, 1)                                   # [co][ci][dt]
    want = np.zeros((tc, CI8))
    for t in range(tc):
        for dt in range(kh):
            if 0 <= t - dt < H2:
                want[t] += D[t - dt] @ W2q[:, :, dt]
    got = np.zeros((tc, CI8))
    for ci in range(CI8):
        P = np.zeros((16, 16))
        for j in range(4):
            for e in range(4):
                c = 16 * j + 4 * kq + e
                a = np.where((j < 3) | (kq == 0), D[fi, np.minimum(c, CP - 1)], 0.0)   # a[3] is loaded by kq == 0 only
                a = np.where(c < CP, a, 0.0)
                Am = np.zeros((16, 4)); Bm = np.zeros((4, 16))
                Am[fi, kq] = a
                Bm[kq, fi] = Wp[ci, j, :, e]
                P += Am @ Bm
        Ps = np.full((32, 20), np.nan)                       # never-written entries stay NaN: the select must drop them
        for tp in range(16):
            for dt in range(16):
                Ps[tp + dt, dt] = P[tp, dt]
        for t in range(tc):
            s = 0.0
            for dt in range(15):
                if 0 <= t - dt < 16:
                    s += Ps[t, dt]
            got[t, ci] = s
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)


@pytest.mark.parametrize("N,hop", [(2048, 512), (1024, 512), (1024, 256), (2048, 1024)])
def test_istft_hop_block_decomposition(N, hop):
    """lat_istft_kernel's decomposition: output hop block b = padded hop block h = b + R/2, frames h-R+1 .. h, thread
    group q contributes segment R-1-q of frame h-R+1+q, segments added in frame order, divided by the sum of window^2 of
    the same frames -- against the reference's istft_norm (executed from the reference source)."""
    from oracle import stft_np        # restatement of transform.py:277-396, pinned bit for bit to the reference's code
    rng = np.random.RandomState(7)
    L = 5 * hop + 123
    x = rng.randn(L)
    w = np.hanning(N)
    X = stft_np.stft_norm(x, w, float(hop), float(N))
    T = X.shape[0]
    want = stft_np.istft_norm(X, w, w, float(hop), float(N))
    R = N // hop
    n_out = len(want)
    frames = np.fft.irfft(X, N)[:, :N] * w[None, :]
    got = np.zeros(n_out)
    for b in range((n_out + hop - 1) // hop):
        h = b + R // 2
        acc = np.zeros(hop); norm = np.zeros(hop)
        for q in range(R):
            t = h - (R - 1) + q
            if 0 <= t < T:
                seg = R - 1 - q
                acc += frames[t, seg * hop:(seg + 1) * hop]
                norm += (w * w)[seg * hop:(seg + 1) * hop]
        norm[norm == 0] = 1.0
        m0 = b * hop
        k = min(hop, n_out - m0)
        got[m0:m0 + k] = (acc / norm)[:k]
    np.testing.assert_allclose(got, want[:n_out], rtol=0, atol=1e-10)
