"""CPU checks of index arithmetic in csrc/generic.hip (round 3; the second test covers the column convolution's LDS layouts).
The iKala graph's fused max-pool: conv1_reg_kernel<30, 3, true>
leaves, instead of the full-resolution activations, 4 routing bits per pooling window -- a wave's 64-bit ballot of "this
position equals its window's maximum" is stored as two 32-bit words, 8 windows per word -- and deconv1_reg_kernel<3, 10, true>
rebuilds the 13 un-pooled inputs of a thread (positions 4 qb - 9 .. 4 qb + 3) from 4 pooled gradients and 16 of those bits.
The kernels' arithmetic is restated lane by lane in NumPy and compared with the plain un-pooling VJP (position f receives
g[f // 4] when a[f] equals its window maximum: every such position, or only the first).  No GPU."""
import numpy as np
import pytest

PW, NT, QB = 4, 10, 4          # pool width; taps per residue of the stride-3 conv1^T; q values per thread


def _routing_words(a, w1, wp, mw, tie_first):
    """conv1_reg_kernel<., ., true>: positions j = 64 wave + lane; quad maximum; ballot; nibble-wise first bit; two words per wave."""
    words = np.zeros(mw, dtype=np.uint64)
    for wave in range((w1 + 63) // 64):
        b = 0
        for lane in range(64):
            j = 64 * wave + lane
            if j >= w1:
                continue                                   # the lane returned early: contributes 0 to the ballot
            q0 = j & ~3
            win = [a[q] for q in range(q0, q0 + 4) if q < w1]       # an incomplete border window: only its active lanes
            if a[j] == max(win):
                b |= 1 << lane
        if tie_first:
            E, C, S8 = 0xEEEEEEEEEEEEEEEE, 0xCCCCCCCCCCCCCCCC, 0x8888888888888888
            b &= ~(((b << 1) & E) | ((b << 2) & C) | ((b << 3) & S8)) & 0xFFFFFFFFFFFFFFFF
        if 64 * wave < w1:                                 # lane 0 active
            words[2 * wave] = b & 0xFFFFFFFF
        if 64 * wave + 32 < w1:                            # lane 32 active
            words[2 * wave + 1] = b >> 32
    return words


def _thread_inputs(gp, words, wp, qb):
    """deconv1_reg_kernel<3, 10, true>: the 13 inputs of thread qb from 4 pooled values and 16 routing bits."""
    k = qb - (NT + 2) // 4
    gv = np.zeros(4)
    bits = 0
    if k >= 0 and k + 3 < wp:                              # interior: one 16-byte and one 8-byte load
        gv[:] = gp[k:k + 4]
        w64 = int(words[k >> 3]) | (int(words[(k >> 3) + 1]) << 32 if (k >> 3) + 1 < len(words) else 0)
        bits = (w64 >> (4 * (k & 7))) & 0xFFFF
    else:
        for i in range(4):
            w = k + i
            if 0 <= w < wp:
                gv[i] = gp[w]
                bits |= ((int(words[w >> 3]) >> (4 * (w & 7))) & 15) << (4 * i)
    return np.array([gv[(x + 3) >> 2] if (bits >> (x + 3)) & 1 else 0.0 for x in range(QB + NT - 1)])


@pytest.mark.parametrize("tie_first", [False, True])
@pytest.mark.parametrize("w1", [81, 92, 162, 332, 64, 65])
def test_routing_bits_rebuild_the_unpooled_gradient(w1, tie_first):
    rng = np.random.RandomState(w1 + 7 * tie_first)
    wp, mw = w1 // PW, ((w1 + 63) // 64) * 2
    a = rng.randn(w1)
    a[8:16] = 0.25                                          # two windows of exact ties
    a[21] = a[22]                                           # a tie inside a window
    if w1 > 70:
        a[64:68] = -1.0                                     # a tied window at a wave boundary
    gp = rng.randn(wp)
    # the plain VJP of the pool at a
    want = np.zeros(w1)
    for w in range(wp):
        win = a[4 * w:4 * w + 4]
        hits = np.flatnonzero(win == win.max())
        if tie_first:
            hits = hits[:1]
        want[4 * w + hits] = gp[w]
    words = _routing_words(a, w1, wp, mw, tie_first)
    nqb = (3 * (w1 - 1) + 30 + 11) // 12                    # threads per row: F = 3 (w1 - 1) + 30 bins, 12 per thread
    for qb in range(nqb + 1):
        got = _thread_inputs(gp, words, wp, qb)
        j0 = 4 * qb - (NT - 1)
        for x in range(QB + NT - 1):
            j = j0 + x
            ref = want[j] if 0 <= j < w1 else 0.0
            assert got[x] == ref, (w1, tie_first, qb, x, j)


def test_column_convolution_fragment_layouts():
    """colconv_kernel (csrc/generic.hip, round 3): the weights go from the global order [u][ci][co swizzled] (colconv_wslot) to
    LDS as Wl[u][kq][half][co 32][4] and the input slab as slab[kq][row][half][x 16][4], so that lane (fi, kq) reads its
    eight K values of a tap -- channels kq, kq + 4, ..., kq + 28 -- as two 16-byte pieces per operand.  The index arithmetic
    of the copy loops and of the fragment reads, lane by lane with v_mfma_f32_16x16x4_f32 semantics (A[i = lane & 15][k =
    lane >> 4], B[k][j = lane & 15], D[4 (lane >> 4) + e][lane & 15]), against the direct 'full' correlation."""
    rng = np.random.RandomState(11)
    kh, H, Cin, Cout, W = 20, 11, 30, 30, 16
    ph, Ho = kh - 1, H + kh - 1
    Wk = rng.randn(kh, 32, 32)
    Wk[:, Cin:, :] = 0.0
    Wk[:, :, Cout:] = 0.0
    x = rng.randn(Cin, H, W)
    # global weight order of the model: slot (u, ci, co) -> (u * 32 + ci) * 32 + ((co + 16 (ci & 1)) & 31)
    Wg = np.zeros(kh * 1024)
    for u in range(kh):
        for ci in range(32):
            for co in range(32):
                Wg[(u * 32 + ci) * 32 + ((co + 16 * (ci & 1)) & 31)] = Wk[u, ci, co]
    # the kernel's copy into LDS
    Wl = np.zeros(kh * 1024)
    for i in range(kh * 1024):
        u, ci, slot = i >> 10, (i >> 5) & 31, i & 31
        co = (slot - 16 * (ci & 1)) & 31
        kk = ci >> 2
        Wl[((((u * 4 + (ci & 3)) * 2 + (kk >> 2)) * 32 + co) << 2) + (kk & 3)] = Wg[i]
    slab = np.zeros(512 * H)
    for ci in range(Cin):
        for r in range(H):
            for xx in range(16):
                kk = ci >> 2
                slab[((((ci & 3) * H + r) * 2 + (kk >> 2)) * 16 + xx) * 4 + (kk & 3)] = x[ci, r, xx]
    Wl4, sl4 = Wl.reshape(-1, 4), slab.reshape(-1, 4)
    lanes = np.arange(64)
    fi, kq = lanes & 15, lanes >> 4
    for y in (0, 5, 10, 19, 29):
        u_lo, u_hi = max(0, ph - y), min(kh - 1, ph - y + H - 1)
        acc = np.zeros((2, 16, 16))                       # [co block][co in block][x]
        for u in range(u_lo, u_hi + 1):
            row = y + u - ph
            for half in range(2):
                b = sl4[kq * H * 32 + fi + row * 32 + half * 16]               # [lane][4]
                for blk in range(2):
                    a = Wl4[kq * 64 + fi + u * 256 + half * 32 + blk * 16]     # [lane][4]
                    for j in range(4):
                        A = np.zeros((16, 4)); B = np.zeros((4, 16))
                        A[fi, kq] = a[:, j]
                        B[kq, fi] = b[:, j]
                        acc[blk] += A @ B
        want = np.zeros((32, 16))
        for u in range(kh):
            row = y + u - ph
            if 0 <= row < H:
                want += Wk[u, :Cin, :].T @ x[:, row, :]
        np.testing.assert_allclose(acc.reshape(32, 16), want, rtol=0, atol=1e-11)


# ------------------------------------------------------------------ colconv_x3.hip (round 4): taps dealt to two waves by parity
def _x3_row0(y, s, PH, NK):
    return (y + s // NK) - PH + 2 * (s % NK)


def _x3_live(y, s, H, PH, NK):
    return -1 <= _x3_row0(y, s, PH, NK) <= H - 1


def test_x3_decoder_slot_lists_rows_and_exchange_reproduce_the_transposed_column_convolution():
    """The f32-class fused decoder keeps taps u = 2 k + par in wave `par` and runs ONE unrolled body in both waves: slot
    (t, k) of row pair (y, y + 1) multiplies tap k of the wave with LDS row h0 + 1 + par, h0 = y + t - 19 + 2 k, where rows
    -1 and 11 of the plane buffer are zero; each wave then hands its partial of the row the partner finishes through LDS.
    Restated on whole rows in NumPy -- slot liveness, the row that depends on `par`, the zero rows, the weight fragment
    packing [parity][k] and the exchange -- against the direct InverseLayer of a 'valid' 20 x 1 convolution."""
    KH, H, NK = 20, 11, 10
    HO, PH = H + KH - 1, KH - 1
    rs = np.random.RandomState(3)
    Cin, Cout, X = 30, 30, 16
    Wt = rs.randn(KH, Cout, Cin)                               # transposed filter: W[u][out][in] (dcs_decoder_x3_pack source order)
    inp = rs.randn(Cin, H, X)
    # direct: out[co][y][x] = sum_u sum_ci W[u][co][ci] in[ci][y + u - PH][x]
    want = np.zeros((Cout, HO, X))
    for y in range(HO):
        for u in range(KH):
            h = y + u - PH
            if 0 <= h < H:
                want[:, y] += Wt[u] @ inp[:, h]
    # the kernel: packed weights w[par][k] = W[2 k + par]; plane rows -1 .. H with zero guard rows
    wq = np.stack([np.stack([Wt[2 * k + par] for k in range(NK)]) for par in range(2)])
    rows = np.zeros((H + 2, Cin, X))
    rows[1:H + 1] = inp.transpose(1, 0, 2)
    got = np.zeros((Cout, HO, X))
    n_slots = [0, 0]
    for y in range(0, HO, 2):
        acc = np.zeros((2, 2, Cout, X))                        # [wave][t]
        for par in range(2):
            for s in range(2 * NK):
                if not _x3_live(y, s, H, PH, NK):
                    continue
                t, k = s // NK, s % NK
                acc[par, t] += wq[par, k] @ rows[_x3_row0(y, s, PH, NK) + 1 + par]
                n_slots[par] += 1
        for par in range(2):                                   # wave `par` finishes row y + par: its own partial + the partner's
            got[:, y + par] = acc[par, par] + acc[1 - par, par]
    assert np.max(np.abs(got - want)) < 1e-12
    # 120 slots per wave and block for 110 (row, tap) products: 10 meet a zero row
    assert n_slots == [120, 120]
    live = [(y, s) for y in range(0, HO, 2) for s in range(2 * NK) if _x3_live(y, s, H, PH, NK)]
    assert sum(1 for y, s in live if _x3_row0(y, s, PH, NK) in (-1, H - 1)) == 20     # one parity of these meets a guard row


def test_x3_decoder_channels_last_fetch_covers_every_channel_once():
    """Fetch task i of a pair (128 threads, 6 rounds): row i // 64, x = (i % 64) // 4, K piece kq = i % 4; TWO 16-byte loads per
    task -- channels 8 kq .. 8 kq + 3 and, `hi_off` floats further, four more.  For a K piece that runs past the last channel
    (kq = 3 of 30 or 28 channels) the second load starts at channel Cin - 4 and its upper half is used twice (`hi_dup`): K
    slots >= Cin meet zero weights and only have to hold finite values from INSIDE the position.  Every (row, x, channel <
    Cin) must land exactly once in K slot = its channel, no load may leave the position's Cin floats, tasks past the last
    row read row 0, and the LDS unit of plane p is ((row + 1) * 3 + p) * 64 + kq * 16 + x -- what the MFMA B fragment of
    lane 16 kq + x reads."""
    H = 11
    for Cin in (28, 30, 32):
        seen = np.zeros((H, 16, 32), dtype=int)
        units = set()
        for i in range(6 * 128):
            h, rem = i >> 6, i & 63
            x, kq = rem >> 2, rem & 3
            ok = i < H * 64
            hi_dup = 8 * kq + 8 > Cin
            hi_off = Cin - 4 - 8 * kq if hi_dup else 4
            row = h if ok else 0                                 # t_off of a task past the last row
            lo = [8 * kq + e for e in range(4)]                  # channels of the first load
            hi = [8 * kq + hi_off + e for e in range(4)]
            assert 0 <= min(lo + hi) and max(lo + hi) < Cin and 0 <= row < H     # inside the position, 8-byte aligned
            assert (8 * kq) % 2 == 0 and (8 * kq + hi_off) % 2 == 0
            if not ok:
                continue
            slots = lo + ([hi[2], hi[3], hi[2], hi[3]] if hi_dup else hi)        # channel that lands in K slot 8 kq + j
            for j, c in enumerate(slots):
                slot = 8 * kq + j
                if slot < Cin:
                    assert c == slot                              # real channels sit in their own slot
                seen[h, x, slot] += 1
            for p in range(3):
                u = ((h + 1) * 3 + p) * 64 + kq * 16 + x
                assert u not in units
                units.add(u)
        assert (seen == 1).all()
        assert min(units) == 192 and max(units) == (H + 1) * 192 - 1     # rows 0 (= -1) and 12 (= 11) of the buffer stay zero


# ------------------------------------------------------------------ slabconv_ps.hip, FAST tap loop (round 4)
def test_slabconv_ps_mask_driven_tap_loop_visits_exactly_the_old_loops_steps():
    """The iKala conv2 / conv2^T kernel walks (tap pair, 16-column block) steps; the old loop decided per step from the rows
    and columns (`r`, `xs`, `inner`, per-lane `ok`), the new one from per-block bit masks (lm: some lane inside the image,
    im: every lane inside), a live filter-row range [bu0, bu1] and a select between the lane's slab index and a record of
    zeros.  Restated for both layers' shapes (conv2: 30 x 83 -> 21 x 64 'valid'; its InverseLayer: 21 x 64 -> 30 x 83 with
    pads 9 / 19), every band the launcher can choose: same live steps, same `inner` flag, same lanes reading zeros, same
    slab element for the others."""
    KH, KW = 10, 20
    nvp = (KW + 1) // 2
    for (Hh, W, Ho, Wo, ph, pw) in ((30, 83, 21, 64, 0, 0), (21, 64, 30, 83, 9, 19)):
        nxb = (Wo + 15) // 16
        for band in (1, 4, 5):
            for y0 in range(0, Ho, band):
                yb = min(y0 + band, Ho)
                rbase = max(y0 - ph, 0)
                for by in range(y0, yb):
                    for bxi in range(nxb):
                        bx = bxi * 16
                        xs0 = bx - pw
                        lm = [(xs0 + 2 * vp + 16 >= 0 and xs0 + 2 * vp < W) for vp in range(nvp)]
                        im = [(xs0 + 2 * vp >= 0 and xs0 + 2 * vp + 17 <= W) for vp in range(nvp)]
                        bu0, bu1 = ph - by, Hh - 1 + ph - by
                        for u in range(KH):
                            for vp in range(nvp):
                                # ---- the old loop
                                r, xs = by + u - ph, bx + 2 * vp - pw
                                old_live = not (r < 0 or r >= Hh or xs + 16 < 0 or xs >= W)
                                old_inner = xs >= 0 and xs + 17 <= W
                                # ---- the new one
                                new_live = bu0 <= u <= bu1 and lm[vp]
                                assert new_live == old_live
                                if not old_live:
                                    continue
                                assert im[vp] == old_inner
                                for lane in range(64):
                                    fi, kq = lane & 15, lane >> 4
                                    xc = xs + fi + (kq >> 1)
                                    old_ok = 0 <= xc < W                      # (rows are live here)
                                    old_idx = (r - rbase) * W + xc
                                    lx = fi + (kq >> 1)
                                    vb = (by - ph - rbase) * W + xs0 + lx      # records; the kernel carries (.. * RP + (kq & 1))
                                    idx = vb + u * W + 2 * vp
                                    reads_zero = (not im[vp]) and not (0 <= xs0 + 2 * vp + lx < W)
                                    assert reads_zero == (not old_ok)
                                    if old_ok:
                                        assert idx == old_idx and idx >= 0


def test_slabconv_ps_column_strips_read_inside_their_slab_and_skip_only_dead_stages():
    """Round 6: the transposed convolution of the iKala graph (21 x 64 -> 30 x 83, 10 x 20 filter, pads 9 / 19) runs on COLUMN
    STRIPS -- a workgroup owns one 16-column block of every output row, its slab holds the input columns [cx0, cx1) that block's
    taps can reach, and weight stages (groups of `pstage` tap pairs of one filter row) that no block of the strip can use are
    skipped by the whole workgroup.  Restated from slabconv_ps_kernel: (i) every in-image read of a live step lands inside the
    slab, at the record the kernel computes (vb with the - cx0 term); (ii) a skipped stage has no live step in any block of
    the strip, so the strip's set of executed (u, tap pair, block) steps is the whole-row form's restricted to its columns."""
    KH, KW = 10, 20
    nvp = (KW + 1) // 2
    Hh, W, Ho, Wo, ph, pw = 21, 64, 30, 83, 9, 19
    for xt in (16, 32):                                   # the launcher picks 16; a two-block tile exercises nxb > 1
        n_xt = ((Wo + 15) // 16 * 16 + xt - 1) // xt
        seen_blocks = set()
        for xtile in range(n_xt):
            xt0 = xtile * xt
            xt1 = min(xt0 + xt, Wo)
            if xt0 >= Wo:
                continue
            nxb = (xt1 - xt0 + 15) >> 4
            cx0 = max(xt0 - pw, 0)
            cx1 = min(xt0 - pw + nxb * 16 + 2 * nvp - 1, W)
            SW = max(cx1 - cx0, 0)
            y0, yb = 0, Ho                                # the strip covers every output row
            rbase = max(y0 - ph, 0)
            rtop = min(yb - 1 - ph + KH - 1, Hh - 1)
            rows = rtop - rbase + 1
            for pstage in (1, 3, 5):
                nvs = (nvp + pstage - 1) // pstage
                live_stage = []
                for sidx in range(nvs):
                    i0 = sidx * pstage
                    npp = min(pstage, nvp - i0)
                    min_xs, max_xs = xt0 - pw + 2 * i0, xt0 + (nxb - 1) * 16 - pw + 2 * (i0 + npp - 1)
                    live_stage.append(max_xs + 16 >= 0 and min_xs < W)
                for by in range(y0, yb):
                    for bxi in range(nxb):
                        bx = xt0 + bxi * 16
                        seen_blocks.add((by, bx))
                        xs0 = bx - pw
                        for u in range(KH):
                            r = by + u - ph
                            for vp in range(nvp):
                                xs = xs0 + 2 * vp
                                live = 0 <= r < Hh and xs + 16 >= 0 and xs < W
                                if live:
                                    assert live_stage[vp // pstage], (xtile, by, bx, u, vp)     # (ii) never skipped
                                    for lane in range(64):
                                        fi, kq = lane & 15, lane >> 4
                                        lx = fi + (kq >> 1)
                                        xc = xs + lx
                                        if 0 <= xc < W:                                          # (i) inside the slab
                                            assert cx0 <= xc < cx1 and 0 <= r - rbase < rows
                                            vb = (by - ph - rbase) * SW + xs0 - cx0 + lx
                                            assert vb + u * SW + 2 * vp == (r - rbase) * SW + (xc - cx0)
        assert seen_blocks == {(by, bx) for by in range(Ho) for bx in range(0, Wo, 16)}          # every block owned once


# ------------------------------------------------------------------ forward STFT edge frames (round 4)
def test_stft_edge_frames_clamped_loads_and_masks_reproduce_zero_padding():
    """stft_forward_wave_kernel / lat_stft_kernel load the samples of a frame that hangs over either end of the signal from a
    CLAMPED index (always inside the clip) and zero the out-of-range ones afterwards, instead of `if (in range) x = a[p]`
    (which the compiler turned into one memory round trip per load).  Same frame as zero padding, every index inside [0, L)."""
    M, hop = 8, 4
    for L in (1, 5, 16, 23):
        a = np.arange(1, L + 1, dtype=np.float64)
        T = (L + M) // hop + 1
        for t in range(T):
            base = t * hop - M
            r_lo = -base if base < 0 else 0
            rem = L - base
            r_hi = (0 if rem < 0 else rem) if rem < 2 * M else 2 * M
            want = np.array([a[base + r] if 0 <= base + r < L else 0.0 for r in range(2 * M)])
            got = np.zeros(2 * M)
            if r_hi > r_lo:
                for r in range(2 * M):
                    rc = r_lo if r < r_lo else (r_hi - 1 if r >= r_hi else r)
                    assert 0 <= base + rc < L
                    got[r] = a[base + rc] if r_lo <= r < r_hi else 0.0
            assert np.array_equal(got, want)
            # lat_stft_kernel clamps the absolute position instead
            got2 = np.zeros(2 * M)
            for r in range(2 * M):
                p = base + r
                pc = 0 if p < 0 else (L - 1 if p >= L else p)
                got2[r] = a[pc] if 0 <= p < L else 0.0
            assert np.array_equal(got2, want)


def test_group_row_shortcut_is_the_division_it_replaces():
    """dcs_group_row: (r / gdiv) * gmul + r % gdiv, taken as r when the launch has one group (gdiv >= M)."""
    for M, gdiv, gmul in ((640, 1 << 30, 0), (640, 640, 7), (640, 32, 40), (1, 1, 5), (33, 32, 100)):
        flat = gdiv >= M
        for r in range(M):
            full = (r // gdiv) * gmul + r % gdiv
            assert (r if flat else full) == full


def test_one_batch_prologues_request_every_element_once_and_only_inside_their_arrays():
    """lat_stft_kernel / lat_ifft_kernel (dsd_lat.hip) request their twiddle table, spectrum row and samples in fully unrolled
    batches with clamped indices (round 4; the loops they replace made one memory round trip per pass).  Restated: every
    table entry 0 .. M is written by exactly one (thread, pass), every spectrum bin 0 .. M by exactly one thread group
    member, and no index leaves [0, M]."""
    for M in (512, 1024):
        NI = M // 256
        # lat_stft_kernel: 256 threads, NI passes + the entry M
        written = np.zeros(M + 1, dtype=int)
        for tid in range(256):
            for i in range(NI):
                assert 0 <= tid + 256 * i < M
                written[tid + 256 * i] += 1
            if tid == 0:
                written[M] += 1
        assert (written == 1).all()
        # lat_ifft_kernel: NG thread groups of 256 share the table, each group fills its own spectrum buffer
        for NG in (1, 2, 4):
            NT = NG * 256
            NTW = (M + NT) // NT
            written = np.zeros(M + 1, dtype=int)
            for tid in range(NT):
                for i in range(NTW):
                    k = tid + NT * i
                    src = k if k <= M else M                          # the clamped load
                    assert 0 <= src <= M
                    if k <= M:
                        written[k] += 1
            assert (written == 1).all()
            bins = np.zeros(M + 1, dtype=int)
            for gt in range(256):
                for i in range(NI + 1):
                    if i == NI and gt != 0:
                        break
                    k = gt + 256 * i if i < NI else M
                    assert 0 <= k <= M
                    bins[k] += 1
            assert (bins == 1).all()


def test_f32_gemm_reads_its_lds_operands_in_groups_of_eight_k_steps_in_the_old_order():
    """gemm_rows_kernel: the operands of eight k steps are read together (two register sets alternate) and multiplied in the
    order of the plain loop -- the accumulation order, hence the bits, are those of `for kk: acc = mfma(a[kk], b[kk], acc)`."""
    for BK in (32, 64, 128):
        KG, steps = 8, BK // 4
        order = []
        sets = {0: list(range(0, KG))}
        for g in range(steps // KG):
            if g + 1 < steps // KG:
                sets[(g + 1) & 1] = list(range((g + 1) * KG, (g + 2) * KG))
            order += sets[g & 1]
        assert order == list(range(steps))


def test_long_k_slices_of_the_bottleneck_layer_cover_every_k_tile_once():
    """dcs_launch_gemm_bf16x3_longk (gemm_bf16x3.hip): K is cut into slices of whole 32-wide k tiles, about two workgroups per CU;
    slice z of the all-rows kernel walks the tiles [z * kts, min(nkt, (z + 1) * kts)).  Restated here: every tile belongs to
    exactly one slice, no slice is empty, the scratch holds one [M][n_cols] block per slice, and the reduce pass deals the
    slices to its four runs in order without gaps."""
    for n_cu in (256, 304, 64):
        for K in (16384, 18810, 166650, 166656, 666600, 32 * 1000 + 4):
            for n_cols in (128, 256, 512):
                col_wgs = n_cols // 128
                ksplit = (2 * n_cu + col_wgs - 1) // col_wgs
                nkt = (K + 31) // 32
                ksplit = min(ksplit, nkt)
                kts = (nkt + ksplit - 1) // ksplit
                ksplit = (nkt + kts - 1) // kts
                assert ksplit >= 2 and kts >= 1
                seen = np.zeros(nkt, dtype=int)
                for z in range(ksplit):
                    lo = z * kts                                   # kt_lo of the kernel (g.kchunk / 32 == kts)
                    hi = lo + kts if lo + kts < nkt else nkt
                    assert lo < hi <= nkt, (K, n_cols, z)
                    seen[lo:hi] += 1
                assert (seen == 1).all()
                per = (ksplit + 3) // 4                            # gemm_longk_reduce_kernel: run r adds slices [r * per, ...)
                order = []
                for run in range(4):
                    z0 = run * per
                    z1 = z0 + per if z0 + per < ksplit else ksplit
                    order += list(range(z0, max(z0, z1)))
                assert order == list(range(ksplit))


def test_conv1_planes_in_lds_give_every_fragment_the_floats_the_old_kernel_split():
    """conv1_mfma_kernel (round 5): a row's floats are split once into [plane][channel][chunk of 4] units of 8 bytes; lane (position
    fi, K piece kq) of column block b reads chunks (16 b + fi) + 2 kq and the next one = floats 4 (16 b + fi) + 8 kq .. + 7 -- the
    eight taps the old kernel read as two 16-byte LDS loads and split itself.  Every chunk a fragment touches lies inside the
    1 056 floats staged per channel, and the output tile may overwrite the planes (it is at least as large)."""
    kPos, kInW, kOutS = 256, 4 * 256 + 32, 256 + 4
    kQ = kInW // 4
    for b in range(kPos // 16):
        for fi in range(16):
            for kq in range(4):
                c0 = (16 * b + fi) + 2 * kq
                assert c0 + 1 < kQ
                floats = [4 * c0 + j for j in range(8)]
                assert floats == [4 * (16 * b + fi) + 8 * kq + j for j in range(8)]
    for C in (1, 4):
        planes_bytes = 3 * C * kInW * 2
        assert max(planes_bytes, 32 * kOutS * 4) == 32 * kOutS * 4         # the tile decides the LDS size


def test_weights_in_registers_conv2_kernel_visits_every_row_tap_pair_once_and_never_overwrites_a_ring_slot_in_use():
    """colconv_fwd_x3_kernel (the default conv2 of the Bach10 / score-informed graphs since round 6): KH = 20 taps, H = 30 input rows, HO = 11 output rows.  Step s multiplies input row s
    with every tap u whose output row y = s - u exists; the rows live in a ring of three chunks of two rows (slot = chunk % 3).
    Restated: every (y, u) pair is visited exactly once, in increasing u for a fixed y (the accumulation order of a chain); in
    interval i the chunk being written (i + 2, of the next block from i = 13 on) never shares a slot with the chunks being read
    (i: steps 2 i, 2 i + 1; i + 1: the prefetch for step 2 i + 2); the raw-register slot of a fetched chunk (chunk % 3) is the
    one its split reads three intervals later, also across the block boundary (15 chunks per block, 15 % 3 == 0)."""
    KH, H = 20, 30
    HO, NCH, RING = H - KH + 1, H // 2, 3
    seen = {}
    for i in range(NCH):
        for s in (2 * i, 2 * i + 1):
            u_lo, u_hi = max(0, s - (HO - 1)), min(s, KH - 1)
            ring_row = ((s // 2) % RING) * 2 + s % 2
            assert s // 2 == i and ring_row // 2 == i % RING            # the row of step s is in the chunk of interval i
            for u in range(u_lo, u_hi + 1):
                y = s - u
                assert 0 <= y < HO
                seen.setdefault(y, []).append(u)
            if s + 1 < H:                                              # prefetch of the next step's row
                nxt = ((s + 1) // 2) % RING
                assert nxt in (i % RING, (i + 1) % RING)
        written = (i + 2) % RING                                       # chunk i + 2 goes where chunk i - 1 was
        assert written not in (i % RING, (i + 1) % RING)
        fetched = (i + 2 + RING) % NCH if i + 2 + RING >= NCH else i + 2 + RING
        assert fetched % RING == (i + 2) % RING                        # the raw slot just consumed is the one refilled
    assert sorted(seen) == list(range(HO))
    for y, us in seen.items():
        assert us == list(range(KH)), y                                # every tap once, in order
    assert NCH % RING == 0
    # the packed weights: tap u = 2 k + parity sits at [parity][k] of dcs_decoder_x3_pack's array
    for u in range(KH):
        assert 2 * (u >> 1) + (u & 1) == u
