"""CPU restatement of the ownership arithmetic of istft_chain_kernel (csrc/fft_wave.hip): which wave transforms which frame,
which wave stores which hop-block and from which accumulators.  For every (blocks, frames, R, LC) the emulation must store each
block of the signal exactly once, and the frames that were added into what is stored must be exactly the frames that reach the
block (n - R < frame <= n, frame < T) -- the property the device kernel's parity tests then confirm numerically."""
import numpy as np
import pytest

NW = 8


def plan(n_blocks, n_src, R, resident_wg, Cs):
    """launch_inv's choice: the fewest frames per wave (LC >= R - 1) for which every workgroup is resident at once; None when the
    chain is not shorter than the barrier-free kernel's Cs + R - 1."""
    LC = max(R - 1, 1)
    while True:
        Sb = NW * LC - (R - 1)
        G = 1 if n_blocks <= R - 1 else -(-(n_blocks - (R - 1)) // Sb)
        if G * n_src <= resident_wg or LC >= Cs + R - 1:
            break
        LC += 1
    return (LC, G) if LC < Cs + R - 1 else None


def emulate(n_blocks, T, R, LC, G):
    """Returns {block: sorted list of frames added into the stored value}; asserts single ownership."""
    stored = {}
    Sb = NW * LC - (R - 1)
    for jg in range(G):
        wg0 = jg * Sb
        blk0 = 0 if jg == 0 else wg0 + (R - 1)
        if blk0 >= n_blocks:
            continue
        heads, tails = {}, {}
        for w in range(NW):
            f0, f1 = wg0 + w * LC, wg0 + (w + 1) * LC
            acc = {q: [] for q in range(R)}                     # register slot -> frames added since it was last cleared
            nb = (f0 // R) * R
            while nb < f1:
                for j in range(R):
                    n = nb + j
                    if n < f0 or n >= f1:
                        continue
                    if n <= T - 1:
                        for d in range(R):
                            acc[(j + d) % R].append(n)
                    g = n
                    head = g < f0 + (R - 1) and f0 > 0
                    if not head:
                        if g < n_blocks:
                            assert g not in stored, ("stored twice", g)
                            stored[g] = list(acc[j])
                    elif w > 0:
                        heads[(w, g - f0)] = list(acc[j])
                    acc[j] = []
                nb += R
            tails[w] = {d: list(acc[(f1 + d) % R]) for d in range(R - 1)}
        for w in range(NW - 1):                                  # after the barrier: my tail + the right wave's head
            f1 = wg0 + (w + 1) * LC
            for d in range(R - 1):
                g = f1 + d
                if g >= n_blocks:
                    break
                assert g not in stored, ("stored twice", g)
                stored[g] = tails[w][d] + heads[(w + 1, d)]
    return stored


@pytest.mark.parametrize("R", [2, 4])
def test_every_block_is_stored_once_with_exactly_its_frames(R):
    rs = np.random.RandomState(R)
    cases = [(185, 183, 8, 3), (185, 183, 12, 2), (186, 184, 8, 4), (863, 862, 2, 60), (20, 18, 3, 1), (9, 7, 3, 1),
             (3, 1, 3, 1), (64, 62, 8, 1)]
    for _ in range(200):
        LC = int(rs.randint(max(R - 1, 1), 14))
        G = int(rs.randint(1, 6))
        Sb = NW * LC - (R - 1)
        n_blocks = int(rs.randint(max(1, (G - 1) * Sb + R), G * Sb + R))       # G workgroups are needed and suffice
        T = n_blocks - int(rs.randint(0, R + 2))                                 # frames: blocks past T - 1 + R - 1 stay zero
        cases.append((n_blocks, max(T, 1), LC, G))
    for n_blocks, T, LC, G in cases:
        if LC < R - 1:
            continue
        Sb = NW * LC - (R - 1)
        assert G * Sb + (R - 1) >= n_blocks, "the case's G does not cover the signal"
        stored = emulate(n_blocks, T, R, LC, G)
        assert sorted(stored) == list(range(n_blocks)), (n_blocks, T, LC, G)
        for g, frames in stored.items():
            want = [f for f in range(max(0, g - R + 1), g + 1) if f <= T - 1]
            assert frames == want, (g, frames, want)        # also in increasing order: tail frames, then head frames


def test_the_launch_plan_shortens_the_chain_at_launch_group_sizes():
    # 20 clips x 4 sources of 185 hop-blocks (bench.py --steps 20), N = 2048 hop 512: 8 frames per wave instead of 8 + 3
    assert plan(185, 80, 4, 256, Cs=8) == (8, 3)
    # 32 clips: 12 instead of 12 + 3
    assert plan(185, 128, 4, 256, Cs=12) == (12, 2)
    # one long clip whose barrier-free chain is already dominated by its own blocks: no gain claimed, kernel not taken
    lc = plan(20505, 4, 4, 256, Cs=41)
    assert lc is None or lc[0] < 44
    # the plan always covers the signal
    for n_blocks, n_src, R, Cs in [(185, 80, 4, 8), (863, 4, 2, 2), (50, 400, 4, 10), (5, 4, 4, 1)]:
        got = plan(n_blocks, n_src, R, 256, Cs)
        if got:
            LC, G = got
            assert G * (NW * LC - (R - 1)) + (R - 1) >= n_blocks and LC >= R - 1
