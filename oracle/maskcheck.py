"""Checker for soft-masked network outputs on EVERY bin (test infrastructure: helper of the GPU parity tests and of
bench.py's parity_check block; never on a timed or shipped path).

The reference's masks (separate_dsd.py:258-271, separate_bach10.py:251-264) are

    A:  m_i = (p_i + e) / (sum_j p_j + S e)        B:  m_i = p_i / (sum_j p_j + e),      e = 1e-18 * r

with p the rectified network outputs.  Where every p_j is (nearly) zero the mask is discontinuous: a float32
rounding difference that leaves 3e-9 instead of an exact 0 turns a mask of 0 (B) or 1/S (A) into a mask of 1.
No float32 implementation can meet an absolute 1e-4 there, so instead of exempting bins the check uses the bound
that follows from the mask itself.  With d = max |p_gpu - p_ref| (measured on the network output before masking) and
D the reference denominator,

    |m_i' - m_i| = |e_i D - n_i E| / (D (D + E))  <=  (S + 1) d / (D - S d)          (|e_i| <= d, |E| <= S d, n_i <= D)

so  |out_i' - out_i| <= mix * min(1, (S + 1) d / (D - S d))  (+ float32 rounding of the product).  Every bin must
satisfy that bound, the output must be a valid masked magnitude (0 <= out <= mix), and wherever the bound is below
the north-star tolerance the plain 1e-4 bar is asserted.  The number of bins whose error exceeds 1e-4 is counted and
reported (``gpurun_out/mask_bins.txt``).
"""
import os

import numpy as np

EPS_R = 1e-18 * 0.5


def check_masked(got, ref, p_ref, p_got, mix, S, conv, tol=1e-4, label=None, report="gpurun_out/mask_bins.txt", strict=True):
    """got, ref ``[S, n, tc, F]``; p_ref, p_got ``[n, >=S, tc, F]`` network outputs before masking; mix ``[n, tc, F]``
    (the mixture channel the masks multiply).  Returns the dict that is also appended to the report file.
    ``strict=False`` (bench.py): nothing is asserted, the three verdicts come back in the dict instead."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    mix = np.asarray(mix, dtype=np.float64)
    p_ref = np.asarray(p_ref, dtype=np.float64)[:, :S]
    p_got = np.asarray(p_got, dtype=np.float64)[:, :S]
    assert got.shape == ref.shape == (S,) + mix.shape, (got.shape, ref.shape, mix.shape)
    d = float(np.max(np.abs(p_got - p_ref))) if p_got is not p_ref else 0.0
    d = max(d, 1e-7)                                          # never tighter than float32 rounding of p itself
    D = p_ref.sum(axis=1) + (S * EPS_R if conv == 'A' else EPS_R)
    room = D - S * d
    lip = np.where(room > 0, (S + 1) * d / np.where(room > 0, room, 1.0), np.inf)
    bound = mix * np.minimum(1.0, lip) + 2e-6 * (1.0 + mix)
    err = np.abs(got - ref)                                   # [S, n, tc, F]
    worst = err.max(axis=0)
    # 1. every bin within the conditioning bound
    over = worst > bound
    valid = bool(np.all(got >= 0.0) and np.all(got <= mix[None] * (1 + 1e-5) + 1e-12))
    meaningful = bound <= tol
    if strict:
        assert not over.any(), "%s: %d bins exceed the mask-conditioning bound; worst excess %.3e" % (
            label, int(over.sum()), float((worst - bound).max()))
        # 2. valid masked magnitudes everywhere
        assert valid, label
        # 3. the plain north-star bar wherever the mask is conditioned well enough for it to be meaningful
        assert np.all(worst[meaningful] <= tol)
    outside = worst > tol
    rec = dict(label=label or "", bins=int(worst.size), network_output_max_err=d,
               within_conditioning_bound=bool(not over.any()), valid_magnitudes=valid,
               conditioned_bins_within_tol=bool(np.all(worst[meaningful] <= tol)),
               bins_outside_1e4=int(outside.sum()),
               outside_where_all_sources_below_1e5=int((outside & (p_ref.max(axis=1) < 1e-5)).sum()),
               max_err_where_conditioned=float(worst[meaningful].max()) if meaningful.any() else 0.0,
               conditioned_fraction=float(meaningful.mean()), max_err=float(worst.max()),
               zero_fraction_of_p=float((p_ref == 0).mean()))
    if report:
        os.makedirs(os.path.dirname(report), exist_ok=True)
        with open(report, "a") as fh:
            fh.write("%(label)s: %(bins)d bins, network output max|err| %(network_output_max_err).2e, p == 0 on "
                     "%(zero_fraction_of_p).3f; bins outside 1e-4: %(bins_outside_1e4)d (of which every source < 1e-5: "
                     "%(outside_where_all_sources_below_1e5)d); bound <= 1e-4 on %(conditioned_fraction).4f of the bins, "
                     "max err there %(max_err_where_conditioned).2e; max err overall %(max_err).2e\n" % rec)
    return rec
