"""Randomised network draws on the GPU (tests/test_oracle_net.py checks the same 24 draws of oracle/cases.random_draw
on the CPU against the reference's own build_ca source executed on the Lasagne stand-in): the HIP networks against
oracle.net_ref -- network output on every bin within 1e-4, masked sources through the all-bin mask check, both max-pool
tie routings of the iKala graph."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from deepconvsep_amd.arch import ARCHS, EPS_A, EPS_B, TIE_ALL, TIE_FIRST  # noqa: E402
from deepconvsep_amd.runtime import Network, default_context  # noqa: E402
from oracle import cases, net_ref  # noqa: E402
from maskcheck import check_masked  # noqa: E402


@pytest.mark.parametrize("seed", range(24))
def test_random_draws_hip_network_matches_oracle(seed):
    arch, F, params, x = cases.random_draw(seed)
    tc, n, S = 30, x.shape[0], ARCHS[arch].S
    ctx = default_context()
    net = Network(ctx, arch, params, tc, F)
    xd = ctx.to_device(x, np.float32)
    x64 = x.astype(np.float64)
    ties = (("all", TIE_ALL), ("first", TIE_FIRST)) if arch == "ikala" else (("all", TIE_ALL),)
    for tname, tmode in ties:
        want = net_ref.forward(arch, params, x64, tie_mode=tname, inverse='explicit').numpy()
        p = net.forward_raw(xd, tie_mode=tmode).cpu().numpy()
        assert p.shape == want.shape and np.isfinite(p).all()
        assert np.max(np.abs(p - want)) < 1e-4, (seed, arch, tname)
        conv = 'A' if arch in ("dsd", "ikala") else 'B'
        got = net.forward_masked(xd, eps_mode=EPS_A if conv == 'A' else EPS_B, tie_mode=tmode).cpu().numpy()
        ref = net_ref.predict(arch, params, x64, tie_mode=tname, inverse='explicit')
        check_masked(got, np.stack([r[:, 0] for r in ref]), want, p, x64[:, 0], S, conv,
                     label="random draw %d: %s F=%d, %d tiles, ties %s" % (seed, arch, F, n, tname))
