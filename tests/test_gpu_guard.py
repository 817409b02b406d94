"""Guard-band harness (memory safety of the hand-indexed kernels; SURVEY section 5 "race detection / sanitizers").

Parity tests cannot see an out-of-bounds READ that lands in mapped memory, and the kernels use "load everything with a
clamped index, multiply by 0" tricks exactly where such reads would hide.  Here every device buffer handed to the C ABI
(inputs, weights, outputs) is carved out of a larger allocation with a >= 64 KiB red zone on either side, red zones AND
output payloads are filled with a poison byte before the call, and libdcs's own scratch blocks get the same treatment
(DCS_WS_GUARD / DCS_WS_POISON, dcs_debug_check_guards).  The main launch shapes of every entry point run in a child process,
once with poison 0xFF (every float32 / bf16 / f16 word a NaN, every int64 -1) and once with 0x4B (float32 1.3e7, finite):
  (i)   no red zone byte may change (out-of-bounds writes, ABI buffers and scratch);
  (ii)  every output must be finite (a NaN read from a red zone, from unwritten scratch or from an unwritten output element
        reaches it even when it is multiplied by 0);
  (iii) the outputs of the two runs must be BIT-IDENTICAL (a finite out-of-bounds or uninitialised read that influences the
        result shows up as a difference; run-to-run determinism is checked on the way).
No ASAN build: the MI355X boxes of this pool run with xnack off (HSA_XNACK unset, `rocminfo` reports xnack-), which
-fsanitize=address for gfx950 requires.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_CHILD = r'''
import hashlib, json, os, sys
ROOT, POISON, OUT = sys.argv[1], int(sys.argv[2]), sys.argv[3]
only = set(sys.argv[4].split(",")) if len(sys.argv) > 4 and sys.argv[4] else None
sys.path.insert(0, ROOT)
import numpy as np
import torch
import deepconvsep_amd as dcs
from deepconvsep_amd import runtime, score
from deepconvsep_amd.runtime import default_context
from deepconvsep_amd.synth import synth_audio, synth_params, synth_score_text

G = 1 << 16                                    # red zone bytes on either side of every ABI buffer


class Arena(object):
    """Guarded device allocations: [red zone | payload | red zone], all poisoned at birth."""
    def __init__(self):
        self.blocks = []

    def alloc(self, shape, dtype, device):
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
        n = 1
        for s in shape:
            n *= s
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        pay = (nbytes + 255) // 256 * 256
        raw = torch.full((pay + 2 * G,), POISON, dtype=torch.uint8, device=device)
        self.blocks.append((raw, nbytes))       # the padding up to the next 256 bytes counts as red zone
        return raw[G:G + nbytes].view(dtype).view(shape)

    def check(self):
        torch.cuda.synchronize()
        bad = 0
        for raw, nbytes in self.blocks:
            bad += int((raw[:G] != POISON).sum().item()) + int((raw[G + nbytes:] != POISON).sum().item())
        return bad


ARENA = Arena()


class _Host(object):
    def __init__(self, a):
        self.a = a

    def to(self, device):
        t = torch.from_numpy(self.a)
        out = ARENA.alloc(t.shape, t.dtype, device)
        out.copy_(t)
        return out


class GuardedTorch(object):
    """What runtime._torch() returns in this process: torch, with device allocations routed through the arena."""
    def __getattr__(self, k):
        return getattr(torch, k)

    def empty(self, shape, dtype=None, device=None):
        if device is None or torch.device(device).type != "cuda":
            return torch.empty(shape, dtype=dtype, device=device)
        return ARENA.alloc(shape, dtype or torch.float32, device)

    def zeros(self, shape, dtype=None, device=None):
        t = self.empty(shape, dtype=dtype, device=device)
        t.zero_()
        return t

    def from_numpy(self, a):
        return _Host(a)


_proxy = GuardedTorch()
runtime._torch = lambda: _proxy

ctx = default_context()
results = {}


def record(name, *outs):
    """outputs -> digest + finiteness; every red zone (ABI buffers here, scratch in libdcs) must be intact."""
    arrs = [np.ascontiguousarray(o if isinstance(o, np.ndarray) else ctx.to_host(o)) for o in outs]
    h = hashlib.sha256()
    finite = True
    for a in arrs:
        h.update(a.tobytes())
        if a.dtype.kind == "f":
            finite = finite and bool(np.isfinite(a).all())
    bad_ext = ARENA.check()
    try:
        n_ws = ctx.check_guards()
        bad_ws = ""
    except Exception as exc:
        n_ws, bad_ws = -1, str(exc)
    results[name] = {"sha256": h.hexdigest(), "finite": finite, "abi_red_zone_bytes_damaged": bad_ext,
                     "scratch_guard": bad_ws, "scratch_blocks": n_ws, "abi_blocks": len(ARENA.blocks)}


def want(name):
    return only is None or name in only


def tiles_samples(n, tc=30, ov=25, hop=512):
    return (tc + 1 + (n - 1) * (tc - ov) - 2) * hop


# ---- DSD fused path, both kernel families ---------------------------------------------------------------------------
N, F = 2048, 1025
P_DSD = synth_params("dsd", 30, F, seed=2)
sep = dcs.Separator("dsd", P_DSD, 0.3, 30, 25, 32, F, N, 512, np.hanning)
if want("dsd_1"):
    record("dsd_1", sep.separate(synth_audio(tiles_samples(1) + 77, seed=1)))           # one tile, ragged tail
if want("dsd_32_onebatch"):
    record("dsd_32_onebatch", sep.separate(synth_audio(tiles_samples(32), seed=2)))     # one-batch kernels (dsd_lat.hip)
if want("dsd_32_throughput"):
    sep.net.set_latency_stages(0)
    record("dsd_32_throughput", sep.separate(synth_audio(tiles_samples(32) + 311, seed=3)))
    sep.net.set_latency_stages(-1)
if want("dsd_pcm16_group"):
    # the batch driver's int16 path (round 5): int16 frames up, device mix-down, shared launches, device int16 conversion
    frames16 = [(synth_audio(tiles_samples(8) + 100 * i, seed=20 + i, channels=1 + (i & 1)) * 32767).astype(np.int16) for i in range(3)]
    record("dsd_pcm16_group", *[o.astype(np.float32) for o in sep.separate_many_pcm16(frames16, ring=0)])
if want("dsd_640_batch"):
    clips = np.stack([synth_audio(tiles_samples(32), seed=10 + c) for c in range(20)])
    a = ctx.to_device(clips, np.float32)
    record("dsd_640_batch", sep.net.separate_batch(sep.plan, a, 25, sep.tiler, 0.3))     # the driver's launch shape
if want("dsd_ragged"):
    lens = [tiles_samples(32) + 5, tiles_samples(24) + 400, tiles_samples(27), tiles_samples(32) - 1000]
    outs = sep.separate_many([synth_audio(L, seed=30 + i) for i, L in enumerate(lens)])
    record("dsd_ragged", *outs)
if want("dsd_4096"):
    record("dsd_4096", sep.separate(synth_audio(tiles_samples(4096) + 13, seed=4)))      # 3 min 58 s, one launch set
if want("dsd_spectra_stepwise"):
    a = ctx.to_device(synth_audio(3 * 44100 + 5, seed=5), np.float32)
    s_d, m_d, p_d = sep.net.separate_spectra(sep.plan, a, 25, sep.tiler, 0.3)
    record("dsd_spectra_stepwise", s_d, m_d, p_d, sep.separate_stepwise(synth_audio(2 * 44100, seed=6)))
del sep
if want("dsd_n1024"):
    s1 = dcs.Separator("dsd", synth_params("dsd", 30, 513, seed=2), 0.3, 30, 25, 32, 513, 1024, 512, np.hanning)
    record("dsd_n1024", s1.separate(synth_audio(5 * 44100 + 3, seed=7)), *s1.separate_many(
        [synth_audio(44100 * 2 + 17 * i, seed=40 + i) for i in range(3)]))
    del s1

# ---- transform API (float64 and float32), tiler, cross-fade, int16 ----------------------------------------------------
if want("transform"):
    for N2, hop, dt in ((1024, 512, np.float64), (2048, 512, np.float32), (4096, 512, np.float32), (1024, 256, np.float64), (512, 200, np.float32)):
        plan = runtime.StftPlan(ctx, N2, hop, np.hanning(N2))
        a = ctx.to_device(synth_audio(44100 + 123, seed=8), dt)
        mag, ph = plan.forward(a, phase=True)
        back = plan.inverse(mag, ph, n_out=int(a.numel()))
        tl, n = runtime.tile(ctx, mag.to(torch.float32) if dt == np.float64 else mag, 30, 25, dcs.runtime.TILER_SCRIPT, 0.3)
        record("transform_%d_%d_%s" % (N2, hop, np.dtype(dt).name), mag, ph, back, tl)
    pcm = ctx.to_device(np.linspace(-1.2, 1.2, 100003), np.float32)
    record("pcm_int16", runtime.pcm_to_int16(ctx, pcm))
if want("operators"):
    net = runtime.Network(ctx, "dsd", synth_params("dsd", 30, 513, seed=2), 30, 513)
    rs = np.random.RandomState(3)
    x = ctx.to_device(np.abs(rs.randn(37, 1, 30, 513)).astype(np.float32), np.float32)
    out = net.forward_masked(x)
    record("operators", out, net.forward_raw(x), runtime.overlap_add(ctx, out, 25))
    del net

# ---- generic graphs -----------------------------------------------------------------------------------------------
if want("ikala"):
    for nm, seed in (("ikala", 1), ("ikala_nopool", 6)):
        sp = dcs.Separator("ikala", synth_params(nm, 30, 1025, seed=seed), 0.3, 30, 20, 32, 1025, 2048, 512, np.hanning)
        st = synth_audio(5 * 44100 + 9, seed=9, channels=2)
        record(nm, sp.separate(st[:, 0] + st[:, 1]), *sp.separate_many([synth_audio(44100 * 2 + 100 * i, seed=50 + i) for i in range(3)]))
        del sp
if want("bach10"):
    for f16 in (False, True):
        sp = dcs.Separator("bach10", synth_params("bach10", 30, 2049, seed=3), 0.3, 30, 25, 32, 2049, 4096, 512, dcs.blackmanharris)
        if f16:
            sp.net.set_conv_precision("f16")
        record("bach10_f16" if f16 else "bach10_f32", sp.separate(synth_audio(6 * 44100 + 1, seed=11)))    # 100 tiles: >= 128-row GEMM shapes need 10 s
        del sp
if want("bach10_10s"):
    sp = dcs.Separator("bach10", synth_params("bach10", 30, 2049, seed=3), 0.3, 30, 25, 32, 2049, 4096, 512, dcs.blackmanharris)
    sp.net.set_conv_precision("f16")
    record("bach10_10s_f16", sp.separate(synth_audio(441000, seed=0)))
    del sp
if want("score_informed"):
    import tempfile
    tmp = tempfile.mkdtemp()
    Lc = (25 + 1 + 127 * 5 - 2) * 512
    files = []
    for i in range(4):
        files.append("inst%d.txt" % i)
        with open(os.path.join(tmp, files[-1]), "w") as fh:
            fh.write(synth_score_text(40 + i, Lc / 44100.0 + 0.5, 40 + 5 * i, 64 + 6 * i))
    nframes = int(np.ceil(Lc / 512.0)) + 2
    mel = score.melody_table(files, tmp, nframes, 44100, 512, 4096)
    sp = dcs.Separator("bach10_si", synth_params("bach10_si", 30, 2049, seed=5), 0.3, 30, 25, 128, 2049, 4096, 512,
                       dcs.blackmanharris, tiler="library")
    audio = synth_audio(Lc, seed=4)
    a = ctx.to_device(audio, np.float32)
    mag, _ = sp.plan.forward(a, phase=False)
    inp, mask = score.score_masks(ctx, mag, mel, 0, int(mag.shape[0]), want_input=True, want_mask=True)
    record("score_informed", sp.separate_scoreinformed(audio, mel), inp, mask)
    del sp
if want("stereo"):
    sp = dcs.Separator("dsd_ild", synth_params("dsd_ild", 30, 513, seed=7), 0.3, 30, 25, 32, 513, 1024, 512, np.hanning, tiler="library")
    record("stereo", sp.separate_stereo(synth_audio(4 * 44100 + 21, seed=12, channels=2)))
    del sp

# ---- the harness sees what it claims to see: one byte written just past a payload is counted
probe = ARENA.alloc((1000,), torch.float32, ctx.device)
before = ARENA.check()
ARENA.blocks[-1][0][G + 4000 + 24] = (POISON ^ 0x55)                                   # 24 bytes past the 4000-byte payload
results["_selftest"] = {"before": before, "after_one_stray_byte": ARENA.check()}

json.dump(results, open(OUT, "w"), indent=1)
'''


def _start(poison, out, only=""):
    env = dict(os.environ)
    env.update({"DCS_WS_GUARD": "65536", "DCS_WS_POISON": str(poison)})
    return subprocess.Popen([sys.executable, "-c", _CHILD, ROOT, str(poison), out, only], env=env, stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True)


def _finish(proc, poison, out):
    so, se = proc.communicate(timeout=900)
    assert proc.returncode == 0, (poison, so[-800:], se[-2500:])
    with open(out) as fh:
        return json.load(fh)


def test_red_zones_untouched_outputs_finite_and_independent_of_the_poison(tmp_path):
    only = os.environ.get("DCS_GUARD_CASES", "")
    # the two poison runs are independent processes: side by side on the one GPU (each is ~10 s of start-up and small launches)
    f_nan, f_big = str(tmp_path / "nan.json"), str(tmp_path / "big.json")
    p_nan, p_big = _start(0xFF, f_nan, only), _start(0x4B, f_big, only)
    nan_run = _finish(p_nan, 0xFF, f_nan)
    big_run = _finish(p_big, 0x4B, f_big)
    for run in (nan_run, big_run):
        st = run.pop("_selftest")
        assert st["before"] == 0 and st["after_one_stray_byte"] == 1, st        # a stray byte in a red zone IS seen
    assert sorted(nan_run) == sorted(big_run) and len(nan_run) >= (1 if only else 20)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "guard_harness.json"), "w") as fh:
        json.dump({"poison_0xFF": nan_run, "poison_0x4B": big_run}, fh, indent=1)
    problems = []
    for name in sorted(nan_run):
        for tag, run in (("0xFF", nan_run), ("0x4B", big_run)):
            rec = run[name]
            if rec["abi_red_zone_bytes_damaged"]:
                problems.append("%s [%s]: %d red-zone bytes of the ABI buffers changed" % (name, tag, rec["abi_red_zone_bytes_damaged"]))
            if rec["scratch_guard"]:
                problems.append("%s [%s]: %s" % (name, tag, rec["scratch_guard"]))
            if not rec["finite"]:
                problems.append("%s [%s]: non-finite output" % (name, tag))
            if rec["scratch_blocks"] < 0:
                problems.append("%s [%s]: dcs_debug_check_guards failed" % (name, tag))
        if nan_run[name]["sha256"] != big_run[name]["sha256"]:
            problems.append("%s: outputs differ between the two poison values" % name)
    assert not problems, "\n".join(problems)
    # the switch was honoured: the cases that hold a model see its guarded scratch blocks (the transform API has none)
    if not only:
        for name in ("dsd_640_batch", "dsd_4096", "ikala", "bach10_f16", "score_informed", "stereo"):
            assert nan_run[name]["scratch_blocks"] >= 1 and nan_run[name]["abi_blocks"] >= 10, (name, nan_run[name])
