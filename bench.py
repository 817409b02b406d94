#!/usr/bin/env python3
"""Benchmark of the MI355X separation hot path (BASELINE.json metric: spectrogram-frames/s,
DSD100 4-source model, frameSize=2048 hop=512).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One *step* = one pass of the whole hot path (STFT -> x0.3 -> tiles -> conv encoder/decoder ->
soft mask -> cross-fade overlap-add -> /0.3 -> iSTFT -> truncate) over one batch of 32 tiles
(BASELINE.json configs[1]: "DSD100 4-source separate_dsd.py, batch=32 tiles, fp32, 1xMI355X"),
i.e. 2.14 s of synthetic 44.1 kHz audio already resident in HBM, producing 4 PCM signals in
HBM.  With N GPUs every rank separates its own 32-tile batches (weak scaling, configs[2]) and
the PCM of all ranks is all-gathered over RCCL inside the timed region.

A 32-tile batch holds 0.57 GFLOP -- 3.6 us of the chip's f32 peak -- spread over dependent
kernels, so one batch at a time leaves the GPU mostly idle (`single_stream` reports that regime:
one batch per launch, one stream).  Steps are independent, therefore the K steps of a round are
cut into launch groups of about `--clips-per-launch` (32) batches that share one set of kernel
launches (dcs_separate_batch: every batch is tiled, cross-faded and inverted exactly as if it
were alone) and `--streams S` (default 3) such groups are in flight on S HIP streams, each with
its own libdcs context, plan, model handle and buffers -- the way a batch-of-files server overlaps
requests.  K steps are always K batches; the groups of a round differ in size by at most one
batch, so the launch shape does not depend on how K divides by the group size.  (Group size, 3 streams, MI355X,
profiles/r02_g_group_size_sweep.txt: 14 -> 20.2 M frames/s, 16 -> 20.7, 24 -> 21.9, 32 -> 23.2, 48 -> 23.8, 64 -> 24.6;
32 batches = 1024 tiles = a 68 s clip's worth per launch.)

Timing: a *round* is EXACTLY K steps bracketed by a barrier + torch.cuda.synchronize() on both
sides (max over ranks).  Rounds are repeated until `--min-time` (0.25 s) has been timed; `value`
comes from the MEDIAN round, so it does not depend on K being large (`rounds`, `round_ms` in the
line).  Every 4th round brackets the launches of the dominant kernel with HIP events; such a round
issues its K steps on stream 0 alone, because
with kernels of other streams sharing the CUs an event pair measures a time-sliced duration.

Rank 0 prints ONE compact JSON line (< 4 KB, strict JSON) as the LAST line of stdout: the contract keys, `roofline`
of the dominant kernel of the headline workload (transposed conv1 + bias + rectify + soft mask + cross-fade),
`cpu_baseline` (the CPU oracle -- reference-equivalent NumPy + torch-CPU float64 path -- timed on this host) and
`parity_check`.  The full result object is written to bench_detail.json next to this file (and with --detail-stdout
printed on an earlier stdout line): `legs` holds the other BASELINE configs (iKala 10 s, Bach10 with the f16 MFMA conv
path, score-informed batch of 128), each with its own `roofline` and `cpu_baseline`; `saturating` repeats the headline
measurement on a long clip (4096 tiles, 3 min 58 s) where one launch fills the chip (DESIGN.md "measurement").
"""
import argparse
import json
import os
import statistics
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HOP, TC, OV, SCALE, SR = 512, 30, 25, 0.3, 44100
PEAK_F32_TFLOPS = 157.3    # /opt/skills/guides/MI355X_MICROARCH.md: f32 MFMA = f32 vector peak
PEAK_F16_TFLOPS = 2500.0   # dense f16 / bf16 MFMA
FINAL_NAMES = {"one_batch": "lat_final_kernel (deconv1+bias+relu+mask+crossfade of ONE batch per call: 16 frames x 64 bins per "
                            "workgroup, every covering tile staged in LDS, bf16 MFMA on three-way split operands)",
               "f32x64": "final_kernel<fold, 64-bin workgroups> (deconv1+bias+relu+mask+crossfade, f32 MFMA)",
               "f32x128": "final_kernel<fold> (deconv1+bias+relu+mask+crossfade, f32 MFMA)",
               "bf16x3": "final_bf16x3_kernel (deconv1+bias+relu+mask+crossfade; bf16 MFMA on operands split into "
                         "three bf16 terms, six products kept, f32 accumulation: f32-class results)"}
PEAK_HBM_GBPS = 8000.0
TRAFFIC_FILE = os.path.join("profiles", "r06_traffic.json")
# which kernel carries a timing tag in each leg (substring of the kernel name in the rocprofv3 counter files); used to look
# up the HBM traffic record of a leg's kernels in TRAFFIC_FILE["legs"][leg]
LEG_KERNELS = {
    # (substring, 'max' | 'min'): conv2 and its transpose are the same kernel at two grids -- the transpose (2 branches) moves more
    # (84 tiles: both dense layers run on gemm_rows_kernel -- the bottleneck K-split over 424 workgroups, the per-source layer 1890)
    "ikala": {"conv1": "conv1_reg_kernel", "conv2": ("slabconv_ps_kernel", "min"), "deconv2": "slabconv_ps_kernel",
              "fc": "gemm_rows_kernel@grid_threads=108544", "fc1x": "gemm_rows_kernel@grid_threads=483840", "final": "deconv1_reg_kernel"},
    # (f16 switch, round 6: the dense layers on f16 weight planes, gemm_f16.hip -- the bottleneck layer with K cut into slices, the
    # per-source layers writing D as f16 channels-last for the fused decoder)
    "bach10_f16": {"conv1": "conv1_mfma_kernel", "conv2": "colconv_wreg_scatter_kernel", "decoder": "colconv_deconv1_fused_kernel",
                   "fc": "gemm_f16_longk_kernel", "fc1x": "gemm_f16_skinny_kernel"},
    # (f32-class graphs: the bottleneck layer is the all-rows bf16 x 3 kernel with K cut into 256 slices: 2 x 248 workgroups; the
    # per-source layers are the same kernel at its widest grid; conv2 = colconv_fwd_x3_kernel since round 6)
    "score_informed": {"conv1": "conv1_mfma_kernel", "conv2": "colconv_fwd_x3_kernel", "deconv2": "colconv_kernel",
                       "fc": "gemm_bf16x3_skinny_kernel@grid_threads=126976", "fc1x": "gemm_bf16x3_skinny_kernel",
                       "final": "deconv1_mfma_kernel", "decoder": "colconv_deconv1_fused_x3_kernel"},
    "bach10_f32": {"conv1": "conv1_mfma_kernel", "conv2": "colconv_fwd_x3_kernel", "deconv2": "colconv_kernel",
                   "fc": "gemm_bf16x3_skinny_kernel@grid_threads=126976", "fc1x": "gemm_bf16x3_skinny_kernel",
                   "final": "deconv1_mfma_kernel", "decoder": "colconv_deconv1_fused_x3_kernel"},
}
# kernels that execute on the 16-bit matrix pipe: (products issued per f32 product, K padding factor)
LEG_ISSUED = {
    "ikala": {"conv2": (6, 32.0 / 30.0), "deconv2": (6, 32.0 / 30.0)},   # 84 tiles: the dense layers stay on the f32 MFMA (M < 128)
    # (f16 dense layers: the rows as two f16 terms x one f16 weight plane = 2 products; 32 output channels per position for 30)
    "bach10_f16": {"conv1": (6, 32.0 / 30.0), "conv2": (1, 32.0 / 30.0), "fc": (2, 1.0), "fc1x": (2, 32.0 / 30.0)},
    "score_informed": {"conv1": (6, 32.0 / 30.0), "conv2": (6, 32.0 / 30.0), "final": (6, 32.0 / 30.0), "fc": (6, 1.0), "fc1x": (6, 1.0),
                       "decoder": (6, 32.0 / 30.0 * 120.0 / 110.0)},
    "bach10_f32": {"conv1": (6, 32.0 / 30.0), "conv2": (6, 32.0 / 30.0), "final": (6, 32.0 / 30.0), "fc": (6, 1.0), "fc1x": (6, 1.0),
                   "decoder": (6, 32.0 / 30.0 * 120.0 / 110.0)},     # + the (row, tap) slots that meet a zero row (colconv_x3.hip)
}


def load_traffic():
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as fh:
            return json.load(fh)
    except Exception:
        return {}


def traffic_bytes(rec):
    """2*FETCH + WRITE (gfx950: FETCH_SIZE counts half the bytes of wide coalesced reads, MI355X_MICROARCH.md)."""
    if not rec or "FETCH_SIZE_KiB" not in rec or "WRITE_SIZE_KiB" not in rec:
        return None
    return int((2 * rec["FETCH_SIZE_KiB"] + rec["WRITE_SIZE_KiB"]) * 1024)


def samples_for_tiles(n_tiles, tc=TC, ov=OV, hop=HOP, library=False):
    """Shortest signal (whole hops) whose tiling yields n_tiles (script tiler: start + tc < T,
    separate_dsd.py:123; library tiler: start + ov < T, util.py:230)."""
    guard = ov if library else tc
    frames = guard + 1 + (n_tiles - 1) * (tc - ov)
    return (frames - 2) * hop


def split_groups(k, per_group, lanes):
    """K steps -> launch groups of about per_group steps, sizes differing by <= 1.  A short round (K = 20) is ONE
    group on one stream: measured 13.1 us per step against 15.1 us for two groups of 10 on two streams."""
    g = max(1, int(k / float(per_group) + 0.5))
    g = min(g, k)
    base, extra = divmod(k, g)
    return [base + 1] * extra + [base] * (g - extra)


def plan_round(k, per_group, lanes, gathering, gather_mode="root", no_pipeline=False):
    """(launch groups of a round, pipelined?).  N = 1: split_groups.  N > 1 (gathering) and a round of ONE launch group: two
    half-groups on two lanes, the second's kernels ordered behind the first's by an event, so that the first's collective rides
    under the second's compute -- unless there is no collective (`none`), one lane, one step, or
    --no-pipeline."""
    groups = split_groups(k, per_group, lanes)
    pipelined = bool(gathering and len(groups) == 1 and k >= 2 and lanes >= 2 and gather_mode != "none" and not no_pipeline)
    if pipelined:
        groups = [k - k // 2, k // 2]
    return groups, pipelined


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def time_cpu(fn, min_seconds, max_reps=1000):
    reps, t0 = 0, time.perf_counter()
    while True:
        fn()
        reps += 1
        el = time.perf_counter() - t0
        if el >= min_seconds or reps >= max_reps:
            return el / reps, reps, el


HEADLINE_LIMIT = 4096     # bytes: the driver keeps a bounded tail of stdout and parses the LAST line of it
DETAIL_FILE = "bench_detail.json"


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d} if isinstance(d, dict) else None


def _finite(x):
    """Strict JSON has no NaN / Infinity: such numbers become null."""
    if isinstance(x, float):
        return x if (x == x and x not in (float("inf"), float("-inf"))) else None
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    return x


def headline(line):
    """The compact last stdout line (< HEADLINE_LIMIT bytes, strict JSON) cut out of the full result `line`: the
    contract keys, the dominant kernel's roofline, the CPU baseline, the parity verdict and -- for N > 1 -- the
    gather split.  Everything else (single_stream, launch_group, saturating, host_fed, legs, cli) goes to
    DETAIL_FILE / the earlier `--detail-stdout` line."""
    keep = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "x_realtime", "whole_path_algorithmic_tflops", "whole_path_frac_of_f32_peak",
            "single_stream_ms_per_step", "single_stream_value", "rounds", "timed_region_s", "gather_check"]
    h = {k: line[k] for k in keep if k in line}
    cfg = dict(line.get("config") or {})
    if len(str(cfg.get("workload", ""))) > 420:
        cfg["workload"] = str(cfg["workload"])[:417] + "..."
    h["config"] = cfg
    r = line.get("roofline")
    if r:
        rr = _pick(r, ["bound", "kernel", "variant", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio",
                       "traffic_key_missing", "algorithmic_bytes", "algorithmic_flops", "avg_kernel_ms", "launches", "tiles_per_launch"])
        if len(str(rr.get("kernel", ""))) > 120:
            rr["kernel"] = str(rr["kernel"])[:117] + "..."
        if isinstance(r.get("issued"), dict):
            rr["issued"] = _pick(r["issued"], ["achieved", "peak", "unit", "frac"])
        h["roofline"] = rr
    c = line.get("cpu_baseline")
    if c:
        cc = _pick(c, ["value", "unit", "cores", "kind", "cpu_model", "host_cpu_count", "label", "sample"])
        if len(str(cc.get("sample", ""))) > 330:
            cc["sample"] = str(cc["sample"])[:327] + "..."
        h["cpu_baseline"] = cc
    else:
        h["cpu_baseline"] = None
    pc = line.get("parity_check")
    h["parity_check"] = _pick(pc, ["ok", "max_abs_pcm_err", "tolerance", "network_output_max_err", "mask_criterion_ok", "mask_bins",
                                   "masked_bins_outside_1e4", "conditioned_fraction", "ranks"]) if pc else None
    hs = line.get("hbm_stages")
    if isinstance(hs, dict):        # {"stft": {frac, GBps, traffic_ratio}, "istft": {...}}: < 200 bytes
        h["hbm_stages"] = {k: _pick(v, ["frac", "GBps", "traffic_ratio"]) for k, v in hs.items()}
    g = line.get("gather")
    if g:
        gg = _pick(g, ["mode", "impl", "payload_bytes_per_rank_per_group", "pipelined_half_groups", "round_ms_without_gather",
                       "round_ms_with_gather", "ms_per_group_collective_alone"])
        gg["impl"] = str(gg.get("impl", ""))[:60]
        h["gather"] = gg
    legs = line.get("legs")
    if isinstance(legs, dict):      # one number per extra BASELINE config; the blocks themselves are in the detail
        h["legs_ms_per_clip"] = {k: (v.get("ms_per_clip") if isinstance(v, dict) else None) for k, v in legs.items()}
    h["detail"] = line.get("detail_file", DETAIL_FILE)
    h = _finite(h)
    text = json.dumps(h, allow_nan=False)
    # belt and braces: drop the optional blocks, longest first, until the line fits
    for k in ("legs_ms_per_clip", "gather_check", "whole_path_algorithmic_tflops", "x_realtime"):
        if len(text) < HEADLINE_LIMIT:
            break
        h.pop(k, None)
        text = json.dumps(h, allow_nan=False)
    if len(text) >= HEADLINE_LIMIT:
        h["config"] = {"workload": str(cfg.get("workload", ""))[:200]}
        if isinstance(h.get("cpu_baseline"), dict):
            h["cpu_baseline"]["sample"] = str(h["cpu_baseline"].get("sample", ""))[:120]
        text = json.dumps(h, allow_nan=False)
    return text


def emit(line, detail_stdout=False, out=None):
    """Write the full result to DETAIL_FILE (next to bench.py; best effort), optionally print it on an EARLIER stdout
    line, then print the compact headline as the last stdout line."""
    out = out or sys.stdout
    full = json.dumps(_finite(line), allow_nan=False)
    try:                                                # (temp file + rename: bench processes side by side never interleave)
        tmp = os.path.join(ROOT, "%s.%d.tmp" % (DETAIL_FILE, os.getpid()))
        with open(tmp, "w") as fh:
            fh.write(full + "\n")
        os.replace(tmp, os.path.join(ROOT, DETAIL_FILE))
    except OSError:
        pass
    if detail_stdout:
        out.write(full + "\n")
    out.write(headline(line) + "\n")
    out.flush()


def mask_bin_report(net, arch_name, params, tiles, conv):
    """North-star "within 1e-4 per mask bin", measured on one batch of tiles (outside every timed region): the HIP network
    output and masked sources of `tiles [n, C, tc, F]` against the float64 oracle, through the every-bin checker of the GPU
    tests (oracle/maskcheck.py).  `masked_bins_outside_1e4` counts the bins whose masked magnitude differs by more than 1e-4;
    where every source is ~0 the reference's mask is discontinuous (INTEGRATION.md "mask bins"), so the checker also reports
    the share of bins whose conditioning bound is below 1e-4 (`conditioned_fraction`) and the error there."""
    from oracle import net_ref
    from oracle.maskcheck import check_masked
    S = net.S
    x64 = np.asarray(tiles, dtype=np.float64)
    xd = net.ctx.to_device(np.asarray(tiles, dtype=np.float32), np.float32)
    p_got = net.forward_raw(xd).cpu().numpy()
    got = net.forward_masked(xd).cpu().numpy()
    p_ref = net_ref.forward(arch_name, params, x64, inverse='explicit').numpy()
    ref = np.stack([r[:, 0] for r in net_ref.predict(arch_name, params, x64, inverse='explicit')])
    rec = check_masked(got, ref, p_ref, p_got, x64[:, 0], S, conv, report=None, strict=False)
    return {"tiles": int(x64.shape[0]), "mask_bins": rec["bins"], "masked_bins_outside_1e4": rec["bins_outside_1e4"],
            "outside_where_all_sources_below_1e5": rec["outside_where_all_sources_below_1e5"],
            "conditioned_fraction": round(rec["conditioned_fraction"], 6),
            "max_err_where_conditioned": rec["max_err_where_conditioned"], "max_err": rec["max_err"],
            "network_output_max_err": float(np.max(np.abs(p_got[:, :S] - p_ref[:, :S]))),
            "within_conditioning_bound": rec["within_conditioning_bound"], "valid_magnitudes": rec["valid_magnitudes"],
            "conditioned_bins_within_tol": rec["conditioned_bins_within_tol"],
            "mask_consistent": rec["mask_consistent"], "mask_consistency_max_err": rec["mask_consistency_max_err"],
            "unconditioned_bins": rec["unconditioned_bins"],
            # the four-part criterion of oracle/maskcheck.py: what `ok` requires of the masks
            "criterion_ok": bool(rec["within_conditioning_bound"] and rec["valid_magnitudes"] and
                                 rec["conditioned_bins_within_tol"] and rec["mask_consistent"]),
            "zero_fraction_of_network_output": round(rec["zero_fraction_of_p"], 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=384)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--tiles", type=int, default=32, help="tiles per GPU per step (BASELINE configs[1]: 32)")
    ap.add_argument("--frame-size", type=int, default=2048)
    ap.add_argument("--clips-per-launch", type=int, default=32,
                    help="target number of independent 32-tile batches that share one set of kernel launches "
                         "(dcs_separate_batch); every batch still counts as one step")
    ap.add_argument("--streams", type=int, default=3, help="launch groups in flight per GPU (HIP streams)")
    ap.add_argument("--min-time", type=float, default=0.25, help="seconds of timed rounds to accumulate")
    ap.add_argument("--max-rounds", type=int, default=4000)
    ap.add_argument("--sat-tiles", type=int, default=4096, help="tiles of the saturating extra run (0 = skip)")
    ap.add_argument("--legs", default="ikala,bach10_f16,bach10_f32,score_informed,transform",
                    help="comma list of extra BASELINE configs to measure on rank 0 at N=1 ('' = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the PCIe-inclusive extra leg")
    ap.add_argument("--no-cli", action="store_true", help="skip the command-line leg (separate_dsd.py on one wav, separate_batch.py on 50)")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the oracle check of the timed launches' output")
    ap.add_argument("--gather", choices=["allgather", "root", "none"], default="root",
                    help="N > 1: the collective that carries every launch group's int16 PCM inside the timed region: "
                         "gather to rank 0 (the default; north_star: 'final gather' -- only a writer needs the samples), "
                         "all_gather_into_tensor (every rank gets everything), or none (replicas: every rank keeps its own output)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="N > 1 with a round of ONE launch group (the driver's --steps 20): keep it one group (compute, then the "
                         "collective, on one stream) instead of two half-groups whose collective rides under the other's compute")
    ap.add_argument("--gather-impl", choices=["torch", "dcs"], default="torch",
                    help="who issues the collective: torch.distributed (backend nccl = RCCL), or libdcs's own C-ABI entry "
                         "dcs_gather on an RCCL communicator per HIP stream (deepconvsep_amd.dist.RcclComm)")
    ap.add_argument("--detail-stdout", action="store_true",
                    help="also print the full result object (legs, cli, saturating ...) on a stdout line BEFORE the compact "
                         "headline line; it is always written to bench_detail.json")
    ap.add_argument("--only-legs", action="store_true",
                    help="(counter passes) skip the headline timing: run only the legs given by --legs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs; there is no CPU path to measure")
    same_device = bool(os.environ.get("DCS_BENCH_SAME_DEVICE"))   # code-path check of the N > 1 leg on a 1-GPU box
    if same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # DCS_BENCH_FORCE_GATHER (tests on a 1-GPU box): run the N > 1 code path -- int16 conversion, collective, checks -- with
    # a world of ONE rank, so that `--gather-impl dcs` (RCCL through the C ABI) is exercised where only one GPU exists
    force_gather = bool(os.environ.get("DCS_BENCH_FORCE_GATHER")) and world == 1
    gathering = world > 1 or force_gather
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if same_device:                                           # RCCL refuses two ranks on one GPU: gloo for the check
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif force_gather:
        import socket
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                device_id=torch.device("cuda", local_rank))
    if args.gather_impl == "dcs" and gathering and same_device:
        raise SystemExit("bench.py: --gather-impl dcs needs one GPU per rank (RCCL refuses two ranks on one device)")

    import ctypes

    import deepconvsep_amd as dcs
    from deepconvsep_amd import _lib
    from deepconvsep_amd.arch import ARCHS, TILER_LIBRARY, TILER_SCRIPT
    from deepconvsep_amd.runtime import Context
    from deepconvsep_amd.synth import synth_audio, synth_params, synth_score_text

    if args.only_legs:
        legs = {}
        for name in [x for x in args.legs.split(",") if x]:
            if name == "transform":
                legs[name] = run_transform_leg(torch, dcs, synth_audio)
                continue
            legs[name] = run_leg(name, torch, dcs, _lib, ARCHS, TILER_LIBRARY, TILER_SCRIPT, synth_audio, synth_params,
                                 synth_score_text, not args.no_cpu_baseline, cpu_model(), os.cpu_count() or 1,
                                 with_parity=not args.no_parity_check)
        print(json.dumps({"legs": legs}))
        return
    N = args.frame_size
    F = N // 2 + 1
    K = max(1, args.steps)
    params = synth_params("dsd", TC, F, seed=2)
    L = samples_for_tiles(args.tiles)
    T = _lib.frame_count(L, HOP)
    n_tiles = _lib.tile_count(T, TC, OV, TILER_SCRIPT)
    assert n_tiles == args.tiles, (n_tiles, args.tiles)
    frames_per_step = (n_tiles - 1) * (TC - OV) + TC        # unique frames fully separated
    NS = max(1, args.streams)
    # N > 1 and a round of ONE launch group: compute -> int16 -> collective on one stream is serial by construction (the
    # collective of 20 steps is as long as their compute at N = 8).  Two half-groups on two lanes instead:
    # B's kernels are ordered behind A's kernels (an event, not a synchronisation), so A's collective rides under B's compute.
    groups, pipelined = plan_round(K, max(1, args.clips_per_launch), NS, gathering, args.gather, args.no_pipeline)
    CPL = max(groups)                                        # buffers are sized for the largest group

    gather_mode = [args.gather]                              # a list: the gather split below switches it off and on again

    class Lane(object):                                      # one HIP stream with everything it needs
        def __init__(self, idx):
            self.stream = torch.cuda.Stream()
            self.ctx = Context(stream=self.stream)           # binds libdcs (and every copy the package issues) to it
            with torch.cuda.stream(self.stream):
                self.sep = dcs.Separator("dsd", params, SCALE, TC, OV, 32, F, N, HOP, np.hanning, ctx=self.ctx)
                self.audio_h = np.stack([synth_audio(L, seed=100 + (rank * 16 + idx) * CPL + c) for c in range(CPL)])
                self.audio = self.ctx.to_device(self.audio_h, np.float32)          # [CPL, L]
                self.pcm = torch.empty((CPL, 4, L), dtype=torch.float32, device=self.audio.device)
                # multi-GPU: the separated PCM is gathered in the wav sample format (int16, separate_dsd.py:307-309);
                # RCCL has no int16 type, so the buffers travel as bytes
                self.pcm16 = torch.empty((CPL * 4, L), dtype=torch.int16, device=self.audio.device) if gathering else None
                self.gathered = (torch.empty((world * CPL * 4, L), dtype=torch.int16, device=self.audio.device)
                                 if gathering else None)
            # --gather-impl dcs: one RCCL communicator per lane (a communicator serialises its collectives; the lanes'
            # streams must not), made collectively in lane order on every rank
            self.comm = None
            if gathering and args.gather_impl == "dcs":
                from deepconvsep_amd.dist import RcclComm
                self.comm = RcclComm.from_process_group(device=self.ctx)
            self.stream.synchronize()
            # the C entry point with its arguments bound once: dcs_separate_batch() enqueues on the context's
            # own stream, so the host cost of a launch group is one ctypes call (seven eager kernel launches inside it;
            # DCS_GRAPH=1: one hipGraphLaunch from the second identical call on)
            net, plan = self.sep.net, self.sep.plan
            self._fn = self.ctx._lib.dcs_separate_batch
            self._to16 = self.ctx._lib.dcs_pcm_to_int16
            self._args = lambda nclips: (net._h, plan._h, ctypes.c_void_p(self.audio.data_ptr()), L, nclips, L, OV,
                                         TILER_SCRIPT, ctypes.c_float(SCALE), net.arch.eps_mode, 0,
                                         ctypes.c_void_p(self.pcm.data_ptr()), None, None)
            self._bound = {}
            self.launched_tiles = 0
            self.compute_done = torch.cuda.Event()

        def step(self, nclips, after=None):
            """One launch group = nclips steps (independent 32-tile batches).  `after`: an event of another lane this
            group's kernels are ordered behind (the pipelined N > 1 schedule)."""
            if after is not None:
                self.stream.wait_event(after)
            a = self._bound.get(nclips)
            if a is None:
                a = self._bound[nclips] = self._args(nclips)
            rc = self._fn(*a)
            if rc:
                _lib.check(rc)
            self.launched_tiles += nclips * n_tiles
            if pipelined:
                self.compute_done.record(self.stream)        # the kernels of this group are enqueued; its collective is not
            if gathering and gather_mode[0] != "none":
                rc = self._to16(self.ctx._h, ctypes.c_void_p(self.pcm.data_ptr()), nclips * 4 * L,
                                ctypes.c_void_p(self.pcm16.data_ptr()))
                if rc:
                    _lib.check(rc)
                self.gather(nclips)

        def gather(self, nclips):
            """RCCL over xGMI: the final gather of one launch group's separated PCM (int16, as bytes)."""
            if self.comm is not None:                       # dcs_gather on the lane's own stream: no torch in the data path
                root = 0 if gather_mode[0] == "root" else -1
                src = self.pcm16[: nclips * 4]
                full = self.gathered[: world * nclips * 4].view(world, nclips * 4, L) if (root < 0 or rank == root) else None
                self.comm.gather(self.ctx, src, full, root=root)
                return
            with torch.cuda.stream(self.stream):
                src = self.pcm16[: nclips * 4].view(torch.uint8)
                if gather_mode[0] == "root":
                    dst = None
                    if rank == 0:
                        full = self.gathered[: world * nclips * 4].view(torch.uint8)
                        dst = list(full.chunk(world, dim=0))
                    dist.gather(src, dst, dst=0)
                else:
                    dist.all_gather_into_tensor(self.gathered[: world * nclips * 4].view(torch.uint8), src)

    lanes = [Lane(i) for i in range(NS)]
    ctx0 = lanes[0].ctx

    def barrier():
        if gathering:
            dist.barrier()

    def timed(group_sizes, use):
        """Exactly sum(group_sizes) steps, round-robin over the lanes in `use`, bracketed by barrier + synchronize."""
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        prev = None
        for i, g in enumerate(group_sizes):
            ln = use[i % len(use)]
            ln.step(g, after=prev.compute_done if (pipelined and prev is not None and prev is not ln) else None)
            prev = ln
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - t0
        if gathering:
            tt = torch.tensor([el], dtype=torch.float64, device=lanes[0].audio.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    # ---- warm-up: at least W steps, and every (lane, group size) pair of the schedule at least twice (with DCS_GRAPH=1 the
    # second identical call captures its hipGraph) plus once more
    warm = 0
    while warm < max(args.warmup, 3 * K):
        for i, g in enumerate(groups):
            lanes[i % NS].step(g)
        warm += K
    for ln in lanes:
        for _ in range(3):
            ln.step(1)
    torch.cuda.synchronize()

    # ---- timed rounds
    round_s, events_tiles, events_rounds = [], 0, 0
    ctx0.timing_stride(1)
    ctx0.timing_reset()
    total = 0.0
    while (total < args.min_time or len(round_s) < 3) and len(round_s) < args.max_rounds:
        instrumented = (len(round_s) % 4 == 3) and not os.environ.get("DCS_BENCH_NOEVENTS")
        if instrumented:
            ctx0.timing(["final"])
            before = lanes[0].launched_tiles
        # an event round issues its K steps on stream 0 alone: with other streams' kernels sharing the CUs an event pair
        # would measure the kernel's time-sliced duration, not its own
        el = timed(groups, lanes[:1] if instrumented else lanes)
        if instrumented:
            ctx0.timing(None)
            events_tiles += lanes[0].launched_tiles - before
            events_rounds += 1
        round_s.append(el)
        total += el
    final_ms, final_launches = ctx0.timing_query("final")
    ctx0.timing_reset()
    med = statistics.median(round_s)
    value = world * frames_per_step * K / med
    clean = [t for i, t in enumerate(round_s) if i % 4 != 3] or round_s

    # ---- N > 1: what the collective costs -- the same rounds with the gather switched off, and the collective alone (one
    # launch group's payload per call, back to back on lane 0), so that a scaling curve can be read: compute vs exchange
    gather_split = None
    if gathering:
        saved = gather_mode[0]
        gather_mode[0] = "none"
        for _ in range(2):
            timed(groups, lanes)
        no_g = statistics.median([timed(groups, lanes) for _ in range(5)])
        gather_mode[0] = saved
        alone = None
        if saved != "none":
            g0 = groups[0]
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps_g = 20
            for _ in range(reps_g):
                lanes[0].gather(g0)
            torch.cuda.synchronize()
            barrier()
            alone = (time.perf_counter() - t0) / reps_g
            tt = torch.tensor([alone], dtype=torch.float64, device=lanes[0].audio.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            alone = float(tt.item())
        gather_split = {"mode": saved, "impl": ("dcs_gather (C ABI, RCCL communicator per HIP stream)" if args.gather_impl == "dcs"
                                                else "torch.distributed (%s)" % dist.get_backend()),
                        "payload_bytes_per_rank_per_group": int(groups[0] * 4 * L * 2),
                        "pipelined_half_groups": bool(pipelined),
                        "round_ms_without_gather": round(no_g * 1e3, 4), "round_ms_with_gather": round(med * 1e3, 4),
                        "ms_per_group_collective_alone": round(alone * 1e3, 4) if alone is not None else None,
                        "note": "rounds of the same K steps with the collective switched off, and the collective alone for one "
                                "launch group's int16 PCM; no scaling curve is implied by one run"}

    # ---- optional self-check of the gather (tests): every rank's slice of the gathered buffer must be that rank's own
    # int16 PCM, bit for bit (digests exchanged out of band)
    gather_check = None
    if gathering and os.environ.get("DCS_BENCH_CHECK_GATHER"):
        import hashlib
        ln, g0 = lanes[0], groups[0]
        ln.step(g0)
        torch.cuda.synchronize()
        mine = hashlib.sha1(ln.pcm16[: g0 * 4].cpu().numpy().tobytes()).hexdigest()
        digests = [None] * world
        dist.all_gather_object(digests, mine)
        got = ln.gathered[: world * g0 * 4].cpu().numpy()
        ok = all(hashlib.sha1(got[r * g0 * 4:(r + 1) * g0 * 4].tobytes()).hexdigest() == digests[r] for r in range(world))
        # ranks separate different audio: equal digests would mean the check compares a buffer with itself
        ok = ok and len(set(digests)) == world and bool(np.any(got))
        if gather_mode[0] == "root" and rank != 0:
            ok = True                                       # only the root holds the gathered buffer
        flags = [None] * world
        dist.all_gather_object(flags, bool(ok))
        gather_check = "ok" if all(flags) else "MISMATCH"

    # ---- one batch per launch on ONE stream (no overlap between batches): the BASELINE batch of 32 as one call
    k1 = max(K, 200)
    for _ in range(3):
        lanes[0].step(1)
    single_rounds = []
    tot1 = 0.0
    while tot1 < 0.1 or len(single_rounds) < 3:
        e = timed([1] * k1, lanes[:1])
        single_rounds.append(e / k1)
        tot1 += e
    el1 = statistics.median(single_rounds)
    ctx0.timing(["final"])
    ctx0.timing_stride(8)
    ctx0.timing_reset()
    timed([1] * k1, lanes[:1])
    final_ms1, final_launches1 = ctx0.timing_query("final")
    ctx0.timing(None)
    ctx0.timing_stride(1)

    def breakdown(nclips, reps):
        out = {}
        ctx0.timing("all")
        ctx0.timing_reset()
        for _ in range(reps):
            lanes[0].step(nclips)
        for tag in _lib.TAGS:
            ms, cnt = ctx0.timing_query(tag)
            if cnt:
                out[tag] = round(ms * cnt / reps, 5)
        ctx0.timing(None)
        ctx0.timing_reset()
        return out

    kernels_ms = breakdown(1, 20)            # per-kernel breakdown of one batch (separate instrumented pass)
    group_ms = breakdown(groups[0], 10)      # the same for one full launch group

    # algorithmic FLOPs of the dominant kernel: transposed conv1 of the 3 live branches
    # (separate_dsd.py:212,218,224): per tile 3 * 2 * tc * 50 * F  (DESIGN.md "kernels")
    final_flops_tile = 3 * 2 * TC * 50 * F
    # HBM bytes per launch of that kernel from the PMC passes committed under profiles/ (FETCH_SIZE and
    # WRITE_SIZE need their own rocprofv3 runs, scripts/gpu_traffic.sh); 2*FETCH + WRITE per the gfx950
    # correction of MI355X_MICROARCH.md.  None when the workload has no matching record.
    traffic_all = load_traffic().get("all", {})

    net0 = lanes[0].sep.net

    def roof(tiles_per_launch, ms, launches, frames=None, key=None, clips=1):
        """Roofline block of the dominant kernel for launches of `tiles_per_launch` tiles in `clips` clips.  `achieved`
        is ALGORITHMIC f32 work (the reference's count for the layers the kernel replaces) over the measured duration,
        priced against the f32 peak -- the arithmetic the path computes in.  When the launch runs a bf16x3 kernel the
        work is issued as six bf16 products per f32 product on K padded 50 -> 64 and bins padded to whole workgroups;
        `issued` prices that against the dense bf16 MFMA peak.  `traffic` = 2*FETCH + WRITE of the same kernel at the same
        grid size from the committed rocprofv3 --pmc passes (TRAFFIC_FILE), `traffic_ratio` = traffic / algorithmic bytes."""
        frames = (tiles_per_launch - 1) * (TC - OV) + TC if frames is None else frames
        ach = tiles_per_launch * final_flops_tile / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        clip_frames = int(round(frames / float(clips)))
        variant = net0.final_kernel(clip_frames, clips)
        rows16 = (clip_frames + 2 + 15) // 16          # the kernels run over all T frames of a clip (the tiles cover T - 2 or T - 3)
        binpad = {"bf16x3": 128, "f32x128": 128, "f32x64": 64, "one_batch": 64}.get(variant, 128)
        colg = (F + binpad - 1) // binpad
        kname = {"bf16x3": "final_bf16x3_kernel", "one_batch": "lat_final_kernel"}.get(variant, "final_kernel")
        wg_per_clip = rows16 * colg
        grid_threads = (512 if variant == "one_batch" else 256) * wg_per_clip * clips
        rec = traffic_all.get("%s@grid_threads=%d" % (kname, grid_threads)) if N == 2048 else None
        traffic = traffic_bytes(rec)
        g_bytes = 6 if variant in ("bf16x3", "one_batch") else 4
        alg_bytes = int(tiles_per_launch * (3 * TC * 56 * g_bytes) + frames * F * 4 * 5)
        r = {"bound": "mfma", "kernel": FINAL_NAMES.get(variant, variant), "variant": variant,
             "achieved": round(ach, 3), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
             "frac": round(ach / PEAK_F32_TFLOPS, 4), "traffic": traffic,
             "traffic_ratio": round(traffic / float(alg_bytes), 3) if traffic else None,
             # no committed counter record of this kernel at this grid: said so instead of silently dropping the ratio
             "traffic_key_missing": (None if rec else "%s@grid_threads=%d" % (kname, grid_threads)),
             "traffic_source": ("%s, record %s@grid_threads=%d (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                "this launch shape, 2*FETCH+WRITE; not measured in this run)"
                                % (TRAFFIC_FILE, kname, grid_threads)) if rec else None,
             # G rows read once (3 branches x tc x 56 stored channels) + mixture read + 4 sources written
             "algorithmic_bytes": alg_bytes,
             "avg_kernel_ms": round(ms, 5), "launches": int(launches),
             "tiles_per_launch": round(float(tiles_per_launch), 2)}
        if variant in ("bf16x3", "one_batch"):
            issued = ach * 6.0 * (64.0 / 50.0) * (colg * binpad / float(F))
            r["issued"] = {"achieved": round(issued, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s (bf16 MFMA)",
                           "frac": round(issued / PEAK_F16_TFLOPS, 4),
                           "note": "6 bf16 products per f32 product, K 50->64, bins padded to %d-bin workgroups" % binpad}
            r["note"] = ("frac prices f32-equivalent algorithmic work against the 157.3 TFLOP/s f32 peak; the kernel "
                         "executes on the bf16 matrix pipe (see issued), so frac may exceed what an f32 kernel can reach")
        return r

    ev_tiles = events_tiles / float(final_launches) if final_launches else groups[0] * n_tiles
    roofline = roof(ev_tiles, final_ms, final_launches, frames=ev_tiles / n_tiles * frames_per_step,
                    key="final_kernel_%dx%d_tiles" % (groups[0], n_tiles), clips=max(1, int(round(ev_tiles / n_tiles))))
    roofline["timed"] = ("HIP events around every launch of the kernel in %d of the %d timed rounds; those rounds run "
                         "their K steps on stream 0 alone" % (events_rounds, len(round_s)))
    single = {"ms_per_step": round(el1 * 1e3, 5), "value": round(world * frames_per_step / el1, 1),
              "whole_path_frac_of_f32_peak": round(n_tiles * ARCHS["dsd"].flops_per_tile(TC, F) / el1 / 1e12 / PEAK_F32_TFLOPS, 4),
              "kernels": "one-batch kernels (csrc/dsd_lat.hip), one launch per layer, issued eagerly" if net0.final_kernel(T, 1) == "one_batch"
                         else "throughput kernels",
              "steps_per_round": k1, "rounds": len(single_rounds),
              "roofline": roof(n_tiles, final_ms1, final_launches1), "kernels_ms": kernels_ms,
              "kernels_ms_sum": round(sum(kernels_ms.values()), 5)}
    hbm_stages = None
    launch_group = {"clips": groups[0], "tiles": groups[0] * n_tiles, "kernels_ms": group_ms,
                    "kernels_ms_sum": round(sum(group_ms.values()), 5)}
    # the two HBM-bound kernels of the group against 8 TB/s: algorithmic bytes per frame x the frames of the launch
    # over the kernel's HIP-event duration; counter traffic = the record of that kernel whose grid is closest to this launch's
    if N == 2048:
        g_frames = groups[0] * T
        hbm = {}
        for tag, kname, per_frame, threads in (("stft", "stft_forward_wave_kernel", HOP * 4 + 2 * F * 4, 64.0 * g_frames),
                                               ("istft", "istft_", (4 + 1) * F * 4 + 4 * HOP * 4, 512.0 * groups[0] * 4 * 3)):
            ms = group_ms.get(tag)
            if not ms:
                continue
            alg = int(per_frame * g_frames)
            key = None
            for pref in ((kname,) if tag == "stft" else ("istft_chain_kernel", "istft_seq_kernel", "istft_")):
                cands = [(abs(int(k.split("=")[1]) - threads), k) for k in traffic_all if k.startswith(pref) and "=" in k]
                if cands and min(cands)[0] <= 0.25 * threads:           # a record of (nearly) this launch shape, else none
                    key = min(cands)[1]
                    break
            tr = traffic_bytes(traffic_all.get(key)) if key else None
            hbm[tag] = {"ms": ms, "algorithmic_bytes": alg, "GBps": round(alg / (ms * 1e-3) / 1e9, 1),
                        "frac_of_hbm_peak": round(alg / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4), "traffic": tr,
                        "traffic_ratio": round(tr / float(alg), 3) if tr else None, "traffic_record": key}
        launch_group["hbm_kernels"] = hbm
        # north_star "rocprof reports achieved HBM GB/s on STFT/mask stages": the same numbers, compact, for the headline
        hbm_stages = {t: {"frac": v["frac_of_hbm_peak"], "GBps": v["GBps"], "traffic_ratio": v["traffic_ratio"]}
                      for t, v in hbm.items()}

    # ---- saturating regime (extra): same path, one long clip per launch, one stream
    saturating = None
    if args.sat_tiles and rank == 0:
        net, plan = lanes[0].sep.net, lanes[0].sep.plan
        with torch.cuda.stream(lanes[0].stream):
            Ls = samples_for_tiles(args.sat_tiles)
            a2 = ctx0.to_device(synth_audio(Ls, seed=7), np.float32)
            out2 = torch.empty((4, Ls), dtype=torch.float32, device=a2.device)
            for _ in range(3):
                net.separate(plan, a2, OV, TILER_SCRIPT, SCALE, out=out2)
            torch.cuda.synchronize()
            reps2, tot2, per2 = 0, 0.0, []
            while tot2 < 0.1 or reps2 < 5:
                t0 = time.perf_counter()
                net.separate(plan, a2, OV, TILER_SCRIPT, SCALE, out=out2)
                torch.cuda.synchronize()
                per2.append(time.perf_counter() - t0)
                tot2 += per2[-1]
                reps2 += 1
            e2 = statistics.median(per2)
            ctx0.timing("all")
            ctx0.timing_reset()
            k2 = 5
            for _ in range(k2):
                net.separate(plan, a2, OV, TILER_SCRIPT, SCALE, out=out2)
        fr2 = (args.sat_tiles - 1) * (TC - OV) + TC
        sat_k = {}
        for tag in _lib.TAGS:
            ms, cnt = ctx0.timing_query(tag)
            if cnt:
                sat_k[tag] = round(ms * cnt / k2, 5)
        ctx0.timing(None)
        ctx0.timing_reset()
        total_flops = args.sat_tiles * ARCHS["dsd"].flops_per_tile(TC, F)
        saturating = {"tiles": args.sat_tiles, "audio_seconds": round(Ls / SR, 2), "value": round(fr2 / e2, 1),
                      "unit": "frames/s", "x_realtime": round(fr2 / e2 * HOP / SR, 1),
                      "ms_per_step": round(e2 * 1e3, 4), "rounds": reps2,
                      "roofline": roof(args.sat_tiles, sat_k.get("final", 0.0), k2),
                      "kernels_ms": sat_k, "kernels_ms_sum": round(sum(sat_k.values()), 5),
                      "whole_path_algorithmic_tflops": round(total_flops / e2 / 1e12, 2),
                      "whole_path_frac_of_f32_peak": round(total_flops / e2 / 1e12 / PEAK_F32_TFLOPS, 4)}
        del a2, out2

    # ---- host-fed regime (extra, never `value`): float32 audio comes from pinned host memory and the int16 PCM goes
    # back to it, every launch group, over PCIe; the copies ride on the lanes' streams so one lane's transfers overlap
    # the other lane's kernels
    host_fed = None
    if rank == 0 and world == 1 and not args.no_host_fed:
        G0 = groups[0]
        for ln in lanes:
            with torch.cuda.stream(ln.stream):
                ln.audio_pin = torch.from_numpy(ln.audio_h.astype(np.float32)).pin_memory()
                ln.pcm16 = torch.empty((CPL * 4, L), dtype=torch.int16, device=ln.audio.device)
                ln.out_pin = torch.empty((CPL * 4, L), dtype=torch.int16).pin_memory()

        def host_group(ln):
            with torch.cuda.stream(ln.stream):
                ln.audio[:G0].copy_(ln.audio_pin[:G0], non_blocking=True)
            ln.step(G0)
            rc = ln._to16(ln.ctx._h, ctypes.c_void_p(ln.pcm.data_ptr()), G0 * 4 * L, ctypes.c_void_p(ln.pcm16.data_ptr()))
            if rc:
                _lib.check(rc)
            # SDMA copy; letting the conversion kernel write the pinned buffer directly (2-byte stores over PCIe)
            # measured 6.8 GB/s against 29.5 GB/s this way
            with torch.cuda.stream(ln.stream):
                ln.out_pin[:G0 * 4].copy_(ln.pcm16[:G0 * 4], non_blocking=True)

        for ln in lanes:
            host_group(ln)
        torch.cuda.synchronize()
        kg = 8 * NS
        t0 = time.perf_counter()
        for i in range(kg):
            host_group(lanes[i % NS])
        torch.cuda.synchronize()
        eh = time.perf_counter() - t0
        host_fed = {"value": round(frames_per_step * G0 * kg / eh, 1), "unit": "frames/s",
                    "ms_per_step": round(eh / (kg * G0) * 1e3, 5),
                    "bytes_per_step": {"h2d_f32_audio": int(L * 4), "d2h_int16_pcm": int(4 * L * 2)},
                    "pcie_GBps": round((L * 4 + 4 * L * 2) * G0 * kg / eh / 1e9, 1),
                    "note": "pinned host buffers, async copies on the lanes' streams; not part of `value`"}
    if gathering:
        dist.barrier()

    # ---- CPU baseline: the oracle on this host's cores, same 32-tile batch (rank 0, N=1 only)
    ncpu = os.cpu_count() or 1
    cpu_name = cpu_model()
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pipeline
        audio_h = lanes[0].audio_h[0]

        def run_oracle():
            pipeline.separate("dsd", params, audio_h, SCALE, TC, OV, 32, N, HOP, np.hanning)

        # the reference's own loops are single-threaded Python; its Theano convolutions / dots may use BLAS threads.
        # Reported: one thread, the best of a few thread counts (all 256 hardware threads on 50-channel float64
        # convolutions are far slower than a handful), and all hardware threads.
        per_threads = {}
        for nt in sorted(set([1, 4, 8, 16, min(32, ncpu)])):
            torch.set_num_threads(nt)
            run_oracle()
            sec, reps, el = time_cpu(run_oracle, 1.5)
            per_threads[nt] = (sec, reps, el)
        best_nt = min(per_threads, key=lambda k: per_threads[k][0])
        torch.set_num_threads(best_nt)
        sec_b, reps_b, el_b = time_cpu(run_oracle, 8.0, 400)
        torch.set_num_threads(ncpu)
        run_oracle()
        sec_all, reps_all, el_all = time_cpu(run_oracle, 2.0)
        torch.set_num_threads(best_nt)
        cpu_baseline = {"value": round(frames_per_step / sec_b, 1), "unit": "frames/s",
                        "cores": int(best_nt), "kind": "port",
                        "single_thread": {"value": round(frames_per_step / per_threads[1][0], 1), "cores": 1},
                        "all_cores": {"value": round(frames_per_step / sec_all, 1), "cores": int(ncpu)},
                        "by_threads": {str(k): round(frames_per_step / v[0], 1) for k, v in per_threads.items()},
                        "cpu_model": cpu_name, "host_cpu_count": int(ncpu),
                        "label": "reference-equivalent CPU path (Theano unavailable)",
                        "sample": "%d x the same 32-tile / %.2f s batch through oracle.pipeline.separate "
                                  "(reference NumPy STFT/tiling/overlap-add loops + torch-CPU float64 network; "
                                  "Theano/Lasagne unavailable), %.1f s of CPU time at %d threads (the best of "
                                  "1/4/8/16/32); %.1f s more for the thread sweep and the all-core run"
                                  % (reps_b, L / SR, el_b, best_nt,
                                     sum(v[2] for v in per_threads.values()) + el_all)}

    # ---- the other BASELINE configs (rank 0, N=1): whole path per clip, per-kernel breakdown, roofline of the
    # dominant kernel, CPU oracle on a bounded sample of the same input
    legs = None
    if rank == 0 and world == 1 and args.legs:
        legs = {}
        for name in [x for x in args.legs.split(",") if x]:
            try:
                if name == "transform":
                    legs[name] = run_transform_leg(torch, dcs, synth_audio)
                    continue
                legs[name] = run_leg(name, torch, dcs, _lib, ARCHS, TILER_LIBRARY, TILER_SCRIPT, synth_audio, synth_params,
                                     synth_score_text, not args.no_cpu_baseline, cpu_name, ncpu,
                                     with_parity=not args.no_parity_check)
            except Exception as exc:     # a leg must not take the headline line down with it
                legs[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}

    # ---- the timed launches did the work: PCM of one clip of the last timed launch group (and of the last one-batch
    # call) against the CPU oracle on the same audio (outside every timed region; the oracle only checks here)
    parity_check = None
    if not args.no_parity_check:
        from oracle import pipeline
        ln = lanes[0]
        g_last = groups[0]
        ln.step(g_last)                                     # the launch shape of the timed rounds, same buffers
        torch.cuda.synchronize()
        clip = g_last - 1
        got = ln.pcm[clip].cpu().numpy().astype(np.float64)
        want, _mm, mag_o, _ph = pipeline.separate("dsd", params, ln.audio_h[clip], SCALE, TC, OV, 32, N, HOP, np.hanning,
                                                  return_spectra=True)
        err_group = float(np.max(np.abs(got - want)))
        # per mask bin: the same clip's batch of tiles through the network operators against the float64 oracle
        from oracle import tiling_np
        fb, nch = tiling_np.generate_overlapadd(mag_o, mag_o.shape[-1], TC, OV, 32, tiler=tiling_np.SCRIPT, fill=0.0)
        bins = mask_bin_report(net0, "dsd", params, fb.reshape((-1,) + fb.shape[2:])[:nch], 'A')
        ln.step(1)                                          # one batch per call (single_stream)
        torch.cuda.synchronize()
        got1 = ln.pcm[0].cpu().numpy().astype(np.float64)
        want1 = pipeline.separate("dsd", params, ln.audio_h[0], SCALE, TC, OV, 32, N, HOP, np.hanning)
        err_single = float(np.max(np.abs(got1 - want1)))
        worst = max(err_group, err_single)
        if gathering:
            tt = torch.tensor([worst], dtype=torch.float64, device=lanes[0].audio.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            worst = float(tt.item())
        parity_check = {"max_abs_pcm_err": worst, "tolerance": 1e-4,
                        "ok": bool(worst < 1e-4 and np.isfinite(worst) and bins["criterion_ok"] and
                                   bins["network_output_max_err"] < 1e-4),
                        "mask_criterion_ok": bins["criterion_ok"],
                        "mask_bins": bins["mask_bins"], "masked_bins_outside_1e4": bins["masked_bins_outside_1e4"],
                        "conditioned_fraction": bins["conditioned_fraction"],
                        "network_output_max_err": bins["network_output_max_err"], "mask_bin_check": bins,
                        "vs": "oracle.pipeline.separate (reference NumPy STFT / tiling / overlap-add + float64 network) on the same audio",
                        "launch_group": {"clips": int(g_last), "clip_checked": int(clip), "tiles": int(n_tiles),
                                         "max_abs_pcm_err": err_group, "final_kernel": net0.final_kernel(T, g_last)},
                        "single_stream": {"tiles": int(n_tiles), "max_abs_pcm_err": err_single,
                                          "final_kernel": net0.final_kernel(T, 1)},
                        "ranks": int(world)}

    whole_flops_step = n_tiles * ARCHS["dsd"].flops_per_tile(TC, F)
    cli = None
    if rank == 0 and world == 1 and not args.no_cli:
        try:
            cli = run_cli(torch, dcs, synth_audio, synth_params)
        except Exception as exc:
            cli = {"error": "%s: %s" % (type(exc).__name__, exc)}

    if rank == 0:
        line = {
            "metric": "spectrogram-frames/s", "value": round(value, 1), "unit": "frames/s", "n_gpus": world,
            "steps": K, "warmup": args.warmup, "ms_per_step": round(med / K * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "x_realtime": round(value * HOP / SR, 1),
            "whole_path_algorithmic_tflops": round(world * whole_flops_step * K / med / 1e12, 2),
            "whole_path_frac_of_f32_peak": round(whole_flops_step * K / med / 1e12 / PEAK_F32_TFLOPS, 4),
            "parity_check": parity_check,
            # BASELINE configs[1] exactly as the reference issues it -- predict_function2 on ONE batch of 32 tiles per call
            # (separate_dsd.py:296-298), one stream, no batching across calls; details in `single_stream`
            "single_stream_ms_per_step": single["ms_per_step"], "single_stream_value": single["value"],
            "single_stream_x_realtime": round(single["value"] * HOP / SR, 1),
            "rounds": len(round_s), "timed_region_s": round(total, 4),
            "round_ms": {"median": round(med * 1e3, 4), "min": round(min(round_s) * 1e3, 4),
                         "max": round(max(round_s) * 1e3, 4),
                         "median_without_event_rounds": round(statistics.median(clean) * 1e3, 4)},
            "config": {"workload": "DSD100 4-source separate_dsd path (BASELINE configs[1]): frameSize=%d hop=512 "
                                   "hann, time_context=30 overlap=25 scale=0.3, one batch of %d tiles = %.2f s of "
                                   "44.1 kHz audio per GPU per step, STFT->net->mask->overlap-add->iSTFT, "
                                   "input and output resident in HBM; the %d steps of a round go out as launch "
                                   "groups of %s batches (dcs_separate_batch, the batch-of-files driver) over %d "
                                   "HIP streams per GPU%s; a round = exactly %d steps between barrier+synchronize, "
                                   "median of %d rounds"
                                   % (N, n_tiles, L / SR, K, "/".join(str(g) for g in sorted(set(groups), reverse=True)),
                                      NS, (", int16 PCM %s over RCCL%s" % ({"root": "gathered to rank 0", "allgather": "all-gathered", "none": "kept per rank"}[args.gather],
                                                                             ", two half-groups pipelined (collective of one under the compute of the other)" if pipelined else ""))
                                      if world > 1 else "", K, len(round_s)),
                       "tiles_per_gpu_per_step": n_tiles, "frames_per_gpu_per_step": frames_per_step,
                       "frame_size": N, "bins": F, "launch_groups_per_round": groups, "streams_per_gpu": NS,
                       "weights": "synthetic Glorot-uniform, seed 2, 15-array DSD .pkl layout",
                       "parallelism": "tiles sharded by rank (dp%d)" % world},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "single_stream": single, "launch_group": launch_group,
            "saturating": saturating, "host_fed": host_fed, "legs": legs, "cli": cli,
            # the two HBM-bound kernels of the launch group against 8 TB/s (algorithmic bytes of SURVEY 8d over HIP-event time)
            "hbm_stages": hbm_stages,
        }
        if gather_check is not None:
            line["gather_check"] = gather_check
        if gather_split is not None:
            line["gather"] = gather_split
        emit(line, detail_stdout=args.detail_stdout)
    if gathering:
        for ln in lanes:
            if ln.comm is not None:
                ln.comm.close()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ the command lines
def run_cli(torch, dcs, synth_audio, synth_params):
    """What a user of the reference gets (separate_dsd.py -i -o -m; separate_multiple.ipynb loops it over a folder):
    the drop-in command on one 10 s stereo wav as a process (wall clock incl. interpreter + torch import + model upload)
    and stage by stage in this process, and examples/separate_batch.py over 50 wavs with the model resident."""
    import subprocess
    from deepconvsep_amd import separation as sp
    tmp = tempfile.mkdtemp(prefix="dcs_cli_")
    F = 513                                                  # the as-shipped DSD config: frameSize 1024, hop 512
    params = synth_params("dsd", TC, F, seed=2)
    model = os.path.join(tmp, "model.pkl")
    dcs.save_model(model, params)
    Lc = 441000
    wavs = []
    for i in range(50):
        w = os.path.join(tmp, "clip%02d.wav" % i)
        sp.write_wav(w, synth_audio(Lc, seed=500 + i, channels=2), SR)
        wavs.append(w)
    out1 = os.path.join(tmp, "out1")
    os.makedirs(out1)
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "dsd100", "separate_dsd.py"), "-i", wavs[0], "-o", out1,
                        "-m", model], capture_output=True, text=True, timeout=600)
    t_proc = time.perf_counter() - t0
    if r.returncode != 0 or not os.path.isfile(os.path.join(out1, "vocals.wav")):
        raise RuntimeError("separate_dsd.py failed: " + (r.stderr or r.stdout)[-300:])
    # the same steps in this process (libraries already loaded), timed one by one
    st = {}
    t0 = time.perf_counter(); p2 = sp.load_model(model); st["load_model_pkl"] = time.perf_counter() - t0
    t0 = time.perf_counter(); sr, audio = sp.read_wav(wavs[0]); mono = sp.to_mono(audio, "dsd"); st["read_wav_to_mono"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    sep = sp.Separator("dsd", p2, SCALE, TC, 25, 32, F, 1024, HOP, np.hanning)
    torch.cuda.synchronize(); st["model_upload_and_plan"] = time.perf_counter() - t0
    sep.separate(mono); torch.cuda.synchronize()            # first call: workspace allocation
    ctx = sep.ctx
    t0 = time.perf_counter(); a = ctx.to_device(mono, np.float32); torch.cuda.synchronize(); st["h2d_float32_audio"] = time.perf_counter() - t0
    t0 = time.perf_counter(); pcm_d = sep.net.separate(sep.plan, a, 25, sep.tiler, SCALE); torch.cuda.synchronize(); st["kernels"] = time.perf_counter() - t0
    t0 = time.perf_counter(); pcm = ctx.to_host(pcm_d).astype(np.float64); st["d2h_and_float64"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    for path, sig in zip(sp.output_paths("dsd", wavs[0], out1), pcm):
        sp.write_wav(path, sig, sr)
    st["write_4_wavs"] = time.perf_counter() - t0
    t0 = time.perf_counter(); sep.separate(mono); st["Separator.separate_total"] = time.perf_counter() - t0
    out50 = os.path.join(tmp, "out50")
    os.makedirs(out50)
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "separate_batch.py"), "-a", "dsd", "-m", model, "-o", out50]
                       + wavs, capture_output=True, text=True, timeout=900)
    t_batch = time.perf_counter() - t0
    if r.returncode != 0 or not os.path.isfile(os.path.join(out50, "clip49", "vocals.wav")):
        raise RuntimeError("separate_batch.py failed: " + (r.stderr or r.stdout)[-300:])
    frames = int(np.ceil(Lc / float(HOP))) + 2
    # steady state: the same driver over 550 files (the 50 wavs eleven times: real reads, real separations, real writes) with
    # --stats, which reports the time from model-ready to the last wav written -- interpreter start, torch import and model
    # upload (what the 50-file figure mostly measures) are not in it
    out550 = os.path.join(tmp, "out550")
    os.makedirs(out550)
    steady = None
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "separate_batch.py"), "-a", "dsd", "-m", model, "-o", out550,
                        "--stats"] + wavs * 11, capture_output=True, text=True, timeout=1800)
    if r.returncode == 0:
        for ln in reversed(r.stdout.splitlines()):
            if ln.startswith("{"):
                steady = json.loads(ln)
                break
    import shutil
    res = {"workload": "DSD as shipped (frameSize 1024, hop 512, 513 bins), 10 s stereo 44.1 kHz int16 wavs, synthetic weights",
           "separate_dsd_py_process_s": round(t_proc, 3),
           "separate_dsd_py_note": "python separate_dsd.py -i clip.wav -o out -m model.pkl as a process: interpreter start, torch / "
                                   "libdcs import, model upload, one file, 4 wavs written",
           "in_process_ms": {k: round(v * 1e3, 3) for k, v in st.items()},
           "separate_batch_py_50_files_s": round(t_batch, 3),
           "separate_batch_py_ms_per_file": round(t_batch / 50.0 * 1e3, 2),
           "separate_batch_py_x_realtime": round(50 * Lc / float(SR) / t_batch, 1),
           "separate_batch_py_frames_per_s": round(50 * frames / t_batch, 1)}
    if steady:
        res["steady_state"] = {"files": steady["files"], "seconds_after_model_ready": steady["seconds_after_model_ready"],
                               "ms_per_file": steady["ms_per_file"],
                               "x_realtime": round(Lc / float(SR) / (steady["ms_per_file"] * 1e-3), 1),
                               "frames_per_s": round(frames / (steady["ms_per_file"] * 1e-3), 1),
                               "workers": steady.get("workers"), "path": steady.get("path"), "group": steady.get("group"),
                               "ms_per_file_after_first_group": steady.get("ms_per_file_after_first_group"),
                               "main_thread_ms_per_file": steady.get("main_thread_ms_per_file"),
                               "note": "separate_batch.py --stats over 550 wav files (int16 frames read into pinned staging by "
                                       "the I/O threads of libdcs.so, H2D, device mix-down, kernels, device int16 conversion, "
                                       "D2H, 4 wavs written per file by the same threads; the main thread enqueues a group and "
                                       "collects it one iteration later), clock started when the model is resident"}
        # the float path of the single-file scripts through the same driver (round 4's figure: 2.0 - 2.3 ms per file)
        r2 = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "separate_batch.py"), "-a", "dsd", "-m", model, "-o",
                             out550, "--stats", "--float-path"] + wavs * 4, capture_output=True, text=True, timeout=1800) \
            if os.path.isdir(tmp) else None
        if r2 is not None and r2.returncode == 0:
            for ln in reversed(r2.stdout.splitlines()):
                if ln.startswith("{"):
                    res["steady_state"]["float_path_ms_per_file"] = json.loads(ln)["ms_per_file"]
                    break
    shutil.rmtree(tmp, ignore_errors=True)
    return res


# ------------------------------------------------------------------------------------------------ other configs
def arch_work(arch, tc, F, n, f16):
    """Per launch of n tiles: algorithmic FLOPs and HBM bytes of every kernel tag of the generic graphs
    (SURVEY 8a-4', 8d), the matrix peak that applies to it."""
    d = arch.dims(tc, F)
    NB, C = arch.live_branches(), arch.C      # the branches predict_function2 needs (score-informed: 1 of 4)
    conv1 = 2.0 * d['nf1'] * C * d['kw1'] * tc * d['w1']
    conv2 = 2.0 * d['nf2'] * d['nf1'] * d['kh2'] * d['kw2'] * d['h2'] * d['w2']
    fc = 2.0 * d['flat'] * arch.hidden
    plane_in, plane1, plane2 = C * tc * F * 4.0, d['nf1'] * tc * d['wp'] * 4.0, d['flat'] * 4.0
    mfma16 = PEAK_F16_TFLOPS if f16 else PEAK_F32_TFLOPS
    # what the kernels MOVE (round-4 re-pricing; the round-3 line priced the fused-pool kernels on the unfused layer's bytes and
    # showed traffic ratios of 0.42): with a pooling layer conv1 writes the POOLED rows + 4 routing bits per window and never
    # the full-resolution activations, and its transpose reads the pooled gradient + the bits
    if arch.pool_w:
        windows = d['nf1'] * tc * d['wp']
        conv1_out = windows * 4.0 + windows * 0.5          # pooled f32 rows + 4 bits per window
    else:
        conv1_out = d['nf1'] * tc * d['w1'] * 4.0
    # kind: which unit executes the flops -- 'mfma' (matrix pipe) or 'valu' (register-blocked vector kernels: conv1_reg /
    # deconv1_reg of the stride-3 iKala graph); both have the 157.3 TFLOP/s f32 peak, the label says which one it is
    vec1 = 'valu' if arch.conv1[2] == 3 else 'mfma'
    # f16 switch (BASELINE configs[3], SURVEY 8d "fp16: HBM on weights", 426.7 MB): the dense weights and the per-source layers'
    # output D are 2 bytes per element since round 6 (gemm_f16.hip)
    wb = 2.0 if f16 else 4.0
    plane2_d = d['flat'] * wb
    return {
        "conv1": (n * conv1, n * (plane_in + conv1_out), PEAK_F32_TFLOPS, vec1),
        "conv2": (n * conv2, n * (plane1 + plane2), mfma16, 'mfma'),
        "fc": (n * fc, d['flat'] * arch.hidden * wb + n * (plane2 + arch.hidden * 4.0), mfma16 if f16 else PEAK_F32_TFLOPS, 'mfma'),
        "fc1x": (n * NB * fc, NB * d['flat'] * arch.hidden * wb + n * NB * plane2_d, mfma16 if f16 else PEAK_F32_TFLOPS, 'mfma'),
        "deconv2": (n * NB * conv2, n * NB * (plane2 + plane1), mfma16, 'mfma'),
        "final": (n * NB * conv1, n * NB * (conv1_out + plane_in), PEAK_F32_TFLOPS, vec1),
        # both InverseLayers in one kernel: the activations between them never reach HBM
        "decoder": (n * NB * (conv2 + conv1), n * NB * (plane2_d + plane_in), mfma16, 'mfma'),
    }


def kernel_roofline(tag, ms, flops, nbytes, peak_tf, name, unit_kind='mfma'):
    """Roofline of one kernel: the bound is whichever of flops / peak and bytes / 8 TB/s is the longer time; a flop-bound
    kernel is labelled with the unit that executes it ('mfma', or 'valu' for the register-blocked vector kernels)."""
    t = ms * 1e-3
    t_mfma, t_hbm = flops / (peak_tf * 1e12), nbytes / (PEAK_HBM_GBPS * 1e9)
    if t_mfma >= t_hbm:
        ach = flops / t / 1e12 if t > 0 else 0.0
        return {"bound": unit_kind, "kernel": name, "tag": tag, "achieved": round(ach, 3), "peak": peak_tf,
                "unit": "TFLOP/s", "frac": round(ach / peak_tf, 4), "traffic": None,
                "algorithmic_flops": int(flops), "algorithmic_bytes": int(nbytes), "avg_kernel_ms": round(ms, 5)}
    ach = nbytes / t / 1e9 if t > 0 else 0.0
    return {"bound": "hbm", "kernel": name, "tag": tag, "achieved": round(ach, 1), "peak": PEAK_HBM_GBPS,
            "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBPS, 4), "traffic": None,
            "algorithmic_flops": int(flops), "algorithmic_bytes": int(nbytes), "avg_kernel_ms": round(ms, 5)}


KERNEL_NAMES = {
    "conv1": "conv1_kernel (strided conv1 + biases)", "conv2": "conv2 (slab / column convolution, MFMA)",
    "fc": "gemm_rows (bottleneck DenseLayer)", "fc1x": "gemm_rows (per-source DenseLayers)",
    "deconv2": "transposed conv2 (slab / column convolution, MFMA)", "final": "transposed conv1 (deconv1_reg / deconv1)",
    "decoder": "colconv_deconv1_fused_kernel / colconv_deconv1_fused_x3_kernel (transposed conv2 + transposed conv1 in one kernel, "
               "conv2^T weights in registers; f16 inputs with the switch, three-way split bf16 operands without)",
}


_ORACLE_CACHE = {}     # (graph, frame size, overlap, weight seed, samples) -> (oracle PCM, cpu_baseline block): the two Bach10 legs share one


def run_transform_leg(torch, dcs, synth_audio):
    """transformFFT.compute_transform as the dataset builders call it (examples/dsd100/compute_features.py:83-112:
    one song = the mixture and its four sources as the columns of audio[t, i]).  Per (frameSize, precision): the ONE
    STFT launch over all columns with input and output resident in HBM (HIP events; HBM-bound: algorithmic bytes = samples
    read + magnitudes written, against 8 TB/s) and the whole call host array -> host array (upload, device transpose,
    launch, pinned download)."""
    from deepconvsep_amd.runtime import default_context
    ctx = default_context()
    cols, seconds = 5, 60.0
    L = int(seconds * SR)
    base = np.stack([synth_audio(L, seed=900 + c) for c in range(cols)], axis=1)
    res = {"workload": "compute_transform(audio[t, 5]): 60 s of 44.1 kHz audio x 5 columns (mixture + 4 sources), hop 512, "
                       "magnitudes only, all columns in one launch", "cases": {}}
    for N, precision in ((1024, 'float64'), (4096, 'float64'), (1024, 'float32'), (4096, 'float32')):
        tt = dcs.transformFFT(frameSize=N, hopSize=HOP, precision=precision)
        plan = tt._get_plan()
        dt = np.float64 if precision == 'float64' else np.float32
        a = ctx.to_device(np.ascontiguousarray(base.T), dt)
        T, F = _frame_count(L), N // 2 + 1
        mag = torch.empty((cols, T, F), dtype=a.dtype, device=a.device)
        for _ in range(3):
            plan.forward_clips(a, phase=False, mag_out=mag)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        with ctx.stream_scope():
            ev0.record()
            for _ in range(reps):
                plan.forward_clips(a, phase=False, mag_out=mag)
            ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        nbytes = cols * (L + T * F) * np.dtype(dt).itemsize
        tt.compute_transform(base, phase=False, save=False)
        t0 = time.perf_counter()
        tt.compute_transform(base, phase=False, save=False)
        wall = time.perf_counter() - t0
        # as the dataset builders call it: results into the .data / .shape files (pinned staging block reused song after song)
        tdir = tempfile.mkdtemp(prefix="dcs_tf_")
        tt.suffix = "b"
        tt.compute_transform(base, out_path=os.path.join(tdir, "song.data"), phase=False, save=True)
        t0 = time.perf_counter()
        tt.compute_transform(base, out_path=os.path.join(tdir, "song.data"), phase=False, save=True)
        wall_file = time.perf_counter() - t0
        import shutil
        shutil.rmtree(tdir, ignore_errors=True)
        res["cases"]["N%d_%s" % (N, precision)] = {
            "frames": int(cols * T), "kernel_ms": round(ms, 4), "frames_per_s": round(cols * T / (ms * 1e-3), 1),
            "algorithmic_bytes": int(nbytes), "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1),
            "frac_of_hbm_peak": round(nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
            "host_to_host_ms": round(wall * 1e3, 2), "host_to_host_frames_per_s": round(cols * T / wall, 1),
            "to_data_file_ms": round(wall_file * 1e3, 2), "to_data_file_frames_per_s": round(cols * T / wall_file, 1),
            "output_MB": round(cols * T * F * 8 / 1e6, 1)}
        del a, mag, tt
    torch.cuda.empty_cache()
    return res


def _frame_count(L, hop=HOP):
    return int(np.ceil(L / float(hop)) + 2)


def run_leg(name, torch, dcs, _lib, ARCHS, TILER_LIBRARY, TILER_SCRIPT, synth_audio, synth_params, synth_score_text,
            with_cpu, cpu_name, ncpu, with_parity=True):
    from deepconvsep_amd.runtime import default_context
    ctx = default_context()
    if name == "ikala":
        # BASELINE configs[0]: iKala 2-source, frameSize=2048 hop=512, time_context=30, one 10 s wav (the script sums
        # L+R of a stereo file, separate_ikala.py:229; overlap 20, :275)
        arch_name, Nf, ov, window, f16, batch, seed = "ikala", 2048, 20, np.hanning, False, 32, 1
        Lc = 441000
        st = synth_audio(Lc, seed=0, channels=2)
        audio = st[:, 0] + st[:, 1]
        library, melody = False, None
        what = "iKala 2-source separate_ikala path (BASELINE configs[0]): frameSize=2048 hop=512 hann, overlap 20, 10 s stereo wav summed L+R, f32"
    elif name == "bach10_f16":
        # BASELINE configs[3]: Bach10 4-instrument, fp16 MFMA conv path (frameSize 4096 blackmanharris, separate_bach10.py:325)
        arch_name, Nf, ov, window, f16, batch, seed = "bach10", 4096, 25, dcs.blackmanharris, True, 32, 3
        Lc = 441000
        audio = synth_audio(Lc, seed=0)
        library, melody = False, None
        what = "Bach10 4-instrument separate_bach10 path (BASELINE configs[3]): frameSize=4096 hop=512 blackmanharris, overlap 25, 10 s mono, conv2 / conv2^T with f16 inputs + f32 accumulation (MFMA), everything else f32"
    elif name == "bach10_f32":
        # the same Bach10 clip with every layer f32-class (no f16 switch): the 1e-4 result of that graph; its CPU baseline is
        # the bach10_f16 leg's (same clip, same float64 oracle)
        arch_name, Nf, ov, window, f16, batch, seed = "bach10", 4096, 25, dcs.blackmanharris, False, 32, 3
        Lc = 441000
        audio = synth_audio(Lc, seed=0)
        library, melody = False, None
        what = "Bach10 4-instrument separate_bach10 path, f32-class arithmetic throughout (the 1e-4 variant of configs[3]): frameSize=4096 hop=512 blackmanharris, overlap 25, 10 s mono"
    elif name == "score_informed":
        # BASELINE configs[4]: score-conditioned masks, batch=128 (4-channel input [128,4,30,2049])
        arch_name, Nf, ov, window, f16, batch, seed = "bach10_si", 4096, 25, dcs.blackmanharris, False, 128, 5
        Lc = samples_for_tiles(128, library=True)
        audio = synth_audio(Lc, seed=4)
        library = True
        from deepconvsep_amd import score
        tmp = tempfile.mkdtemp(prefix="dcs_score_")
        files = []
        for i in range(4):
            files.append("inst%d.txt" % i)
            with open(os.path.join(tmp, files[-1]), "w") as fh:
                fh.write(synth_score_text(40 + i, Lc / float(SR) + 0.5, 40 + 5 * i, 64 + 6 * i))
        nframes = int(np.ceil(Lc / float(HOP))) + 2
        melody = score.melody_table(files, tmp, nframes, SR, HOP, Nf)
        what = "bach10_scoreinformed path (BASELINE configs[4]): frameSize=4096 hop=512 blackmanharris, 4 score channels, one batch of 128 library tiles (7.67 s), f32"
    else:
        raise ValueError("unknown leg %r" % name)

    arch = ARCHS[arch_name]
    Fb = Nf // 2 + 1
    params = synth_params(arch_name, TC, Fb, seed=seed)
    sep = dcs.Separator(arch_name, params, SCALE, TC, ov, batch, Fb, Nf, HOP, window,
                        tiler='library' if library else 'script')
    if f16:
        sep.net.set_conv_precision('f16')
    tiler = TILER_LIBRARY if library else TILER_SCRIPT
    Tfr = _lib.frame_count(Lc, HOP)
    n = _lib.tile_count(Tfr, TC, ov, tiler)
    a = ctx.to_device(audio, np.float32)
    out = torch.empty((arch.S, Lc), dtype=torch.float32, device=a.device)

    def run():
        if melody is not None:
            return sep.separate_scoreinformed_device(a, melody)
        return sep.net.separate(sep.plan, a, ov, tiler, SCALE, out=out)

    for _ in range(2):
        run()
    torch.cuda.synchronize()
    per, tot = [], 0.0
    while tot < 0.25 or len(per) < 3:
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        per.append(time.perf_counter() - t0)
        tot += per[-1]
    sec = statistics.median(per)
    ctx.timing("all")
    ctx.timing_reset()
    reps = 3
    for _ in range(reps):
        run()
    k_ms, k_launch = {}, {}
    for tag in _lib.TAGS:
        ms, cnt = ctx.timing_query(tag)
        if cnt:
            k_ms[tag] = round(ms * cnt / reps, 5)
            k_launch[tag] = cnt // reps
    ctx.timing(None)
    ctx.timing_reset()
    # dominant kernel among the network's kernels; priced per launch (a clip may go through in several chunks /
    # batches: launches-per-clip launches of n / launches tiles each)
    work_tags = [t for t in ("conv1", "conv2", "fc", "fc1x", "deconv2", "final", "decoder") if t in k_ms]
    dom = max(work_tags, key=lambda t: k_ms[t])
    per_tag_launches = {"fc1x": arch.live_branches()}       # one GEMM per live branch
    calls = max(1, k_launch[dom] // per_tag_launches.get(dom, 1))   # network passes per clip
    tiles_per_pass = n / float(calls)
    work = arch_work(arch, TC, Fb, tiles_per_pass, f16)
    flops, nbytes, peak, kind = work[dom]
    roofline = kernel_roofline(dom, k_ms[dom] / calls, flops, nbytes, peak, KERNEL_NAMES[dom], kind)
    roofline["tiles_per_launch"] = round(tiles_per_pass, 1)
    roofline["timed"] = "HIP events around every kernel of %d instrumented whole-path passes (separate from the timed passes)" % reps
    leg_traffic = load_traffic().get("legs", {}).get(name, {})

    def leg_record(tag):
        """Counter record (per launch) of the kernel that carries `tag` in this leg: the matching kernel with the most bytes."""
        sub = LEG_KERNELS.get(name, {}).get(tag)
        pick = "max"
        if isinstance(sub, tuple):
            sub, pick = sub
        best = None
        for kname, rec in leg_traffic.items():
            if sub and kname.startswith(sub):          # a prefix, not a substring: 'deconv1_mfma_kernel' contains 'conv1_mfma_kernel'
                b = traffic_bytes(rec)
                if b and (best is None or (b > best[0] if pick == "max" else b < best[0])):
                    best = (b, kname)
        return best

    def issued_block(tag, ach_tf):
        if tag not in LEG_ISSUED.get(name, {}):
            return None
        prod, pad = LEG_ISSUED[name][tag]
        iss = ach_tf * prod * pad
        return {"achieved": round(iss, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s (16-bit MFMA)",
                "frac": round(iss / PEAK_F16_TFLOPS, 4),
                "note": "%d 16-bit products per f32 product, K padded x%.3f" % (prod, pad)}

    def finish(r, tag):
        tr = leg_record(tag)
        if tr:
            r["traffic"] = tr[0]
            r["traffic_ratio"] = round(tr[0] / float(r["algorithmic_bytes"]), 3)
            r["traffic_source"] = "%s legs/%s/%s (rocprofv3 --pmc passes of this leg, 2*FETCH+WRITE per launch)" % (TRAFFIC_FILE, name, tr[1])
        if r["bound"] in ("mfma", "valu"):
            ach_f32 = r["algorithmic_flops"] / (r["avg_kernel_ms"] * 1e-3) / 1e12 if r["avg_kernel_ms"] > 0 else 0.0
            ib = issued_block(tag, ach_f32)
            if ib:
                r["issued"] = ib
                r["note"] = ("runs on the 16-bit matrix pipe (operands split into three bf16 terms, or f16 inputs): `frac` prices "
                             "the reference's f32 flop count against `peak`, `issued` the products actually issued against 2.5 PFLOP/s")
        return r

    roofline = finish(roofline, dom)
    table = {}
    for t in work_tags:
        fl, by, pk, kd = arch_work(arch, TC, Fb, n / float(max(1, k_launch[t] // per_tag_launches.get(t, 1))), f16)[t]
        r = finish(kernel_roofline(t, k_ms[t] / max(1, k_launch[t] // per_tag_launches.get(t, 1)), fl, by, pk, KERNEL_NAMES[t], kd), t)
        table[t] = {"ms_per_clip": k_ms[t], "bound": r["bound"], "frac": r["frac"], "peak": r["peak"], "unit": r["unit"]}
        if r.get("issued"):
            table[t]["issued_frac_of_16bit_peak"] = r["issued"]["frac"]
        if r.get("traffic"):
            table[t]["traffic_ratio"] = r["traffic_ratio"]
    frames = Tfr
    total_flops = n * arch.flops_per_tile(TC, Fb, live_only=True)
    res = {"workload": what, "tiles": int(n), "frames": int(frames), "audio_seconds": round(Lc / float(SR), 2),
           "batch_size": batch, "ms_per_clip": round(sec * 1e3, 4), "rounds": len(per),
           "value": round(frames / sec, 1), "unit": "frames/s", "x_realtime": round(Lc / float(SR) / sec, 1),
           "dtype": "f16 conv2 inputs / f32 accumulate, f32 elsewhere" if f16 else "f32",
           "kernels_ms": k_ms, "kernels_ms_sum": round(sum(k_ms.values()), 5), "kernel_rooflines": table,
           "whole_path_algorithmic_tflops": round(total_flops / sec / 1e12, 2),
           "roofline": roofline}
    wp = total_flops / sec / 1e12
    if wp <= PEAK_F32_TFLOPS and not f16:
        res["whole_path_frac_of_f32_peak"] = round(wp / PEAK_F32_TFLOPS, 4)
    else:
        # part of this path runs on the 16-bit matrix pipe: the f32 peak is not the governing one for the whole path
        res["whole_path_frac_of_16bit_peak"] = round(wp / PEAK_F16_TFLOPS, 4)
        res["whole_path_note"] = ("reference-counted f32 flops; conv2 / conv2^T (and the dense layers' bf16x3 GEMMs) execute on the "
                                  "16-bit matrix pipe, so the 157.3 TFLOP/s f32 peak does not bound this path -- per-kernel fractions "
                                  "are in kernel_rooflines")
    if with_cpu:
        from oracle import pipeline, tiling_np
        # bounded sample of the same input: the first ~32 tiles (the reference's batch), about 10 s of CPU work
        n_s = min(int(n), 32)
        Ls = min(Lc, samples_for_tiles(n_s, ov=ov, library=library))
        if name == "ikala":        # BASELINE configs[0] IS the CPU reference on the whole 10 s wav: no sampling
            n_s, Ls = int(n), Lc
        audio_s = np.asarray(audio[:Ls], dtype=np.float64)
        T_s = int(np.ceil(Ls / float(HOP))) + 2
        last = [None]
        if melody is not None:
            mel_s = melody.copy()

            def run_cpu():
                last[0] = pipeline.separate_scoreinformed(params, audio_s, mel_s, SCALE, TC, ov, 32, Nf, HOP, window)
        else:
            def run_cpu():
                last[0] = pipeline.separate(arch_name, params, audio_s, SCALE, TC, ov, 32, Nf, HOP, window,
                                            tiler=tiling_np.LIBRARY if library else tiling_np.SCRIPT)
        key = (arch_name, Nf, ov, seed, Ls)
        if key in _ORACLE_CACHE:
            # the float64 oracle does not depend on the conv-precision switch: the two Bach10 legs share clip, weights and
            # therefore CPU timing and oracle output
            last[0], cached = _ORACLE_CACHE[key]
            res["cpu_baseline"] = dict(cached, shared_with="the leg measured first on the same clip and weights (same float64 oracle run)")
        else:
            nt = min(16, ncpu)
            torch.set_num_threads(nt)
            sec_c, reps_c, el_c = time_cpu(run_cpu, 6.0, 50)
            torch.set_num_threads(1)
            sec_1, reps_1, el_1 = time_cpu(run_cpu, 4.0 if name != "ikala" else 0.5, 50)
            torch.set_num_threads(nt)
            res["cpu_baseline"] = {"value": round(T_s / sec_c, 1), "unit": "frames/s", "cores": int(nt), "kind": "port",
                                   "single_thread": {"value": round(T_s / sec_1, 1), "cores": 1},
                                   "cpu_model": cpu_name, "host_cpu_count": int(ncpu),
                                   "label": "reference-equivalent CPU path (Theano unavailable)",
                                   "sample": ("the WHOLE clip: " if Ls == Lc else "a bounded sample: ") +
                                             "the first %.2f s (%d tiles, %d frames) of the same clip through the oracle "
                                             "(reference NumPy loops + torch-CPU float64 network), %d x at %d threads "
                                             "(%.1f s) and %d x at 1 thread (%.1f s)"
                                             % (Ls / float(SR), n_s, T_s, reps_c, nt, el_c, reps_1, el_1)}
            _ORACLE_CACHE[key] = (last[0], res["cpu_baseline"])
        if with_parity and last[0] is not None:
            # the leg's own parity check: the PCM of the HIP path on the SAME sample (the whole clip for iKala), outside
            # every timed region, against the oracle output the CPU baseline just produced
            if melody is not None:
                got = sep.separate_scoreinformed(audio_s, mel_s)
            else:
                got = sep.separate(audio_s)
            err = float(np.max(np.abs(np.asarray(got, dtype=np.float64) - np.asarray(last[0], dtype=np.float64))))
            tol = 2e-3 if f16 else 1e-4
            # ... and the TIMED launch shape itself (the whole clip in one pass: other kernel variants are selected from 128
            # tiles on, e.g. the channels-last dense output + fused decoder of the Bach10 f16 leg): its PCM on the stretch of
            # the sample that no tile reaching past the sample's end touches -- there the two inputs give identical frames
            err_full, n_cmp = None, 0
            if Ls < Lc:
                n_cmp = int(Ls - (TC + 10) * HOP - Nf)
                if n_cmp > 0:
                    full = ctx.to_host(run()).astype(np.float64)
                    err_full = float(np.max(np.abs(full[:, :n_cmp] - np.asarray(last[0], dtype=np.float64)[:, :n_cmp])))
                    err = max(err, err_full)
            res["parity_check"] = {"max_abs_pcm_err": err, "tolerance": tol, "ok": bool(np.isfinite(err) and err < tol),
                                   "samples": int(Ls), "whole_clip": bool(Ls == Lc),
                                   "timed_launch_shape": ({"max_abs_pcm_err": err_full, "samples_compared": n_cmp,
                                                           "note": "PCM of the whole-clip launch (the shape that is timed) against the "
                                                                   "oracle's output for the sample, on the samples whose frames and tiles "
                                                                   "are the same in both"} if err_full is not None else None),
                                   "vs": "the oracle output of this leg's cpu_baseline run (reference NumPy STFT / tiling / "
                                         "overlap-add + float64 network) on the same samples",
                                   "tolerance_note": ("the stated bound of the f16-input conv path (tests/test_gpu_configs.py::"
                                                      "test_bach10_f16_conv_path_at_full_size_stated_tolerance); PCM measures ~1e-5"
                                                      if f16 else "north_star: 1e-4 fp32")}
            if not f16 and melody is None:
                # per mask bin (single-channel f32-class legs): the first tiles of the clip through the network operators
                from oracle import stft_np
                n_b = min(int(n_s), 8)
                L_b = min(Ls, samples_for_tiles(n_b, ov=ov, library=library) + HOP)
                mag_b = SCALE * stft_np.compute_file(audio_s[:L_b], phase=False, frameSize=Nf, hopSize=HOP, window=window).astype(np.float32)
                fb, nch = tiling_np.generate_overlapadd(mag_b, mag_b.shape[-1], TC, ov, 32,
                                                        tiler=tiling_np.LIBRARY if library else tiling_np.SCRIPT, fill=0.0)
                tiles_b = fb.reshape((-1,) + fb.shape[2:])[:min(nch, n_b)]
                bins = mask_bin_report(sep.net, sep.net.arch.name, params, tiles_b, 'A' if arch.eps_mode == 0 else 'B')
                res["parity_check"].update(mask_bins=bins["mask_bins"], masked_bins_outside_1e4=bins["masked_bins_outside_1e4"],
                                           conditioned_fraction=bins["conditioned_fraction"],
                                           network_output_max_err=bins["network_output_max_err"], mask_bin_check=bins)
                res["parity_check"]["mask_criterion_ok"] = bins["criterion_ok"]
                res["parity_check"]["ok"] = bool(res["parity_check"]["ok"] and bins["criterion_ok"] and
                                                 bins["network_output_max_err"] < 1e-4)
    del sep, a, out, params
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
