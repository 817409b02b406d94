"""The last stdout line of bench.py must be something the driver can parse: compact (< 4 KB), strict JSON, carrying the
contract keys plus `roofline` and `cpu_baseline`.  (Round 4's single 22 KB line was cut by the driver's stdout tail:
BENCH_r04.json.parsed = null.)  The canned full result is round 4's own line, tests/data/bench_full_line_r04.json."""
import io
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_check"]


def _full():
    with open(os.path.join(ROOT, "tests", "data", "bench_full_line_r04.json")) as fh:
        return json.load(fh)


def _no_constants(s):
    raise ValueError("non-strict JSON constant %r" % s)


def test_headline_is_compact_strict_json_with_the_contract_keys():
    full = _full()
    assert len(json.dumps(full)) > 20000                       # the thing that did not parse
    text = bench.headline(full)
    assert "\n" not in text and len(text.encode()) < bench.HEADLINE_LIMIT
    h = json.loads(text, parse_constant=_no_constants)
    for k in CONTRACT:
        assert k in h, k
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "scaling"):
        assert h[k] == full[k]
    r = h["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio", "algorithmic_bytes",
              "avg_kernel_ms", "tiles_per_launch"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["issued"]["frac"] == full["roofline"]["issued"]["frac"]
    c = h["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "cpu_model", "host_cpu_count", "sample"):
        assert k in c, k
    assert h["parity_check"]["ok"] is True and h["parity_check"]["tolerance"] == 1e-4
    assert "workload" in h["config"] and "model" not in h["config"]
    assert h["single_stream_ms_per_step"] == full["single_stream_ms_per_step"]
    assert h["whole_path_frac_of_f32_peak"] == full["whole_path_frac_of_f32_peak"]
    for big in ("legs", "cli", "saturating", "host_fed", "single_stream", "launch_group"):
        assert big not in h


def test_headline_carries_the_hbm_stage_fractions_and_says_when_a_traffic_record_is_missing():
    """north_star "achieved HBM GB/s on STFT / mask stages": the STFT / iSTFT fractions of the 8 TB/s peak ride in the
    driver-parsed line (< 200 bytes), and a roofline without a committed counter record of its grid says which key missed."""
    full = _full()
    full["hbm_stages"] = {"stft": {"frac": 0.2171, "GBps": 1736.9, "traffic_ratio": 1.587, "ms": 0.02},
                          "istft": {"frac": 0.2927, "GBps": 2341.4, "traffic_ratio": None}}
    full["roofline"]["traffic"] = None
    full["roofline"]["traffic_ratio"] = None
    full["roofline"]["traffic_key_missing"] = "final_bf16x3_kernel@grid_threads=552960"
    text = bench.headline(full)
    h = json.loads(text, parse_constant=_no_constants)
    assert len(text.encode()) < bench.HEADLINE_LIMIT
    assert h["hbm_stages"] == {"stft": {"frac": 0.2171, "GBps": 1736.9, "traffic_ratio": 1.587},
                               "istft": {"frac": 0.2927, "GBps": 2341.4, "traffic_ratio": None}}
    assert len(json.dumps(h["hbm_stages"])) < 200
    assert h["roofline"]["traffic"] is None and h["roofline"]["traffic_key_missing"].startswith("final_bf16x3_kernel@")


def test_headline_of_an_n_gpu_line_carries_the_gather_split_and_stays_small():
    full = _full()
    full.update(n_gpus=8, cpu_baseline=None, legs=None, cli=None, host_fed=None, saturating=None, gather_check="ok")
    full["gather"] = {"mode": "allgather", "impl": "dcs_gather (C ABI, RCCL communicator per HIP stream)",
                      "payload_bytes_per_rank_per_group": 15073280, "round_ms_without_gather": 0.2, "round_ms_with_gather": 0.3,
                      "ms_per_group_collective_alone": 0.09, "note": "x" * 400}
    full["parity_check"]["ranks"] = 8
    text = bench.headline(full)
    h = json.loads(text, parse_constant=_no_constants)
    assert len(text.encode()) < bench.HEADLINE_LIMIT
    assert h["n_gpus"] == 8 and h["cpu_baseline"] is None and h["gather_check"] == "ok"
    g = h["gather"]
    assert g["mode"] == "allgather" and g["impl"].startswith("dcs_gather") and "note" not in g
    assert len(json.dumps(g)) < 1024
    assert g["round_ms_with_gather"] == 0.3 and g["round_ms_without_gather"] == 0.2


def test_headline_never_emits_nan_or_infinity_and_survives_oversized_strings():
    full = _full()
    full["roofline"]["frac"] = float("nan")
    full["value"] = float("inf")
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["roofline"]["kernel"] = "k" * 5000
    full["config"]["weights"] = "g" * 5000                      # an unexpectedly long extra field: the last-resort cut
    text = bench.headline(full)
    assert len(text.encode()) < bench.HEADLINE_LIMIT
    h = json.loads(text, parse_constant=_no_constants)
    assert h["roofline"]["frac"] is None and h["value"] is None


def test_emit_prints_the_headline_last_and_writes_the_detail(tmp_path, monkeypatch):
    full = _full()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    out = io.StringIO()
    bench.emit(full, detail_stdout=True, out=out)
    lines = out.getvalue().splitlines()
    assert len(lines) == 2
    assert json.loads(lines[0])["legs"].keys() == full["legs"].keys()      # the detail comes first ...
    assert lines[1] == bench.headline(full)                                # ... the compact line last
    with open(os.path.join(str(tmp_path), bench.DETAIL_FILE)) as fh:
        assert json.load(fh)["cli"] == full["cli"]
    out = io.StringIO()
    bench.emit(full, out=out)
    assert out.getvalue().splitlines() == [bench.headline(full)]           # default: ONE stdout line


def test_round_plan_pipelines_the_gather_only_where_it_can():
    """bench.py's schedule of a round: the driver's --steps 20 is ONE launch group on one GPU; with N > 1 it becomes two half-groups
    whose collectives overlap the other half's compute -- not without a collective, not on one lane, not with --no-pipeline, and
    rounds that already have several groups keep them."""
    assert bench.plan_round(20, 32, 3, False) == ([20], False)
    assert bench.plan_round(20, 32, 3, True, "root") == ([10, 10], True)
    assert bench.plan_round(7, 32, 3, True, "allgather") == ([4, 3], True)
    assert bench.plan_round(20, 32, 3, True, "none") == ([20], False)
    assert bench.plan_round(20, 32, 1, True, "root") == ([20], False)
    assert bench.plan_round(20, 32, 3, True, "root", no_pipeline=True) == ([20], False)
    assert bench.plan_round(1, 32, 3, True, "root") == ([1], False)
    groups, piped = bench.plan_round(384, 32, 3, True, "root")
    assert sum(groups) == 384 and len(groups) == 12 and not piped
