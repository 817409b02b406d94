#!/usr/bin/env python3
"""Summarise the --pmc passes of scripts/gpu_counters.sh: per (kernel, grid size) averages per launch, HBM traffic
(traffic.json: FETCH_SIZE / WRITE_SIZE in KiB; 2*FETCH + WRITE per the gfx950 note of MI355X_MICROARCH.md) and derived
figures (effective clock, matrix-pipe busy share, instructions per wave)."""
import collections
import csv
import glob
import json
import os
import re
import sys

KEYS = ("lat_final_kernel", "lat_gemm_kernel", "lat_deconv2_kernel", "lat_stft_kernel", "lat_ifft_kernel", "lat_ola_kernel",
        "lat_mid_kernel", "lat_stft_conv1_kernel", "final_bf16x3_kernel", "deconv2_stream_bf16_kernel", "g_split_kernel", "istft_seq_kernel", "gemm_bf16x3_skinny_kernel",
        "gemm_bf16x3_kernel", "gemm_pack_bq_kernel", "slabconv_ps_kernel", "slabconv_mx_kernel", "colconv_deconv1_fused_kernel",
        "colconv_wreg_gather_kernel", "colconv_wreg_scatter_kernel", "conv1_mfma_kernel", "deconv1_mfma_kernel",
        "conv1_reg_kernel", "mask_ola_kernel", "final_kernel", "deconv2_stream_kernel",
        "deconv2_kernel", "istft_wave_kernel", "istft_fused_kernel", "gemm_rows_splitk_kernel", "gemm_ksplit_reduce_kernel",
        "gemm_rows_kernel", "stft_forward_wave_kernel", "stft_forward_kernel", "slabconv_kernel", "colconv_f16_kernel",
        "colconv_kernel", "deconv1_reg_kernel", "deconv1_kernel", "conv1_kernel", "unpool_kernel", "pool_kernel",
        "mask_kernel", "overlap_add_kernel", "tile_kernel", "score_rect_kernel", "score_floor_kernel", "pcm_int16_kernel")


def short(name):
    for k in KEYS:
        if k in name:
            return k
    return re.sub(r"\(.*", "", name)[:40]


def load(out, name):
    fs = glob.glob(os.path.join(out, "pmc_" + name, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in fs:
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                agg[key]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
    fetch, write = load(out, "fetch"), load(out, "write")
    traffic = {}
    for k in sorted(set(fetch) | set(write)):
        rec = {}
        if k in fetch and "FETCH_SIZE" in fetch[k]:
            rec["FETCH_SIZE_KiB"] = round(fetch[k]["FETCH_SIZE"], 2)
        if k in write and "WRITE_SIZE" in write[k]:
            rec["WRITE_SIZE_KiB"] = round(write[k]["WRITE_SIZE"], 2)
        traffic["%s@grid_threads=%d" % k] = rec
    doc = {"all": traffic}
    for name in ("final_bf16x3_kernel", "final_kernel"):
        grids = sorted(g for (kk, g) in set(fetch) | set(write) if kk == name)
        for g in grids:
            rec = traffic["%s@grid_threads=%d" % (name, g)]
            # grid threads = 256 * row groups * column groups (* clips): 9 column groups of 128 bins at F = 1025
            doc.setdefault("by_grid", {})["%s@%d" % (name, g)] = rec
    try:
        cfg = json.loads(open(os.path.join(out, "pmc_fetch.json")).read().strip().splitlines()[-1])
        doc["bench_line_of_the_fetch_pass"] = {k: cfg[k] for k in ("value", "ms_per_step", "config") if k in cfg}
    except Exception:
        pass
    doc["note"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (scripts/gpu_counters.sh) of "
                   "bench.py --steps 32 --streams 1 --sat-tiles 4096; per launch, KiB, averaged over the launches of the same "
                   "kernel and grid.  On gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads: bench.py reports "
                   "2*FETCH + WRITE.")
    # the records bench.py looks up
    def pick(name, tiles_clips):
        for (kk, g), _ in sorted(fetch.items()):
            pass
    fin = sorted(g for (kk, g) in fetch if kk == "final_bf16x3_kernel")
    if fin:
        doc["final_kernel_4096_tiles"] = traffic["final_bf16x3_kernel@grid_threads=%d" % fin[-1]]
        if len(fin) >= 2:
            # the launch group of the counters run: bench.py looks the record up as final_kernel_<batches>x<tiles>_tiles
            g0 = 16
            try:
                g0 = int(cfg["config"]["launch_groups_per_round"][0])
            except Exception:
                pass
            doc["final_kernel_%dx32_tiles" % g0] = traffic["final_bf16x3_kernel@grid_threads=%d" % fin[0]]
    small = sorted(g for (kk, g) in fetch if kk == "final_kernel")
    if small:
        doc["final_kernel_32_tiles"] = traffic["final_kernel@grid_threads=%d" % small[0]]
    json.dump(doc, open(os.path.join(out, "traffic.json"), "w"), indent=1)

    waves, insts = load(out, "waves"), load(out, "insts")
    print("kernel, grid threads | dur us | clock GHz | waves | per wave: VALU SALU LDS VMEM MFMA insts | busy: inst-active/wave-cycles, "
          "wait-inst/wave-cycles | MFMA-busy share of SIMD cycles | LDS bank-conflict cycles / LDS-active | FETCH KiB, WRITE KiB")
    for k in sorted(set(waves) | set(insts)):
        w, i = waves.get(k, {}), insts.get(k, {})
        dur = w.get("dur_us") or i.get("dur_us") or 0.0
        gui = w.get("GRBM_GUI_ACTIVE", 0.0)
        nw = max(w.get("SQ_WAVES", 0.0), 1.0)
        clk = gui / 8.0 / (dur * 1e3) if dur else 0.0          # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        wc = max(w.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        cu_cycles = i.get("GRBM_GUI_ACTIVE", gui) / 8.0
        mfma_share = 100.0 * i.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(cu_cycles * 256 * 4, 1.0)
        f, wr = fetch.get(k, {}).get("FETCH_SIZE", float("nan")), write.get(k, {}).get("WRITE_SIZE", float("nan"))
        print("%-28s %9d | %8.1f | %.2f | %8.0f | %7.1f %7.1f %6.1f %6.1f %6.1f | %.3f %.3f | %5.1f%% | %.3f | %.0f %.0f" % (
            k[0], k[1], dur, clk, nw, w.get("SQ_INSTS_VALU", 0) / nw, w.get("SQ_INSTS_SALU", 0) / nw,
            i.get("SQ_INSTS_LDS", 0) / nw, i.get("SQ_INSTS_VMEM", 0) / nw, i.get("SQ_INSTS_MFMA", 0) / nw,
            w.get("SQ_ACTIVE_INST_ANY", 0) / wc, w.get("SQ_WAIT_INST_ANY", 0) / wc, mfma_share,
            i.get("SQ_LDS_BANK_CONFLICT", 0) / max(i.get("SQ_ACTIVE_INST_LDS", 0), 1.0), f, wr))
    return 0


if __name__ == "__main__":
    sys.exit(main())
