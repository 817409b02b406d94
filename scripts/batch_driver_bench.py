#!/usr/bin/env python3
"""The steady-state figure of bench.py's command-line leg on its own, over a few driver settings:
    python scripts/batch_driver_bench.py [--files 550] [--variants "-w 16 -g 16" "-w 8 -g 32" ...]
50 synthetic 10 s stereo int16 wavs (DSD as shipped: frameSize 1024, hop 512, 513 bins), listed 11 times; prints the
`--stats` line of examples/separate_batch.py per variant."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=550)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--variants", nargs="*", default=["-w 16 -g 16"])
    args = ap.parse_args()
    import deepconvsep_amd as dcs
    from deepconvsep_amd.synth import synth_audio, synth_params
    from deepconvsep_amd import separation as sp
    tmp = tempfile.mkdtemp(prefix="dcs_bd_")
    model = os.path.join(tmp, "model.pkl")
    dcs.save_model(model, synth_params("dsd", 30, 513, seed=2))
    wavs = []
    for i in range(50):
        w = os.path.join(tmp, "clip%02d.wav" % i)
        sp.write_wav(w, synth_audio(441000, seed=500 + i, channels=2), 44100)
        wavs.append(w)
    files = (wavs * (args.files // 50 + 1))[:args.files]
    for v in args.variants:
        for rep in range(args.reps):
            out = os.path.join(tmp, "out")
            shutil.rmtree(out, ignore_errors=True)
            os.makedirs(out)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "separate_batch.py"), "-a", "dsd", "-m", model, "-o", out,
                                "--stats"] + v.split() + files, capture_output=True, text=True, timeout=1800)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                print("%-28s FAILED rc %d: %s" % (v, r.returncode, (r.stderr or r.stdout)[-400:]))
                continue
            s = json.loads(line[-1])
            print("%-28s %.3f ms/file  after first group %s  main thread %s" % (
                v, s["ms_per_file"], s.get("ms_per_file_after_first_group"), json.dumps(s.get("main_thread_ms_per_file"))), flush=True)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
