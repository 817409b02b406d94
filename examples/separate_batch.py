#!/usr/bin/env python3
"""Batch-of-files driver: what examples/dsd100/separate_multiple.ipynb of the reference does with
``os.system("python separate_dsd.py -i ... -o ... -m ...")`` per song, with the model resident on the GPU.

    python separate_batch.py -a dsd -m model.pkl -o outdir  a.wav b.wav ...
    python -m torch.distributed.run --nproc-per-node 8 separate_batch.py -a dsd -m model.pkl -o outdir *.wav

Every file gets its own sub-directory ``outdir/<stem>/`` holding the wav files the single-file script
writes.  With several ranks (one process per GPU) the files are dealt round-robin: replicas only, no
collective on the data path.  While the GPU separates file i the host reads file i+1 and writes file i-1
(two worker threads), so wav I/O overlaps the kernels.  ``--group G`` (default 8) takes G files at a time; those
of equal length share one set of kernel launches.
"""
import argparse
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-a", "--arch", default="dsd", choices=["dsd", "hiphop", "ikala", "bach10"])
    ap.add_argument("-m", "--mfile", required=True)
    ap.add_argument("-o", "--odir", required=True)
    ap.add_argument("-g", "--group", type=int, default=8, help="files read ahead and separated together")
    ap.add_argument("--stats", action="store_true",
                    help="print one JSON line at the end: files, seconds from model-ready to the last wav written, ms per file")
    ap.add_argument("files", nargs="+")
    args = ap.parse_args(argv)
    import time

    import torch
    from deepconvsep_amd import separation as sp

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    frame, hop, window, overlap, bins = sp._SCRIPT_DEFAULTS[args.arch]
    sep = sp.Separator(args.arch, sp.load_model(args.mfile), 0.3, 30, overlap, 32, bins, frame, hop, window)
    mine = args.files[rank::world]
    torch.cuda.synchronize()
    t_ready = time.perf_counter()          # interpreter, torch import, model upload and plan are behind us

    failed = []

    def read(path):
        # a file that cannot be read or mixed down fails alone (the notebook runs one process per file)
        try:
            sr, audio = sp.read_wav(path)
            return path, sr, (sp.to_mono(audio, args.arch) if sr == 44100 else None)
        except Exception as exc:
            return path, None, exc

    def write(path, sr, pcm):
        out = os.path.join(args.odir, os.path.splitext(os.path.basename(path))[0])
        os.makedirs(out, exist_ok=True)
        for dst, sig in zip(sp.output_paths(args.arch, path, out), pcm):
            sp.write_wav(dst, sig, sr)

    # --group G: G files are read ahead and separated together; those of equal length (a dataset cut into fixed
    # excerpts) share one set of kernel launches (dcs_separate_batch), the rest go one by one
    G = max(1, args.group)
    with ThreadPoolExecutor(max_workers=2) as pool:
        chunks = [mine[i:i + G] for i in range(0, len(mine), G)]
        nxt = [pool.submit(read, f) for f in chunks[0]] if chunks else []
        pending = []
        for ci in range(len(chunks)):
            got = [f.result() for f in nxt]
            nxt = [pool.submit(read, f) for f in chunks[ci + 1]] if ci + 1 < len(chunks) else []
            ok = []
            for path, sr, audio in got:
                if isinstance(audio, Exception):
                    failed.append((path, audio))
                elif audio is None:
                    print("Sample rate is not 44100")          # separate_dsd.py:313
                else:
                    ok.append((path, sr, audio))
            pcms = sep.separate_many([a for _, _, a in ok], on_error='return')
            for f in pending:
                f.result()
            pending = []
            for (path, sr, _), pcm in zip(ok, pcms):
                if isinstance(pcm, Exception):
                    failed.append((path, pcm))
                else:
                    pending.append(pool.submit(write, path, sr, pcm))
        for f in pending:
            f.result()
    for path, exc in failed:
        print("%s: %s: %s" % (path, type(exc).__name__, exc), file=sys.stderr)
    if args.stats:
        import json
        el = time.perf_counter() - t_ready
        print(json.dumps({"rank": rank, "files": len(mine), "failed": len(failed), "seconds_after_model_ready": round(el, 4),
                          "ms_per_file": round(el / max(1, len(mine)) * 1e3, 3), "group": G}))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
