#!/usr/bin/env python3
"""Batch-of-files driver: what examples/dsd100/separate_multiple.ipynb of the reference does with
``os.system("python separate_dsd.py -i ... -o ... -m ...")`` per song, with the model resident on the GPU.

    python separate_batch.py -a dsd -m model.pkl -o outdir  a.wav b.wav ...
    python -m torch.distributed.run --nproc-per-node 8 separate_batch.py -a dsd -m model.pkl -o outdir *.wav

Every file gets its own sub-directory ``outdir/<stem>/`` holding the wav files the single-file script
writes.  With several ranks (one process per GPU) the files are dealt round-robin: replicas only, no
collective on the data path.  ``--group G`` (default 16) takes G files at a time; they share sets of kernel launches
(``Separator.separate_many_pcm16``: equal lengths ``dcs_separate_batch``, different lengths ``dcs_separate_ragged``).

16-bit PCM files (the datasets' format) never become floats on the host: a worker thread reads a file's int16 frames
straight into a pinned staging buffer (``wavio.read_pcm16_into``), the frames go to the GPU as they are, the division by
32767, the mix-down (separate_dsd.py:278-287) and the int16 conversion of the results (:307-309) run on the device in the
scripts' float64 arithmetic, the int16 samples come back into pinned memory and worker threads write them behind the 44-byte
header ``scipy.io.wavfile.write`` would produce -- the output files are byte-identical to those of the float path.  Any other
sample format takes the float path (``read_wav`` / ``to_mono`` / ``separate_many`` / ``write_wav``) file by file.
While the GPU separates group i the workers read group i+1 and write group i-1.
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
    ap.add_argument("-g", "--group", type=int, default=16, help="files read ahead and separated together")
    ap.add_argument("-w", "--workers", type=int, default=0, help="I/O worker threads (default: min(16, cores))")
    ap.add_argument("--float-path", action="store_true",
                    help="force the float path of the single-file scripts (read_wav / to_mono on the host, float32 upload, "
                         "float64 download, scipy writes) also for 16-bit PCM files")
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
    from deepconvsep_amd import wavio
    n_workers = max(2, min(args.workers or 16, (os.cpu_count() or 2)))
    G = max(1, args.group)
    chunks = [mine[i:i + G] for i in range(0, len(mine), G)]
    # pinned input arenas, one per chunk in flight (the chunk being separated and the one being read)
    arenas = [None, None]

    def arena_for(k, nbytes):
        a = arenas[k % 2]
        if a is None or a.numel() < nbytes:
            a = arenas[k % 2] = torch.empty((max(nbytes, 1 << 22) * 5 // 4,), dtype=torch.uint8).pin_memory()
        return a

    def read(path, arena, off, cap):
        """16-bit PCM: (path, rate, pinned int16 tensor [L] / [L, ch]); otherwise the float path's mono signal."""
        try:
            got = wavio.read_pcm16_into(path, arena.numpy()[off:off + cap]) if not args.float_path else None
            if got is not None:
                sr, frames, ch = got
                t = arena[off:off + 2 * frames * ch].view(torch.int16)
                return path, sr, (t if ch == 1 else t.view(frames, ch)), True
            sr, audio = sp.read_wav(path)
            return path, sr, (sp.to_mono(audio, args.arch) if sr == 44100 else None), False
        except Exception as exc:      # a file that cannot be read or mixed down fails alone (the notebook: one process per file)
            return path, None, exc, False

    def submit_reads(pool, k):
        if k >= len(chunks):
            return []
        sizes = [os.path.getsize(f) if os.path.exists(f) else 0 for f in chunks[k]]
        offs = [0]
        for sz in sizes:
            offs.append(offs[-1] + (sz + 63) // 64 * 64)
        arena = arena_for(k, offs[-1])
        return [pool.submit(read, f, arena, offs[j], sizes[j]) for j, f in enumerate(chunks[k])]

    def out_dir(path):
        out = os.path.join(args.odir, os.path.splitext(os.path.basename(path))[0])
        os.makedirs(out, exist_ok=True)
        return out

    def write16(path, sr, pcm):
        for dst, sig in zip(sp.output_paths(args.arch, path, out_dir(path)), pcm):
            wavio.write_pcm16(dst, sr, sig)

    def write_float(path, sr, pcm):
        for dst, sig in zip(sp.output_paths(args.arch, path, out_dir(path)), pcm):
            sp.write_wav(dst, sig, sr)

    t_stage = {"wait_reads": 0.0, "separate": 0.0, "wait_writes": 0.0, "submit": 0.0}
    t_first = None
    with ThreadPoolExecutor(max_workers=n_workers) as pool:
        nxt = submit_reads(pool, 0)
        pending = []
        for ci in range(len(chunks)):
            t0 = time.perf_counter()
            got = [f.result() for f in nxt]
            t_stage["wait_reads"] += time.perf_counter() - t0
            nxt = submit_reads(pool, ci + 1)
            fast, slow = [], []
            for path, sr, audio, is16 in got:
                if isinstance(audio, Exception):
                    failed.append((path, audio))
                elif sr != 44100 or audio is None:
                    print("Sample rate is not 44100")          # separate_dsd.py:313
                else:
                    (fast if is16 else slow).append((path, sr, audio))
            t0 = time.perf_counter()
            res16 = sep.separate_many_pcm16([a for _, _, a in fast], max_group=max(16, G), on_error='return') if fast else []
            resf = sep.separate_many([a for _, _, a in slow], on_error='return') if slow else []
            t1 = time.perf_counter()
            for f in pending:                                   # the writes of the previous chunk (their pinned block is two
                f.result()                                      # calls old when it is reused: ring of three)
            t2 = time.perf_counter()
            t_stage["separate"] += t1 - t0
            t_stage["wait_writes"] += t2 - t1
            pending = []
            for (path, sr, _), pcm in zip(fast, res16):
                if isinstance(pcm, Exception):
                    failed.append((path, pcm))
                    continue
                pending.append(pool.submit(write16, path, sr, pcm))
            for (path, sr, _), pcm in zip(slow, resf):
                if isinstance(pcm, Exception):
                    failed.append((path, pcm))
                else:
                    pending.append(pool.submit(write_float, path, sr, pcm))
            t_stage["submit"] += time.perf_counter() - t2
            if ci == 0:
                t_first = time.perf_counter()          # the first group also pays for the pinned staging blocks and the workspace
        t0 = time.perf_counter()
        for f in pending:
            f.result()
        t_stage["wait_writes"] += time.perf_counter() - t0
    for path, exc in failed:
        print("%s: %s: %s" % (path, type(exc).__name__, exc), file=sys.stderr)
    if args.stats:
        import json
        el = time.perf_counter() - t_ready
        print(json.dumps({"rank": rank, "files": len(mine), "failed": len(failed), "seconds_after_model_ready": round(el, 4),
                          "ms_per_file": round(el / max(1, len(mine)) * 1e3, 3), "group": G, "workers": n_workers,
                          "ms_per_file_after_first_group": (round((time.perf_counter() - t_first) / (len(mine) - G) * 1e3, 3)
                                                            if t_first is not None and len(mine) > G else None),
                          "path": "float" if args.float_path else "int16 frames (device mix-down and int16 conversion)",
                          "main_thread_ms_per_file": {k: round(v / max(1, len(mine)) * 1e3, 3) for k, v in t_stage.items()}}))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
