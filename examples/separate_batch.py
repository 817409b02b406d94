#!/usr/bin/env python3
"""Batch-of-files driver: what examples/dsd100/separate_multiple.ipynb of the reference does with
``os.system("python separate_dsd.py -i ... -o ... -m ...")`` per song, with the model resident on the GPU.

    python separate_batch.py -a dsd -m model.pkl -o outdir  a.wav b.wav ...
    python -m torch.distributed.run --nproc-per-node 8 separate_batch.py -a dsd -m model.pkl -o outdir *.wav

Every file gets its own sub-directory ``outdir/<stem>/`` holding the wav files the single-file script
writes.  With several ranks (one process per GPU) the files are dealt round-robin: replicas only, no
collective on the data path.  ``--group G`` (default 16) takes G files at a time; they share sets of kernel launches
(``Separator.separate_many_pcm16``: equal lengths ``dcs_separate_batch``, different lengths ``dcs_separate_ragged``).

16-bit PCM files (the datasets' format) never become floats on the host: the I/O threads of ``libdcs.so``
(``wavio.WavPool``: one call per group of files, no interpreter lock on the workers' side) read a file's int16 frames
straight into a pinned staging buffer, the frames go to the GPU as they are, the division by
32767, the mix-down (separate_dsd.py:278-287) and the int16 conversion of the results (:307-309) run on the device in the
scripts' float64 arithmetic, the int16 samples come back into pinned memory and the same threads write them behind the 44-byte
header ``scipy.io.wavfile.write`` would produce -- the output files are byte-identical to those of the float path.  Any other
sample format takes the float path (``read_wav`` / ``to_mono`` / ``separate_many`` / ``write_wav``) file by file.
While the GPU separates group i the workers read group i+1 and write group i-1, and the main thread does not wait for
the device either: a group is enqueued (``separate_many_pcm16(..., wait=False)``) and collected one iteration later.
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
    # pinned input arenas, one per chunk in flight (being read | enqueued, its upload possibly not yet executed | collected)
    arenas = [None, None, None]

    def arena_for(k, nbytes):
        a = arenas[k % 3]
        if a is None or a.numel() < nbytes:
            a = arenas[k % 3] = torch.empty((max(nbytes, 1 << 22) * 9 // 8,), dtype=torch.uint8, pin_memory=True)
        return a

    def read_float(path):
        """The scripts' own read (scipy) + mix-down: (path, rate, mono float signal | None | the exception)."""
        try:
            sr, audio = sp.read_wav(path)
            return path, sr, (sp.to_mono(audio, args.arch) if sr == 44100 else None)
        except Exception as exc:      # a file that cannot be read or mixed down fails alone (the notebook: one process per file)
            return path, None, exc

    def submit_reads(io, tp, k):
        """Chunk k's reads: a native batch into the chunk's pinned arena (16-bit PCM) or scipy reads on the Python pool."""
        if k >= len(chunks):
            return None
        files = chunks[k]
        if args.float_path:
            return ("float", [tp.submit(read_float, f) for f in files])
        sizes = []
        for f in files:
            try:
                sizes.append(os.path.getsize(f))
            except OSError:
                sizes.append(0)
        offs = [0]
        for sz in sizes:
            offs.append(offs[-1] + (sz + 63) // 64 * 64)
        arena = arena_for(k, offs[-1])
        flat = arena.numpy()
        return ("pcm16", io.read_into(files, [flat[offs[j]:offs[j] + sizes[j]] for j in range(len(files))]), arena, offs)

    def take_reads(pending_reads, ci):
        """-> (fast, slow): [(path, rate, pinned int16 tensor [L] / [L, ch])], [(path, rate, mono float signal)]."""
        fast, slow = [], []

        def classify(path, sr, audio):
            if isinstance(audio, Exception):
                failed.append((path, audio))
            elif sr != 44100 or audio is None:
                print("Sample rate is not 44100")          # separate_dsd.py:313
            else:
                slow.append((path, sr, audio))
        if pending_reads[0] == "float":
            for f in pending_reads[1]:
                classify(*f.result())
            return fast, slow
        _, batch, arena, offs = pending_reads
        for j, (path, got) in enumerate(zip(chunks[ci], batch.result())):
            if isinstance(got, Exception):
                failed.append((path, got))
            elif got is None:                                # not plain 16-bit PCM: the scripts' float path for this file
                classify(*read_float(path))
            elif got[0] != 44100:
                print("Sample rate is not 44100")
            else:
                sr, frames, ch = got
                t = arena[offs[j]:offs[j] + 2 * frames * ch].view(torch.int16)
                fast.append((path, sr, t if ch == 1 else t.view(frames, ch)))
        return fast, slow

    def out_dir(path):
        return os.path.join(args.odir, os.path.splitext(os.path.basename(path))[0])

    def write_float(path, sr, pcm):
        os.makedirs(out_dir(path), exist_ok=True)
        for dst, sig in zip(sp.output_paths(args.arch, path, out_dir(path)), pcm):
            sp.write_wav(dst, sig, sr)

    t_stage = {"wait_reads": 0.0, "enqueue": 0.0, "wait_device": 0.0, "wait_writes": 0.0, "submit": 0.0}
    t_first = None
    now = time.perf_counter

    def finish_writes(pending):
        for kind, w, paths in pending:
            if kind == "pcm16":
                for path, err in zip(paths, w.result()):
                    if err is not None:
                        failed.append((path, err))
            else:
                try:
                    w.result()
                except Exception as exc:
                    failed.append((paths, exc))

    def collect(io, tp, inflight, pending):
        """The device's results of an enqueued chunk -> write batches.  Waits for the writes of the chunk before it first:
        their pinned block is the one the NEXT enqueue reuses (ring of three)."""
        fast, h16, slow, resf = inflight
        t0 = now()
        res16 = h16.result() if h16 is not None else []
        t1 = now()
        finish_writes(pending)
        t2 = now()
        t_stage["wait_device"] += t1 - t0
        t_stage["wait_writes"] += t2 - t1
        out = []
        paths, rates, arrays = [], [], []
        for (path, sr, _), pcm in zip(fast, res16):
            if isinstance(pcm, Exception):
                failed.append((path, pcm))
                continue
            for dst, sig in zip(sp.output_paths(args.arch, path, out_dir(path)), pcm):
                paths.append(dst)
                rates.append(sr)
                arrays.append(sig)
        if paths:
            out.append(("pcm16", io.write(paths, rates, arrays), paths))        # one call; the directories are made by the pool
        for (path, sr, _), pcm in zip(slow, resf):
            if isinstance(pcm, Exception):
                failed.append((path, pcm))
            else:
                out.append(("float", tp.submit(write_float, path, sr, pcm), path))
        t_stage["submit"] += now() - t2
        return out

    with wavio.WavPool(n_workers) as io, ThreadPoolExecutor(max_workers=min(n_workers, 8)) as tp:
        nxt = submit_reads(io, tp, 0)
        pending, inflight = [], None
        for ci in range(len(chunks)):
            t0 = now()
            fast, slow = take_reads(nxt, ci)
            t_stage["wait_reads"] += now() - t0
            t0 = now()
            h16 = (sep.separate_many_pcm16([a for _, _, a in fast], max_group=max(16, G), on_error='return', wait=False)
                   if fast else None)
            resf = sep.separate_many([a for _, _, a in slow], on_error='return') if slow else []
            t_stage["enqueue"] += now() - t0
            # the arena of chunk ci + 1 was last used by chunk ci - 2, collected in the previous iteration
            t0 = now()
            nxt = submit_reads(io, tp, ci + 1)
            t_stage["submit"] += now() - t0
            if inflight is not None:
                pending = collect(io, tp, inflight, pending)
            inflight = (fast, h16, slow, resf)
            if ci == 0:
                t_first = now()          # the first group also pays for the pinned staging blocks and the workspace
        if inflight is not None:
            pending = collect(io, tp, inflight, pending)
        t0 = now()
        finish_writes(pending)
        t_stage["wait_writes"] += now() - t0
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
