#!/usr/bin/env python3
"""CPU-only look at what the compiler did with a translation unit's waits (round 4: three kernels lost 10 - 25 % to waits the
source does not show).  Compile with `hipcc --offload-arch=gfx950 -O3 -std=c++17 -c file.hip -save-temps=obj -o /tmp/x.o`, then

    python scripts/isa_wait_scan.py /tmp/file-hip-amdgcn-amd-amdhsa-gfx950.s [kernel-name-substring]

per kernel: (a) every `s_waitcnt vmcnt(N)` within a dozen instructions of the last global / buffer load -- a load whose
result is copied, selected or masked at once turns a prefetch into a blocking load; (b) how many MFMAs sit directly behind
an `s_waitcnt lgkmcnt(0)` with at most two LDS reads in front of it -- read / wait / multiply, one LDS latency per MFMA,
which nothing hides when a wave is alone on its SIMD."""
import re
import sys


def kernels(path, pat):
    lines = open(path).read().split("\n")
    for i, l in enumerate(lines):
        if re.match(r"^_Z[\w.]+:", l) and pat in l:
            name = l.split(":")[0]
            try:
                end = next(j for j in range(i, len(lines)) if lines[j].startswith("\t.set " + name + "."))
            except StopIteration:
                continue
            body = [x.strip() for x in lines[i:end] if x.startswith("\t") and not x.strip().startswith((";", "."))]
            yield name, body


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, body in kernels(path, pat):
        ops = [b.split()[0] for b in body]
        n_mfma = sum(o.startswith("v_mfma") for o in ops)
        print("== %s: %d instructions, %d MFMA, %d barriers" % (name[:100], len(body), n_mfma, ops.count("s_barrier")))
        last, nld = None, 0
        for k, ins in enumerate(body):
            if ops[k].startswith(("global_load", "buffer_load", "flat_load")):
                last, nld = k, nld + 1
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if ops[k] == "s_waitcnt" and m and last is not None and k - last <= 12:
                print("   (a) %-28s %2d instructions behind load #%d" % (ins, k - last, nld))
        tight = 0
        for k, o in enumerate(ops):
            if not o.startswith("v_mfma"):
                continue
            j = k - 1
            while j >= 0 and ops[j] in ("s_nop",):
                j -= 1
            if j >= 0 and ops[j] == "s_waitcnt" and "lgkmcnt(0)" in body[j]:
                reads = 0
                i = j - 1
                while i >= 0 and ops[i].startswith("ds_read"):
                    reads += 1
                    i -= 1
                if 1 <= reads <= 2:
                    tight += 1
        if n_mfma:
            print("   (b) %d of %d MFMAs directly behind `<= 2 LDS reads; s_waitcnt lgkmcnt(0)`" % (tight, n_mfma))


if __name__ == "__main__":
    main()
