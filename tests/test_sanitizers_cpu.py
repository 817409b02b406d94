"""Sanitizer runs of the HOST-side native code.  Device AddressSanitizer needs xnack+ code objects, which
this pool cannot run (the device kernels are covered by the guard-band harness of tests/test_gpu_guard.py);
the native code that runs THREADS on the host and writes into caller memory -- the wav I/O pool of the batch-of-files driver,
csrc/wavio.hip, pure host C++ -- is compiled here with g++ under -fsanitize=address,undefined and under -fsanitize=thread and
driven through the C ABI of include/dcs.h by tests/native/wavio_san_main.cpp (batches in flight at once, odd and damaged files, a
buffer one byte short, canaries behind every destination, late collection, the pool destroyed with work enqueued)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "deepconvsep_amd", "csrc", "wavio.hip"), os.path.join(ROOT, "tests", "native", "wavio_san_main.cpp"),
       os.path.join(ROOT, "tests", "native", "san_stub.cpp")]
HIP_INC = "/opt/rocm/include"


@pytest.mark.parametrize("name,flags,markers", [
    ("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], ["AddressSanitizer", "runtime error:", "LeakSanitizer"]),
    ("tsan", ["-fsanitize=thread"], ["ThreadSanitizer"]),
])
def test_wav_io_pool_under_sanitizers(name, flags, markers, tmp_path):
    if shutil.which("g++") is None or not os.path.isdir(HIP_INC):
        pytest.skip("g++ or the HIP headers are not here")
    exe = str(tmp_path / ("wavio_" + name))
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I" + HIP_INC] + flags + \
          ["-x", "c++"] + SRC + ["-lpthread", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and ("cannot find -l" in b.stderr or "unrecognized" in b.stderr):
        pytest.skip("this toolchain has no %s runtime: %s" % (name, b.stderr[-200:]))
    assert b.returncode == 0, b.stderr[-3000:]
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", TSAN_OPTIONS="halt_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, str(scratch)], capture_output=True, text=True, timeout=600, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    for m in markers:
        assert m not in out, out[-4000:]
    assert "late collection: ok" in r.stdout
