"""Builds the host C++ layer (checkpoint format, control board) with AddressSanitizer + UBSan and
with ThreadSanitizer and runs its self-test (csrc/tests/host_selftest.cc): round trips, a corrupted
checkpoint, and the board's arrival / token / barrier protocols hammered from six threads under
uneven load."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = ["csrc/tests/host_selftest.cc", "csrc/ckpt/tensor_bundle.cc", "csrc/host/board.cc"]


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_host_layer_under_sanitizers(tmp_path, san):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "selftest")
    cmd = [gxx, "-std=c++17", "-O1", "-g", f"-fsanitize={san}", "-fno-omit-frame-pointer", "-o", exe] + \
          [os.path.join(ROOT, s) for s in SRCS] + ["-lpthread", "-lrt"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if b.returncode != 0 and "sanitize" in b.stderr:
        pytest.skip("sanitizer runtime not available: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0", TSAN_OPTIONS="halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "host selftest OK" in r.stdout
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr
