"""Code-generation checks of the gfx950 kernels that need no GPU (hipcc cross-compiles)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_no_kernel_keeps_a_prefetch_array_in_scratch_memory():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_private_arrays.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 unexpected private arrays" in r.stdout


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_no_training_step_kernel_spills_registers():
    """A spilled register is reloaded through the in-order vector memory queue, behind the next tile's prefetch."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_spills.py")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 kernels above their spill allowance" in r.stdout
