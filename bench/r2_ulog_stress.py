#!/usr/bin/env python3
"""Longer stress of the device update log: W replicas sharing one GPU, M-of-W backup workers,
random per-iteration delays, until U updates; all replicas must end bit-identical and every update
must have taken exactly M gradients. Usage: r2_ulog_stress.py W M U"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from mp_util import run_workers  # noqa: E402
from test_update_log_gpu import _stress_worker  # noqa: E402

if __name__ == "__main__":
    w, m, u = (int(x) for x in sys.argv[1:4])
    out = run_workers(_stress_worker, w, f"dtf_stress_{os.getpid()}", m, u, timeout=900)
    versions = {o["version"] for o in out}
    psums = {o["psum"] for o in out}
    included = sum(o["included"] for o in out)
    print(f"W={w} M={m}: versions {versions}, distinct parameter sums {len(psums)}, gradients included {included} "
          f"(= M x updates: {included == m * max(versions)}), iterations per replica {[o['iters'] for o in out]}")
    assert len(versions) == 1 and len(psums) == 1 and included == m * max(versions)
