#!/usr/bin/env python3
"""`python tools/benchmark.py use_dir <cfg dir> | select_files <cfg>...` for this workload (see
tools/benchmark.py at the repository root)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import benchmark  # noqa: E402

if __name__ == "__main__":
    benchmark.main()
