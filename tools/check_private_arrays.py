#!/usr/bin/env python3
"""Lists private-memory arrays that survive optimisation in the gfx950 kernels.

A register array that the optimizer cannot break up (a struct array filled by memcpy, a dynamically
indexed array) stays in scratch memory without being reported as a spill; for a prefetch buffer
this silently makes the prefetch synchronous. The check compiles each kernel file to LLVM IR and
reports every remaining `alloca` with the kernel it belongs to.

    python tools/check_private_arrays.py [file.hip ...]     (default: every csrc/kernels/*.hip)

Exit status 1 if any kernel outside the allow-list keeps a private array.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
# dynamically indexed per-row logits of the tiny classifier-head kernels: a few launches per step
ALLOWED = ("head_fwd_kernel", "softmax_xent_kernel")


def private_arrays(src: Path) -> list[tuple[str, str]]:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = Path(d) / "k.ll"
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-I", str(ROOT / "csrc"),
                        "--cuda-device-only", "-S", "-emit-llvm", "-o", str(out), str(src)],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        found, fn = [], "?"
        for line in out.read_text().splitlines():
            if line.startswith("define "):
                m = re.search(r"@([\w$.]+)\(", line)
                fn = m.group(1) if m else "?"
            elif " = alloca " in line:
                found.append((fn, line.strip()))
        return found


def main(argv: list[str]) -> int:
    files = [Path(a) for a in argv] or sorted((ROOT / "csrc" / "kernels").glob("*.hip")) + sorted((ROOT / "csrc" / "comm").glob("*.hip"))
    with ThreadPoolExecutor(max_workers=min(8, len(files))) as ex:
        results = list(ex.map(private_arrays, files))
    bad = 0
    for f, res in zip(files, results):
        for fn, line in res:
            ok = any(a in fn for a in ALLOWED)
            bad += not ok
            print(f"{f.name}: {fn}: {line}{'  (allowed)' if ok else ''}")
    print(f"{len(files)} files, {bad} unexpected private arrays")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
