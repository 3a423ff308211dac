#!/usr/bin/env python3
"""Lists the gfx950 kernels whose registers spill to scratch memory.

A spilled register is reloaded with a vector memory operation, and those retire in order: inside a
persistent loop the reload queues behind the next tile's prefetch and waits for its trip to HBM (the
64-channel split forward kernel lost a quarter of its time that way). The check compiles each kernel
file with -Rpass-analysis=kernel-resource-usage and reports every kernel above its allowance.

    python tools/check_spills.py [file.hip ...]     (default: the kernels of the training steps)

Exit status 1 if a kernel spills more registers than ALLOWED grants it (default 0).
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
# the kernels a ResNet / LeNet training step launches (the general fallbacks conv_fprop.hip and
# conv_fprop_persist.hip, which serve odd shapes only, are not held to this)
STEP_FILES = ("conv_fprop_band.hip", "conv_bwd_fused.hip", "conv_fprop_split.hip", "conv_bwd_split.hip", "conv_bwd_split_pc.hip",
              "conv_wgrad_band.hip", "conv_dgrad_s2.hip", "conv_stem.hip", "conv5x5_tile.hip", "conv5x5_split.hip", "lenet_conv1.hip",
              "gemm.hip", "optim.hip", "bn.hip", "head.hip", "fc_head.hip", "elementwise.hip", "conv_wgrad.hip")
# substring of the mangled kernel name -> spilled registers tolerated
ALLOWED = {
    "conv_fprop_split_kernelILi64ELi8ELi8ELb0E": 2,    # two reloads per tile, outside the MFMA loop
    "conv_fprop_split_kernelILi64ELi8ELi8ELb1E": 32,   # data gradient as a forward product: not launched by the fused backward step
    "conv_bwd_split_kernelILi16ELi32ELi8ELb1ELi0ELb0E": 3,
    "conv_bwd_split_kernelILi16ELi32ELi8ELb1ELi0ELb1E": 8,   # the instrumented instantiation of bench/diag/split_bwd_timeline.py
}


def spills(src: Path) -> list[tuple[str, int]]:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-I", str(ROOT / "csrc"),
                        "-I", str(ROOT / "csrc" / "kernels"), "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        "-c", "-o", os.devnull, str(src)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{src.name}: compile failed\n{r.stderr[-2000:]}")
    out, fn = [], "?"
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            fn = m.group(1)
        m = re.search(r"VGPRs Spill: (\d+)", line)
        if m:
            out.append((fn, int(m.group(1))))
    return out


def main(argv: list[str]) -> int:
    files = [Path(a) for a in argv] or [ROOT / "csrc" / "kernels" / f for f in STEP_FILES if (ROOT / "csrc" / "kernels" / f).exists()]
    with ThreadPoolExecutor(max_workers=min(6, len(files))) as ex:
        results = list(ex.map(spills, files))
    bad = kernels = 0
    for f, res in zip(files, results):
        for fn, n in res:
            kernels += 1
            allowed = max([v for k, v in ALLOWED.items() if k in fn] or [0])
            if n:
                over = n > allowed
                bad += over
                print(f"{f.name}: {fn}: {n} spilled registers (allowed {allowed}){'  UNEXPECTED' if over else ''}")
    print(f"{len(files)} files, {kernels} kernels, {bad} kernels above their spill allowance")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
