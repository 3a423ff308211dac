#!/usr/bin/env python3
"""Prints VGPR/SGPR/scratch/occupancy per kernel of one HIP source (hipcc -Rpass-analysis)."""
import re
import subprocess
import sys

src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-I", "csrc",
       "-I", "/usr/include/python3.10", "-c", src, "-o", "/tmp/_res.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        if cur and filt in cur.get("Function Name", ""):
            print(cur)
        cur = {}
    cur[k] = v
if cur and filt in cur.get("Function Name", ""):
    print(cur)
