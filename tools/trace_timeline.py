#!/usr/bin/env python3
"""Prints the kernel timeline of the last training step found in a rocprofv3 --kernel-trace
database (rocpd sqlite): start offset, duration, queue, grid, LDS, kernel name."""
import re
import sqlite3
import sys


def short(n: str) -> str:
    m = re.search(r"(\w+_kernel)", n)
    base = m.group(1) if m else n[:40]
    t = re.search(r"_kernel<([^>]*)>", n)
    if t:
        base += "<" + t.group(1).replace(" ", "").replace("false", "0").replace("true", "1") + ">"
    return base


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, start, end, queue_id, grid_x, workgroup_x, lds_size, vgpr_count from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if "pack_multi" in r[0]]
    s = marks[-1] if marks else 0
    t0 = rows[s][1]
    busy = 0
    for r in rows[s:]:
        busy += r[2] - r[1]
        print(f"{(r[1] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.1f}us q{r[3]} blocks{r[4] // max(1, r[5]):5d} lds{r[6]:6d} vgpr{r[7]:4d} {short(r[0])}")
    print(f"span {(rows[-1][2] - t0) / 1e3:.1f} us, sum of kernel durations {busy / 1e3:.1f} us, {len(rows) - s} kernels")


if __name__ == "__main__":
    main()
