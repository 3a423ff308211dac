#!/usr/bin/env python3
"""Per-kernel totals of the last training step in a rocprofv3 --kernel-trace database, as a
markdown table (kernel, template arguments, launches, mean us, total us, share)."""
import collections
import sqlite3
import sys

from trace_timeline import short


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, start, end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if "pack_multi" in r[0]]
    s = marks[-1] if marks else 0
    e = len(rows)
    # stop at the optimizer of that step
    for i in range(s, len(rows)):
        if "adam_kernel" in rows[i][0] or "momentum_kernel" in rows[i][0]:
            e = i + 1
            break
    agg = collections.OrderedDict()
    for name, t0, t1 in rows[s:e]:
        k = short(name)
        c, tot = agg.get(k, (0, 0.0))
        agg[k] = (c + 1, tot + (t1 - t0) / 1e3)
    total = sum(v[1] for v in agg.values())
    span = (rows[e - 1][2] - rows[s][1]) / 1e3
    print(f"step span {span:.1f} us, sum of kernel durations {total:.1f} us, {e - s} kernels\n")
    print("| kernel | launches | mean us | total us | share |")
    print("|---|---:|---:|---:|---:|")
    for k, (c, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {c} | {tot / c:.1f} | {tot:.1f} | {100 * tot / total:.1f}% |")


if __name__ == "__main__":
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    main()
