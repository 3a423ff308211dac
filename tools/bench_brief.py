#!/usr/bin/env python3
"""Prints the essentials of bench.py JSON lines (one file per argument)."""
import json
import sys

for path in sys.argv[1:]:
    for line in open(path):
        line = line.strip()
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        c = d.get("config", {})
        print(f"{d['metric']}: {d['value']:.0f} {d['unit']}  {d['ms_per_step']} ms/step  ranks={d.get('ranks')} "
              f"gpus={d.get('distinct_gpus')} backend={d.get('backend')} batch/gpu={c.get('per_gpu_batch')} mode={c.get('mode', 'sync')}")
        for k in ("replicas_to_aggregate", "interval_ms", "inject_straggler_ms", "updates_applied", "gradients_aggregated",
                  "gradients_computed", "async_stats"):
            if k in c:
                print(f"    {k}: {c[k]}")
        for row in c.get("per_rank", []):
            print("    ", row)
