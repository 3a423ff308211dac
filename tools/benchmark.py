#!/usr/bin/env python3
"""Experiment sweeps, log scraping and plots (reference: DistributedLeNet/tools/benchmark.py).

  python tools/benchmark.py use_dir <cfg_dir> [n_iters]
  python tools/benchmark.py select_files <cfg> [<cfg> ...]

For every cfg: (re)launch the local cluster, start training + the evaluator, poll `out_master`
until `step N` passes `n_iters`, stop everything, collect `out_master` / `out_evaluator`, then
parse them with the same regular expressions as the reference and draw the same figures:
time-loss, time-step, time-precision, step-loss and (for the worker-time CDF method) the compute
time CDF plus the percentile report.
"""
from __future__ import annotations

import json
import os
import re
import sys
import time
from typing import Dict, List, Sequence, Tuple

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from distributed_tensorflow_amd.cfg import Cfg, load_cfg_dir, load_cfg_from_file  # noqa: E402,F401
from node_launch import tf_ec2_run  # noqa: E402


def shutdown_and_launch(cfg: Cfg, **kw) -> None:
    tf_ec2_run("tools/node_launch.py shutdown".split(), cfg, **kw)
    tf_ec2_run("tools/node_launch.py launch".split(), cfg, **kw)


def current_iteration(path: str) -> int:
    cur = 0
    if not os.path.exists(path):
        return 0
    with open(path, "r", errors="replace") as f:
        for line in f:
            m = re.match(".*step ([0-9]*),.*", line)
            if m and m.group(1):
                cur = max(cur, int(m.group(1)))
    return cur


def check_if_reached_iters(cluster_string: str, n_iters: int, cfg: Cfg, master_file_name: str = "out_master", outdir: str = "/tmp/", **kw) -> bool:
    fname = tf_ec2_run(["tools/node_launch.py", "download_file", cluster_string, master_file_name, outdir], cfg, **kw)
    cur = current_iteration(fname)
    print("Currently on iteration %d" % cur)
    return cur > n_iters


def run_tf_and_download_files(n_iters: int, cfg: Cfg, evaluator_file_name: str = "out_evaluator",
                              master_file_name: str = "out_master", outdir: str = "result_dir", poll_s: float = 2.0,
                              timeout_s: float = 3600.0, **kw) -> str:
    tf_ec2_run("tools/node_launch.py kill_all_python".split(), cfg, **kw)
    cluster_specs = tf_ec2_run("tools/node_launch.py run_tf".split(), cfg, **kw)
    cluster_string = cluster_specs["cluster_string"]
    deadline = time.time() + timeout_s
    from node_launch import LocalCluster

    lc = LocalCluster(cfg, **{k: v for k, v in kw.items() if k in ("out_root", "max_workers")})
    while not check_if_reached_iters(cluster_string, n_iters, cfg, master_file_name, "/tmp/dtf_poll", **kw):
        if time.time() > deadline or "master" not in lc.list_running_instances():
            break
        time.sleep(poll_s)
    time.sleep(1.5)  # let the evaluator score the last checkpoint
    tf_ec2_run("tools/node_launch.py kill_all_python".split(), cfg, **kw)
    dst = os.path.join(outdir, cfg["name"])
    for name in (evaluator_file_name, master_file_name):
        try:
            tf_ec2_run(["tools/node_launch.py", "download_file", cluster_string, name, dst], cfg, **kw)
        except FileNotFoundError:
            pass
    return dst


# ---- log parsers (same regular expressions as the reference) -------------------------------------
def extract_times_losses_precision(fname: str) -> Tuple[List[float], List[float], List[float], List[int]]:
    """From out_evaluator: (times, losses, precisions, steps) of every evaluated checkpoint."""
    times, losses, precisions, steps = [], [], [], []
    with open(fname, "r", errors="replace") as f:
        for line in f:
            m = re.match(".*Num examples: ([0-9]*)  Precision @ 1: ([0-9.eE+-]*) Loss: ([0-9.eE+-]*) Time: ([0-9.eE+-]*)", line)
            if m:
                precisions.append(float(m.group(2)))
                losses.append(float(m.group(3)))
                times.append(float(m.group(4)))
            m = re.match(".* step=([0-9]*)", line)
            if m:
                steps.append(int(m.group(1)))
    return times, losses, precisions, steps


def extract_step_times_losses(fname: str) -> Tuple[List[int], List[float], List[float]]:
    """From out_master: (step, loss, sec/batch) of every training line."""
    steps, losses, secs = [], [], []
    pat = re.compile(r".*step ([0-9]+), loss = ([0-9.eE+-]+|nan)\(([0-9.eE+-]+) examples/sec; ([0-9.eE+-]+)  sec/batch\)")
    with open(fname, "r", errors="replace") as f:
        for line in f:
            m = pat.match(line)
            if m:
                steps.append(int(m.group(1)))
                losses.append(float(m.group(2)))
                secs.append(float(m.group(4)))
    return steps, losses, secs


def _last_list(fname: str, tag: str):
    found = None
    with open(fname, "r", errors="replace") as f:
        for line in f:
            i = line.find(tag)
            if i >= 0:
                found = line[i + len(tag):].strip()
    if found is None:
        return []
    import ast

    return ast.literal_eval(found)


def extract_compute_records(fname: str) -> List[Tuple[float, int, int]]:
    """Last `ELAPSED TIMES [(time, worker, iteration), ...]` line of out_master, whole records."""
    return _last_list(fname, "ELAPSED TIMES ")


def extract_compute_times(fname: str) -> List[float]:
    """The compute times of that line alone, what the time-CDF plot draws
    (reference: DistributedLeNet/tools/benchmark.py:113-122)."""
    return [r[0] for r in extract_compute_records(fname)]


def extract_iteration_times(fname: str) -> List[float]:
    return _last_list(fname, "ITERATION TIMES ")


def print_worker_sorted_times(fname: str) -> Dict[str, float]:
    """Percentile report of the per-worker compute times of a worker-time CDF run, with the rows of
    the reference's report (reference: DistributedLeNet/tools/benchmark.py:60-111): spread of all
    times, then - per iteration, nearest-rank - the 95th / 99th / 100th percentile over the workers,
    each summarised by its mean and its median across iterations. Returns the same numbers."""
    print("File: %s" % fname)
    print("-----------------------------")
    records = extract_compute_records(fname)
    if not records:
        print("no ELAPSED TIMES line in", fname)
        return {}
    times = np.array([r[0] for r in records], dtype=np.float64)
    by_iteration: Dict[int, List[float]] = {}
    for t, _worker, it in records:
        by_iteration.setdefault(it, []).append(t)

    def nearest(values: List[float], q: float) -> float:
        return float(np.percentile(values, q, method="nearest"))

    p95 = [nearest(v, 95) for v in by_iteration.values()]
    p99 = [nearest(v, 99) for v in by_iteration.values()]
    p100 = [nearest(v, 100) for v in by_iteration.values()]
    rows = [
        ("Stdev: ", "std", float(times.std())),
        ("Max: ", "max", float(times.max())),
        ("80 Percentile:", "p80", float(np.percentile(times, 80))),
        ("90 Percentile:", "p90", float(np.percentile(times, 90))),
        ("95 Percentile:", "p95", float(np.percentile(times, 95))),
        ("99 Percentile:", "p99", float(np.percentile(times, 99))),
        ("Mean of 95 percentile (across iterations)", "iter_p95_mean", float(np.mean(p95))),
        ("Med of 95 percentile (across iterations)", "iter_p95_med", float(np.median(p95))),
        ("Mean of 99 percentile (across iterations)", "iter_p99_mean", float(np.mean(p99))),
        ("Med of 99 percentile (across iterations)", "iter_p99_med", float(np.median(p99))),
        ("Mean of 100 percentile (across iterations)", "iter_p100_mean", float(np.mean(p100))),
        ("Med of 100 percentile (across iterations)", "iter_p100_med", float(np.median(p100))),
        ("Mean:", "mean", float(times.mean())),
    ]
    out: Dict[str, float] = {}
    for label, key, value in rows:
        print(label, value)
        out[key] = value
    print(p99)
    out["p50"] = float(np.percentile(times, 50))
    its = extract_iteration_times(fname)
    if its:
        out["iteration_time_mean"] = float(np.mean(its))
        out["iteration_time_p50"] = float(np.percentile(its, 50))
    return out


# ---- figures ------------------------------------------------------------------------------------------
def _plt():
    import matplotlib

    matplotlib.use("Agg")
    from matplotlib import pyplot as plt

    return plt


def plot_time_loss(results: Dict[str, Dict], outdir: str) -> None:
    plt = _plt()
    for key, ylabel, fname in (("losses", "Loss", "time_loss.png"), ("precisions", "Precision @ 1", "time_precision.png")):
        plt.figure()
        for name, r in results.items():
            if r["eval_times"]:
                # wall-clock of each evaluation when the log carries it, else the evaluation's index
                t = r["eval_wall"][: len(r[key])] if r.get("eval_wall") else np.arange(len(r[key]))
                plt.plot(t, r[key], label=name)
        plt.xlabel("Time (s)")
        plt.ylabel(ylabel)
        plt.legend(fontsize=6)
        plt.savefig(os.path.join(outdir, fname))
        plt.close()
    plt.figure()
    for name, r in results.items():
        plt.plot(np.cumsum(r["secs"]), r["steps"], label=name)
    plt.xlabel("Time (s)")
    plt.ylabel("Step")
    plt.legend(fontsize=6)
    plt.savefig(os.path.join(outdir, "time_step.png"))
    plt.close()
    plt.figure()
    for name, r in results.items():
        plt.plot(r["steps"], r["train_losses"], label=name)
    plt.xlabel("Step")
    plt.ylabel("Loss")
    plt.legend(fontsize=6)
    plt.savefig(os.path.join(outdir, "step_losses.png"))
    plt.close()


def plot_time_cdfs(results: Dict[str, Dict], outdir: str) -> None:
    plt = _plt()
    plt.figure()
    any_data = False
    for name, r in results.items():
        ts = sorted(r.get("compute_times", []))
        if ts:
            any_data = True
            plt.plot(ts, np.arange(1, len(ts) + 1) / len(ts), label=name)
    if any_data:
        plt.xlabel("Compute time (s)")
        plt.ylabel("CDF")
        plt.legend(fontsize=6)
        plt.savefig(os.path.join(outdir, "time_cdfs.png"))
    plt.close()


def collect_results(dirs: Dict[str, str]) -> Dict[str, Dict]:
    results = {}
    for name, d in dirs.items():
        master, evaluator = os.path.join(d, "out_master"), os.path.join(d, "out_evaluator")
        steps, train_losses, secs = extract_step_times_losses(master) if os.path.exists(master) else ([], [], [])
        times, losses, precisions, esteps = extract_times_losses_precision(evaluator) if os.path.exists(evaluator) else ([], [], [], [])
        # wall time of an evaluated step = cumulative training time at that step
        wall_at = dict(zip(steps, np.cumsum(secs))) if steps else {}
        results[name] = {
            "steps": steps, "train_losses": train_losses, "secs": secs, "eval_times": times, "losses": losses,
            "precisions": precisions, "eval_steps": esteps, "eval_wall": [float(wall_at.get(s, 0.0)) for s in esteps],
            "compute_times": extract_compute_times(master) if os.path.exists(master) else [],
        }
    return results


def plot_figs(cfgs: Sequence[Cfg], n_iters: int = 300, outdir: str = "result_dir", **kw) -> Dict[str, Dict]:
    os.makedirs(outdir, exist_ok=True)
    dirs = {}
    for c in cfgs:
        shutdown_and_launch(c, **kw)
        dirs[c["name"]] = run_tf_and_download_files(n_iters, c, outdir=outdir, **kw)
        tf_ec2_run("tools/node_launch.py shutdown".split(), c, **kw)
    results = collect_results(dirs)
    plot_time_loss(results, outdir)
    plot_time_cdfs(results, outdir)
    with open(os.path.join(outdir, "summary.json"), "w") as f:
        json.dump({k: {"final_step": (v["steps"] or [0])[-1], "final_loss": (v["train_losses"] or [None])[-1],
                       "mean_sec_per_batch": float(np.mean(v["secs"])) if v["secs"] else None,
                       "last_precision": (v["precisions"] or [None])[-1]} for k, v in results.items()}, f, indent=1)
    return results


def main() -> None:
    if len(sys.argv) < 3:
        print(__doc__)
        raise SystemExit(2)
    mode = sys.argv[1]
    extra = []
    args = sys.argv[2:]
    if "--" in args:
        i = args.index("--")
        extra = args[i + 1:]
        args = args[:i]
    if mode == "use_dir":
        cfgs = load_cfg_dir(args[0])
        n_iters = int(args[1]) if len(args) > 1 else 300
    elif mode == "select_files":
        cfgs = [load_cfg_from_file(p) for p in args]
        n_iters = 300
    else:
        raise SystemExit("mode must be use_dir or select_files")
    plot_figs(cfgs, n_iters, extra_flags=extra)


if __name__ == "__main__":
    main()
