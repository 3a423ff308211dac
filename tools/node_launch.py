#!/usr/bin/env python3
"""Single-node cluster tool: the local replacement of the reference's EC2 launcher.

The reference drives experiments with `python tools/tf_ec2.py <command>` against a fleet of EC2
instances (reference: DistributedLeNet/tools/tf_ec2.py:145-811): request instances, mount NFS,
ssh the per-role command lines built from a `Cfg`, poll, kill, scp results back. On one
8 x MI355X node the same verbs act on local processes, one per GPU:

  launch                 prepare the run directory (`base_out_dir`) and record the slot table
  run_tf                 start master (worker task 0) + workers + evaluator from the cfg's commands
  clean_launch_and_run   shutdown + launch + run_tf
  kill_all_python        stop every process this tool started for the cfg
  kill_python            stop the training/evaluator processes (same on one node)
  list_idle_instances    slots without a live process
  list_running_instances slots with a live process
  run_command <cmd>      run a shell command in the run directory
  download_outdir <dst>  copy the run directory
  download_file <cluster_string> <name> <dst>   copy one output file (out_master, out_evaluator, ...)
  shutdown               kill_all_python + forget the slot table

A cfg's `train_commands` / `evaluate_commands` are not executed as shell text: the flags are
parsed out of them (with %(key)s references and the PS_HOSTS / WORKER_HOSTS / TASK_ID / JOB_NAME /
ROLE_ID placeholders resolved as in reference tf_ec2.py:482-498) and passed to this repository's
entry scripts. EC2, ssh and NFS keys are ignored; `nfs_mount_point` is mapped to a local directory.
Parameter-server roles are not started because no process hosts parameters here; roles and output
files keep the reference's names (`out_master`, `out_worker<i>`, `out_evaluator`). A topology with
more workers than GPUs (the 50-worker sweeps) is scaled to the available replicas, keeping the
aggregated fraction.
"""
from __future__ import annotations

import json
import os
import shutil
import signal
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from distributed_tensorflow_amd import cfg as cfg_mod  # noqa: E402
from distributed_tensorflow_amd.cfg import Cfg  # noqa: E402

WORKLOADS = {
    "mnist_distributed_train.py": ("DistributedLeNet", "src/mnist_distributed_train.py", "src/mnist_eval.py"),
    "resnet_distributed_train.py": ("DistributedResNet", "src/resnet_distributed_train.py", "src/resnet_eval.py"),
}

DEFAULT_CFG = Cfg({
    "name": "local_resnet20_sync",
    "n_masters": 1, "n_workers": 7, "n_ps": 1, "n_evaluators": 1, "num_replicas_to_aggregate": "8",
    "nfs_mount_point": "runs", "base_out_dir": "%(nfs_mount_point)s/%(name)s",
    "batch_size": "256", "max_steps": "2000", "initial_learning_rate": ".001",
    "learning_rate_decay_factor": ".95", "num_epochs_per_decay": "1.0",
    "train_commands": [
        "python src/resnet_distributed_train.py --batch_size=%(batch_size)s --initial_learning_rate=%(initial_learning_rate)s "
        "--learning_rate_decay_factor=%(learning_rate_decay_factor)s --num_epochs_per_decay=%(num_epochs_per_decay)s "
        "--train_dir=%(base_out_dir)s/train_dir --worker_hosts='WORKER_HOSTS' --ps_hosts='PS_HOSTS' --task_id=TASK_ID "
        "--max_steps=%(max_steps)s --num_residual_blocks=3 --backup_worker_method=true "
        "--num_replicas_to_aggregate=%(num_replicas_to_aggregate)s --job_name=JOB_NAME > %(base_out_dir)s/out_ROLE_ID 2>&1 &"
    ],
    "evaluate_commands": [
        "python src/resnet_eval.py --eval_dir=%(base_out_dir)s/eval_dir --checkpoint_dir=%(base_out_dir)s/train_dir "
        "--num_residual_blocks=3 > %(base_out_dir)s/out_evaluator 2>&1 &"
    ],
})


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _alive(pid: int) -> bool:
    try:
        os.kill(pid, 0)
    except OSError:
        return False
    # a zombie child still answers signal 0: reap it if it is ours
    try:
        done, _ = os.waitpid(pid, os.WNOHANG)
        return done == 0
    except ChildProcessError:
        return True


def _gpu_count() -> int:
    """GPUs on this node, counted in the kernel driver's sysfs tree: the launcher is the parent of
    every rank process and never loads the HIP runtime itself."""
    from distributed_tensorflow_amd.launch import visible_gpus

    return visible_gpus()


class LocalCluster:
    def __init__(self, configuration: Cfg, out_root: Optional[str] = None, max_workers: Optional[int] = None,
                 extra_flags: Optional[List[str]] = None):
        root = out_root or os.environ.get("DTF_RUNS", os.path.join(ROOT, "runs"))
        self.cfg = configuration.with_overrides(nfs_mount_point=root)
        self.base_out_dir = self.cfg["base_out_dir"]
        self.state_path = os.path.join(self.base_out_dir, "cluster_state.json")
        self.max_workers = max_workers
        self.extra_flags = list(extra_flags or [])

    # -- state -----------------------------------------------------------------------------------
    def _load(self) -> Dict:
        if os.path.exists(self.state_path):
            with open(self.state_path) as f:
                return json.load(f)
        return {"slots": [], "procs": {}}

    def _save(self, st: Dict) -> None:
        os.makedirs(self.base_out_dir, exist_ok=True)
        with open(self.state_path, "w") as f:
            json.dump(st, f, indent=1)

    def topology(self) -> Dict:
        """Roles started for this cfg. A cfg written for more replicas than this node runs (the
        reference's sweeps use 50 EC2 workers, reference: DistributedLeNet/cfg/50_workers/*) is scaled:
        `max_workers` (default: one replica per GPU) replicas, with the number of gradients per update
        scaled by the same factor, so "40 of 50" becomes "6 of 8". The caps behind that: the device
        update log addresses 8 replicas (one node's GPUs; csrc/comm/update_log.h: kUlogMaxPeers, a
        16-bit contributor mask next to the update number and count in ONE 64-bit word that arrivals
        compare-exchange), the host update log 16 (csrc/host/board.h: kAggMaxRanks, same word layout);
        plain synchronous training has no such cap (the process group alone, any number of ranks).
        Passing max_workers above the GPU count puts several replicas on a GPU (--comm_backend=gloo)."""
        n_rep = self.cfg.num_workers
        agg = self.cfg.replicas_to_aggregate
        cap = self.max_workers or (_gpu_count() or os.cpu_count() or 1)
        if n_rep > cap:
            agg = max(1, round(agg * cap / n_rep))
            n_rep = cap
        return {"replicas": n_rep, "aggregate": min(agg, n_rep), "evaluators": int(self.cfg.get("n_evaluators", 0)),
                "ps": int(self.cfg.get("n_ps", 0))}

    # -- verbs -----------------------------------------------------------------------------------
    def launch(self) -> Dict:
        topo = self.topology()
        if os.path.isdir(self.base_out_dir):
            self.kill_all_python()
            shutil.rmtree(self.base_out_dir)
        os.makedirs(self.base_out_dir)
        slots = ["master"] + [f"worker{i}" for i in range(topo["replicas"] - 1)] + \
                [f"ps{i}" for i in range(topo["ps"])] + (["evaluator"] if topo["evaluators"] else [])
        st = {"slots": slots, "procs": {}, "topology": topo}
        self._save(st)
        print("launched %d local slots in %s" % (len(slots), self.base_out_dir))
        return st

    def _workload(self, flags_text: str):
        for script, spec in WORKLOADS.items():
            if script in flags_text:
                return spec
        raise ValueError("cfg train command names no known training script")

    def run_tf(self) -> Dict:
        st = self._load()
        if not st["slots"]:
            st = self.launch()
        topo = st["topology"]
        train_cmds = self.cfg.get("train_commands", [])
        workdir, train_script, eval_script = self._workload(" ".join(train_cmds))
        n = topo["replicas"]
        ports = [_free_port() for _ in range(n)]
        worker_hosts = ",".join(f"127.0.0.1:{p}" for p in ports)
        ps_hosts = ",".join(f"127.0.0.1:{_free_port()}" for _ in range(max(1, topo["ps"])))
        board = "dtf_%s_%d" % (self.cfg["name"][:40].replace("/", "_"), os.getpid())
        flags = cfg_mod.train_flags(self.cfg)
        ngpu = _gpu_count()
        env_base = dict(os.environ)
        env_base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        for task in range(n):
            role = "master" if task == 0 else f"worker{task - 1}"
            f = dict(flags)
            f.update({"worker_hosts": worker_hosts, "ps_hosts": ps_hosts, "task_id": str(task), "job_name": "worker",
                      "num_replicas_to_aggregate": str(topo["aggregate"]), "board_name": board})
            argv = [sys.executable, os.path.join(ROOT, workdir, train_script)] + [f"--{k}={v}" for k, v in f.items()] + self.extra_flags
            env = dict(env_base)
            env["LOCAL_RANK"] = str(task % ngpu) if ngpu else str(task)
            out = open(os.path.join(self.base_out_dir, f"out_{role}"), "w")
            p = subprocess.Popen(argv, stdout=out, stderr=subprocess.STDOUT, env=env, cwd=os.path.join(ROOT, workdir),
                                 start_new_session=True)
            st["procs"][role] = p.pid
        for i in range(topo["ps"]):
            print(f"ps{i}: no process started (parameters are resident on the worker GPUs)")
        if topo["evaluators"]:
            ef = cfg_mod.evaluate_flags(self.cfg)
            for k in ("num_residual_blocks", "data_dir", "synthetic_data"):
                if k in flags and k not in ef:
                    ef[k] = flags[k]
            argv = [sys.executable, os.path.join(ROOT, workdir, eval_script)] + [f"--{k}={v}" for k, v in ef.items()]
            argv += [a for a in self.extra_flags if a.split("=")[0] in ("--synthetic_data", "--device", "--data_dir")]
            env = dict(env_base)
            out = open(os.path.join(self.base_out_dir, "out_evaluator"), "w")
            p = subprocess.Popen(argv, stdout=out, stderr=subprocess.STDOUT, env=env, cwd=os.path.join(ROOT, workdir),
                                 start_new_session=True)
            st["procs"]["evaluator"] = p.pid
        self._save(st)
        cluster_save = {
            "cluster_string": self.base_out_dir, "worker_hosts": worker_hosts, "ps_hosts": ps_hosts,
            "roles": sorted(st["procs"].keys()), "base_out_dir": self.base_out_dir,
        }
        return cluster_save

    def kill_all_python(self) -> int:
        st = self._load()
        killed = 0
        for role, pid in list(st["procs"].items()):
            if _alive(pid):
                try:
                    os.killpg(pid, signal.SIGTERM)   # each child leads its own session/group
                    killed += 1
                except OSError:
                    pass
        deadline = time.time() + 10
        while time.time() < deadline and any(_alive(p) for p in st["procs"].values()):
            time.sleep(0.1)
        for pid in st["procs"].values():
            if _alive(pid):
                try:
                    os.killpg(pid, signal.SIGKILL)
                except OSError:
                    pass
        st["procs"] = {}
        if os.path.isdir(self.base_out_dir):
            self._save(st)
        return killed

    kill_python = kill_all_python

    def list_running_instances(self) -> List[str]:
        st = self._load()
        return [r for r, pid in st["procs"].items() if _alive(pid)]

    def list_idle_instances(self) -> List[str]:
        st = self._load()
        running = set(self.list_running_instances())
        return [s for s in st["slots"] if s not in running]

    def wait(self, roles: Optional[List[str]] = None, timeout_s: float = 3600) -> bool:
        st = self._load()
        deadline = time.time() + timeout_s
        pids = [pid for r, pid in st["procs"].items() if roles is None or r in roles]
        while time.time() < deadline:
            if not any(_alive(p) for p in pids):
                return True
            time.sleep(0.2)
        return False

    def run_command(self, command: str) -> int:
        os.makedirs(self.base_out_dir, exist_ok=True)
        return subprocess.call(command, shell=True, cwd=self.base_out_dir)

    def download_outdir(self, dst: str) -> str:
        target = os.path.join(dst, os.path.basename(self.base_out_dir))
        if os.path.exists(target):
            shutil.rmtree(target)
        shutil.copytree(self.base_out_dir, target)
        return target

    def download_file(self, cluster_string: str, name: str, dst: str) -> str:
        src_dir = cluster_string if os.path.isdir(cluster_string) else self.base_out_dir
        os.makedirs(dst, exist_ok=True)
        out = os.path.join(dst, name)
        shutil.copyfile(os.path.join(src_dir, name), out)
        return out

    def shutdown(self) -> None:
        self.kill_all_python()
        if os.path.exists(self.state_path):
            os.unlink(self.state_path)


def tf_ec2_run(argv: List[str], configuration: Cfg, **kw):
    """Same calling convention as the reference: argv[0] is the tool name, argv[1] the command."""
    if len(argv) < 2:
        print(__doc__)
        return None
    cluster = LocalCluster(configuration, **kw)
    cmd, args = argv[1], argv[2:]
    if cmd == "launch":
        return cluster.launch()
    if cmd == "run_tf":
        return cluster.run_tf()
    if cmd == "clean_launch_and_run":
        cluster.shutdown()
        cluster.launch()
        return cluster.run_tf()
    if cmd in ("kill_all_python", "kill_python"):
        return cluster.kill_all_python()
    if cmd == "list_idle_instances":
        r = cluster.list_idle_instances()
        print("\n".join(r))
        return r
    if cmd == "list_running_instances":
        r = cluster.list_running_instances()
        print("\n".join(r))
        return r
    if cmd == "run_command":
        return cluster.run_command(" ".join(args))
    if cmd == "download_outdir":
        return cluster.download_outdir(args[0] if args else "/tmp")
    if cmd == "download_file":
        return cluster.download_file(args[0], args[1], args[2] if len(args) > 2 else "/tmp")
    if cmd == "shutdown":
        return cluster.shutdown()
    if cmd == "wait":
        return cluster.wait()
    raise SystemExit(f"unknown command {cmd}")


def main() -> None:
    argv = list(sys.argv)
    if len(argv) < 2 or argv[1] in ("-h", "--help", "help"):
        print(__doc__)
        print("usage: node_launch.py [--cfg FILE] [--out_root DIR] [--max_workers N] <command> [args] [-- extra entry-script flags]")
        return
    configuration = DEFAULT_CFG
    extra: List[str] = []
    kw = {}
    # --cfg FILE, --out_root DIR, --max_workers N, and `-- extra flags for the entry scripts`
    if "--" in argv:
        i = argv.index("--")
        extra = argv[i + 1:]
        argv = argv[:i]
    rest = [argv[0]]
    i = 1
    while i < len(argv):
        if argv[i] == "--cfg":
            configuration = cfg_mod.load_cfg_from_file(argv[i + 1])
            i += 2
        elif argv[i] == "--out_root":
            kw["out_root"] = argv[i + 1]
            i += 2
        elif argv[i] == "--max_workers":
            kw["max_workers"] = int(argv[i + 1])
            i += 2
        else:
            rest.append(argv[i])
            i += 1
    tf_ec2_run(rest, configuration, extra_flags=extra, **kw)


if __name__ == "__main__":
    main()
