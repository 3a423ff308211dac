"""Experiment configuration files.

An experiment is described by a `Cfg({...})` literal whose string values may reference other
keys with `%(key)s`, resolved lazily and recursively on access (reference:
DistributedLeNet/tools/tf_ec2.py:16-24). The reference loads these files with `eval`
(reference: DistributedLeNet/tools/benchmark.py:13-15); here they are parsed with `ast` and
only literals are accepted, so a cfg file cannot execute code. Reference cfg files load
unchanged; their EC2/NFS/ssh keys are simply unused by the single-node launcher.
"""
from __future__ import annotations

import ast
import json
import os
import re
from typing import Any, Dict, Iterable, List

_REF = re.compile(r"%\(([A-Za-z_][A-Za-z0-9_]*)\)s")

# Keys a reference cfg may contain (reference: tf_ec2.py:26-143) and what the local launcher
# does with them.
CLUSTER_KEYS = ("n_masters", "n_workers", "n_ps", "n_evaluators", "num_replicas_to_aggregate")
EC2_ONLY_KEYS = (
    "key_name", "method", "region", "availability_zone", "master_type", "worker_type", "ps_type", "evaluator_type",
    "image_id", "spot_price", "ssh_username", "path_to_keyfile", "nfs_ip_address",
)


class Cfg(dict):
    """dict whose string values (also inside lists) expand `%(key)s` references on access."""

    _MAX_DEPTH = 32

    def _expand(self, value: Any, depth: int = 0) -> Any:
        if depth > self._MAX_DEPTH:
            raise ValueError("cfg: circular %(key)s reference")
        if isinstance(value, str):
            def sub(m: "re.Match[str]") -> str:
                key = m.group(1)
                if not dict.__contains__(self, key):
                    raise KeyError(f"cfg: %({key})s refers to a missing key")
                return str(self._expand(dict.__getitem__(self, key), depth + 1))

            return _REF.sub(sub, value)
        if isinstance(value, list):
            return [self._expand(v, depth) for v in value]
        if isinstance(value, tuple):
            return tuple(self._expand(v, depth) for v in value)
        return value

    def __getitem__(self, key: str) -> Any:
        return self._expand(dict.__getitem__(self, key))

    def get(self, key: str, default: Any = None) -> Any:
        return self[key] if dict.__contains__(self, key) else default

    def raw(self, key: str) -> Any:
        return dict.__getitem__(self, key)

    def resolved(self) -> Dict[str, Any]:
        return {k: self[k] for k in self.keys()}

    def with_overrides(self, **kw: Any) -> "Cfg":
        c = Cfg(dict(self))
        for k, v in kw.items():
            dict.__setitem__(c, k, v)
        return c

    # topology helpers -------------------------------------------------------------------------------
    @property
    def num_workers(self) -> int:
        """master + workers = replicas (the master is worker task 0; reference: tf_ec2.py:471-498)."""
        return int(self.get("n_masters", 1)) + int(self.get("n_workers", 0))

    @property
    def replicas_to_aggregate(self) -> int:
        v = int(self.get("num_replicas_to_aggregate", -1))
        return self.num_workers if v < 0 else v


def parse_cfg_text(text: str) -> Cfg:
    """Parses `Cfg({...})` or a bare dict literal (Python syntax, comments allowed) or JSON."""
    stripped = text.strip()
    if stripped.startswith("{") and '"' in stripped:
        try:
            return Cfg(json.loads(stripped))
        except ValueError:
            pass
    tree = ast.parse(stripped, mode="eval")
    node = tree.body
    if isinstance(node, ast.Call):
        if not (isinstance(node.func, ast.Name) and node.func.id == "Cfg" and len(node.args) == 1 and not node.keywords):
            raise ValueError("cfg: expected a single Cfg({...}) literal")
        node = node.args[0]
    if not isinstance(node, ast.Dict):
        raise ValueError("cfg: expected a dict literal")
    return Cfg(ast.literal_eval(node))


def load_cfg_from_file(path: str) -> Cfg:
    with open(path, "r") as f:
        return parse_cfg_text(f.read())


def load_cfgs(paths: Iterable[str]) -> List[Cfg]:
    return [load_cfg_from_file(p) for p in paths]


def load_cfg_dir(directory: str) -> List[Cfg]:
    """All cfg files of a directory, sorted by name (reference: benchmark.py `use_dir`)."""
    names = sorted(n for n in os.listdir(directory) if not n.startswith(".") and os.path.isfile(os.path.join(directory, n)))
    return [load_cfg_from_file(os.path.join(directory, n)) for n in names]


_FLAG = re.compile(r"--([A-Za-z_][A-Za-z0-9_]*)(?:=('[^']*'|\"[^\"]*\"|\S+))?")


def flags_from_command(command: str) -> Dict[str, str]:
    """Extracts `--flag=value` pairs from a train/evaluate command line of a cfg."""
    out: Dict[str, str] = {}
    for m in _FLAG.finditer(command):
        val = m.group(2)
        if val is None:
            val = "true"
        elif val[0] in "'\"" and val[-1] == val[0]:
            val = val[1:-1]
        out[m.group(1)] = val
    return out


def train_flags(cfg: Cfg) -> Dict[str, str]:
    """Flags of the cfg's training command with placeholders still in place."""
    cmds = cfg.get("train_commands", [])
    flags: Dict[str, str] = {}
    for c in cmds:
        flags.update(flags_from_command(c))
    return flags


def evaluate_flags(cfg: Cfg) -> Dict[str, str]:
    flags: Dict[str, str] = {}
    for c in cfg.get("evaluate_commands", []):
        if "eval" in c and "tensorboard" not in c:
            flags.update(flags_from_command(c))
    return flags


def substitute_placeholders(command: str, *, ps_hosts: str, worker_hosts: str, task_id: int, job_name: str, role_id: str) -> str:
    """Launch-time placeholder substitution (reference: tf_ec2.py:482-498)."""
    return (command.replace("PS_HOSTS", ps_hosts).replace("WORKER_HOSTS", worker_hosts).replace("TASK_ID", str(task_id))
            .replace("JOB_NAME", job_name).replace("ROLE_ID", role_id))
