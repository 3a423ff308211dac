#!/usr/bin/env python3
"""Builds the native extensions in-tree with explicit compiler invocations.

  distributed_tensorflow_amd/_hip<ext>   gfx950 HIP kernels + step-program runtime (hipcc)
  distributed_tensorflow_amd/_host<ext>  host-only C++: TF tensor-bundle checkpoint I/O,
                                         shared-memory control board (g++)

Objects are cached under build/ keyed by a hash of the source, its local includes and the
command line, so rebuilding after a one-file edit takes seconds. `hipcc` cross-compiles for
gfx950 without a GPU present.

Usage: python build_native.py [--force] [--jobs N] [--only hip|host]
"""
from __future__ import annotations

import argparse
import hashlib
import os
import re
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
BUILD = ROOT / "build"
PKG = ROOT / "distributed_tensorflow_amd"
EXT = sysconfig.get_config_var("EXT_SUFFIX")
ARCH = "gfx950"


def _pybind_include() -> str:
    import pybind11

    return pybind11.get_include()


def _local_includes(path: Path, seen: set[Path] | None = None) -> set[Path]:
    seen = set() if seen is None else seen
    if path in seen or not path.exists():
        return seen
    seen.add(path)
    for m in re.finditer(r'#include\s+"([^"]+)"', path.read_text()):
        for base in (path.parent, CSRC):
            cand = (base / m.group(1)).resolve()
            if cand.exists():
                _local_includes(cand, seen)
                break
    return seen


def _digest(src: Path, cmd: list[str]) -> str:
    h = hashlib.sha256()
    for p in sorted(_local_includes(src)):
        h.update(p.read_bytes())
    h.update(" ".join(cmd).encode())
    return h.hexdigest()[:16]


def _compile(src: Path, cmd_prefix: list[str], force: bool) -> Path:
    obj = BUILD / (src.relative_to(CSRC).as_posix().replace("/", "_") + ".o")
    cmd = cmd_prefix + ["-c", str(src), "-o", str(obj)]
    stamp = obj.with_suffix(".hash")
    dig = _digest(src, cmd_prefix)
    if not force and obj.exists() and stamp.exists() and stamp.read_text() == dig:
        return obj
    print("[build]", src.relative_to(ROOT), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError(f"compile failed: {src}")
    if res.stderr.strip():
        sys.stderr.write(res.stderr)
    stamp.write_text(dig)
    return obj


def build_hip(force: bool = False, jobs: int = 8) -> Path:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    py_inc = sysconfig.get_paths()["include"]
    common = [
        hipcc, "-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-fast-math",
        "-I", str(CSRC), "-I", py_inc, "-I", _pybind_include(), "-fvisibility=hidden", "-Wno-unused-result",
    ]
    if os.environ.get("DTF_CONV_DEBUG") == "1":   # run-time ablation switches of the band kernels (bench/conv_ablation.py)
        common.append("-DDTF_CONV_DEBUG=1")
    common += os.environ.get("DTF_EXTRA_DEFS", "").split()   # experiment switches, e.g. "-DDTF_FBAND_MTW4=1"
    srcs = sorted((CSRC / "kernels").glob("*.hip")) + sorted((CSRC / "comm").glob("*.hip")) + [
        CSRC / "runtime" / "program.cc", CSRC / "bindings_hip.cc",
    ]
    srcs = [s for s in srcs if s.exists()]
    BUILD.mkdir(exist_ok=True)

    def one(src: Path) -> Path:
        prefix = list(common)
        if src.suffix != ".hip":
            prefix += ["-x", "hip"]
        return _compile(src, prefix, force)

    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(one, srcs))
    out = PKG / f"_hip{EXT}"
    newest = max(o.stat().st_mtime for o in objs)
    if force or not out.exists() or out.stat().st_mtime < newest:
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(out)] + [str(o) for o in objs]
        print("[link]", out.relative_to(ROOT), flush=True)
        subprocess.run(cmd, check=True)
    return out


def build_host(force: bool = False, jobs: int = 8) -> Path | None:
    srcs = sorted((CSRC / "ckpt").glob("*.cc")) + sorted((CSRC / "host").glob("*.cc"))
    binding = CSRC / "bindings_host.cc"
    if not binding.exists():
        return None
    srcs.append(binding)
    cxx = os.environ.get("CXX", "g++")
    py_inc = sysconfig.get_paths()["include"]
    common = [cxx, "-O2", "-std=c++17", "-fPIC", "-I", str(CSRC), "-I", py_inc, "-I", _pybind_include(),
              "-fvisibility=hidden", "-pthread"]
    BUILD.mkdir(exist_ok=True)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, list(common), force), srcs))
    out = PKG / f"_host{EXT}"
    newest = max(o.stat().st_mtime for o in objs)
    if force or not out.exists() or out.stat().st_mtime < newest:
        cmd = [cxx, "-shared", "-fPIC", "-pthread", "-o", str(out)] + [str(o) for o in objs] + ["-lrt"]
        print("[link]", out.relative_to(ROOT), flush=True)
        subprocess.run(cmd, check=True)
    return out


def build_all(force: bool = False, jobs: int = 8, only: str | None = None) -> None:
    if only in (None, "host"):
        build_host(force, jobs)
    if only in (None, "hip"):
        build_hip(force, jobs)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--only", choices=["hip", "host"], default=None)
    args = ap.parse_args()
    build_all(args.force, args.jobs, args.only)
    print("native build OK")
