"""The self-launcher behind `python bench.py --gpus N`: N fresh ranks with the torchrun environment,
rank 0's output relayed, failure of any rank reported; and bench.py's refusal to report a number
for a topology other than the one asked for."""
import json
import os
import subprocess
import time
import sys
import textwrap
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(tmp_path, body):
    p = tmp_path / "rank_stub.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_spawn_local_ranks_sets_the_torchrun_contract_and_relays_rank0(tmp_path, capfd):
    from distributed_tensorflow_amd.launch import spawn_local_ranks

    argv = _script(tmp_path, """
        import json, os, sys
        out = {k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        open(os.path.join(os.path.dirname(__file__), "seen_%s" % out["RANK"]), "w").write(json.dumps(out))
        print(json.dumps(out))            # only rank 0's line may reach the parent's stdout
    """)
    code = spawn_local_ranks(argv, 3, devices=2)
    assert code == 0
    out = capfd.readouterr().out.strip().splitlines()
    assert len(out) == 1
    line = json.loads(out[0])
    assert line["RANK"] == "0" and line["WORLD_SIZE"] == "3" and line["MASTER_ADDR"] == "127.0.0.1"
    seen = [json.loads((tmp_path / f"seen_{r}").read_text()) for r in range(3)]
    assert [s["RANK"] for s in seen] == ["0", "1", "2"]
    assert [s["LOCAL_RANK"] for s in seen] == ["0", "1", "0"]          # ranks wrap over the visible devices
    assert len({s["MASTER_PORT"] for s in seen}) == 1


def test_spawn_local_ranks_fails_when_any_rank_fails(tmp_path):
    from distributed_tensorflow_amd.launch import spawn_local_ranks

    argv = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)                    # the healthy rank would hang in a collective; the parent must stop it
    """)
    import time

    t0 = time.time()
    code = spawn_local_ranks(argv, 2, devices=0)
    assert code == 7
    assert time.time() - t0 < 30


def _run_bench(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=e,
                          timeout=300)


def test_bench_refuses_more_ranks_than_gpus():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a machine with fewer than 2 GPUs")
    r = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert r.stdout.strip() == ""                                    # no JSON line for a topology that was not measured
    assert "--share-gpu" in r.stderr


def test_bench_share_gpu_needs_gloo():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a machine with fewer than 2 GPUs")
    r = _run_bench("--gpus", "2", "--share-gpu", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and "gloo" in r.stderr and r.stdout.strip() == ""


def test_bench_check_topology_rejects_a_world_that_differs_from_gpus():
    sys.path.insert(0, ROOT)
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_main", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    class FakeComm:
        def __init__(self, world, distinct, backend):
            self.world, self._d, self._b = world, distinct, backend

        def topology(self):
            return {"backend": self._b, "ranks": self.world, "distinct_gpus": self._d,
                    "rccl_ranks": self.world if self._b == "nccl" else 0}

    args = types.SimpleNamespace(gpus=8, share_gpu=False)
    with pytest.raises(SystemExit):                                   # torchrun started 1 rank, --gpus says 8
        mod.check_topology(args, FakeComm(1, 1, "none"))
    args = types.SimpleNamespace(gpus=2, share_gpu=False)
    with pytest.raises(SystemExit):                                   # two ranks on one device
        mod.check_topology(args, FakeComm(2, 1, "gloo"))
    args = types.SimpleNamespace(gpus=2, share_gpu=True)
    topo = mod.check_topology(args, FakeComm(2, 1, "gloo"))
    assert topo == {"backend": "gloo", "ranks": 2, "distinct_gpus": 1, "rccl_ranks": 0}
    args = types.SimpleNamespace(gpus=8, share_gpu=False)
    assert mod.check_topology(args, FakeComm(8, 8, "nccl"))["rccl_ranks"] == 8


def test_visible_gpus_counts_sysfs_nodes_without_the_hip_runtime(tmp_path):
    """The launcher parent counts GPUs in the kernel driver's topology tree (nodes with SIMDs; CPU
    nodes have none) and applies the visibility variables; it never calls into HIP."""
    from distributed_tensorflow_amd import launch

    nodes = tmp_path / "nodes"
    for i, simd in enumerate([0, 0, 1024, 1024, 1024, 1024]):      # two CPU sockets, four GPUs
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
    assert launch.visible_gpus({}, str(nodes)) == 4
    assert launch.visible_gpus({"HIP_VISIBLE_DEVICES": "0,2"}, str(nodes)) == 2
    assert launch.visible_gpus({"ROCR_VISIBLE_DEVICES": "1,-1,3"}, str(nodes)) == 1      # the list ends at -1
    assert launch.visible_gpus({"HIP_VISIBLE_DEVICES": ""}, str(nodes)) == 0
    assert launch.visible_gpus({"HIP_VISIBLE_DEVICES": "0,1,2,3,4,5,6,7"}, str(nodes)) == 4
    assert launch.visible_gpus({}, str(tmp_path / "missing")) == 0                        # no amdgpu driver
    # and counting leaves torch's GPU runtime untouched in the parent
    code = "import torch; from distributed_tensorflow_amd import launch; launch.visible_gpus(); print(torch.cuda.is_initialized())"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.stdout.strip() == "False", (r.stdout, r.stderr)


def test_spawn_local_ranks_reports_every_exit_status(tmp_path):
    """A rank that ignores SIGTERM is killed AND reaped: the exit-code list never holds None."""
    from distributed_tensorflow_amd import launch

    script = tmp_path / "r.py"
    script.write_text(
        "import os, signal, sys, time\n"
        "r = int(os.environ['RANK'])\n"
        "if r == 0: sys.exit(7)\n"
        "signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
        "time.sleep(600)\n")
    import io
    import contextlib

    err = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stderr(err):
        code = launch.spawn_local_ranks([sys.executable, str(script)], 2, devices=1)
    assert code == 7 and time.time() - t0 < 60
    assert "None" not in err.getvalue() and "exit codes by rank [7, -9]" in err.getvalue()
