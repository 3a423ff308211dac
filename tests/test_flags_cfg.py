"""Flag surface and experiment-cfg parsing (reference flag names/defaults and cfg files)."""
import glob
import os

import pytest

from distributed_tensorflow_amd import cfg as cfg_mod
from distributed_tensorflow_amd.flags import FLAGS

REF = "/root/reference"


@pytest.fixture(autouse=True)
def _reset_flags():
    FLAGS.reset()
    yield
    FLAGS.reset()


def test_reference_flag_defaults():
    FLAGS.parse([])
    expect = dict(
        worker_times_cdf_method=False, interval_method=False, backup_worker_method=False, soft_kill_method=False,
        should_summarize=False, timeline_logging=False, job_name="", ps_hosts="", worker_hosts="",
        train_dir="/tmp/imagenet_train", rpc_port=1235, max_steps=1000000, batch_size=128, subset="train",
        log_device_placement=False, task_id=0, num_replicas_to_aggregate=-1, save_interval_secs=20,
        save_summaries_secs=300, initial_learning_rate=0.1, num_epochs_per_decay=2.0, learning_rate_decay_factor=0.999,
        interval_ms=1000, should_stop_worker_num=2, num_worker_kill=3, random_seed=142, num_of_instances_cifar10=50000,
        eval_dir="/tmp/imagenet_eval", checkpoint_dir="/tmp/imagenet_train", eval_interval_secs=1, run_once=False,
        num_residual_blocks=18, weight_decay=0.0002, padding_size=2, train_steps=80000, is_full_validation=True,
        train_batch_size=128, validation_batch_size=250, test_batch_size=125, init_lr=0.1, lr_decay_factor=0.1,
        decay_step0=40000, decay_step1=60000, is_use_ckpt=False,
    )
    for k, v in expect.items():
        assert getattr(FLAGS, k) == v, k


def test_flag_spellings_of_the_cfg_files():
    rest = FLAGS.parse(["--interval_method=true", "--backup_worker_method=True", "--soft_kill_method", "false",
                        "--batch_size=256", "--initial_learning_rate", ".0008", "--worker_hosts='a:1,b:2'",
                        "--nolog_device_placement", "--unknown_flag=3", "positional"])
    assert FLAGS.interval_method is True and FLAGS.backup_worker_method is True and FLAGS.soft_kill_method is False
    assert FLAGS.batch_size == 256 and abs(FLAGS.initial_learning_rate - 0.0008) < 1e-12
    assert FLAGS.log_device_placement is False
    assert rest == ["--unknown_flag=3", "positional"]
    assert FLAGS.is_present("batch_size") and not FLAGS.is_present("max_steps")


def test_cfg_interpolation_and_placeholders():
    c = cfg_mod.parse_cfg_text('''Cfg({
        "name": "x",            # comment
        "nfs_mount_point": "/shared", "base_out_dir": "%(nfs_mount_point)s/%(name)s",
        "n_masters": 1, "n_workers": 3, "num_replicas_to_aggregate": "3", "batch_size": "64",
        "train_commands": ["python src/mnist_distributed_train.py --batch_size=%(batch_size)s "
                           "--train_dir=%(base_out_dir)s/train_dir --worker_hosts='WORKER_HOSTS' --task_id=TASK_ID "
                           "--interval_method=true --job_name=JOB_NAME > %(base_out_dir)s/out_ROLE_ID 2>&1 &"],
    })''')
    assert c["base_out_dir"] == "/shared/x"
    assert c.num_workers == 4 and c.replicas_to_aggregate == 3
    flags = cfg_mod.train_flags(c)
    assert flags["batch_size"] == "64" and flags["train_dir"] == "/shared/x/train_dir" and flags["interval_method"] == "true"
    cmd = cfg_mod.substitute_placeholders(c["train_commands"][0], ps_hosts="p:1", worker_hosts="w:1,w:2", task_id=2,
                                          job_name="worker", role_id="worker1")
    assert "--worker_hosts='w:1,w:2'" in cmd and "--task_id=2" in cmd and "out_worker1" in cmd
    with pytest.raises(ValueError):
        cfg_mod.parse_cfg_text("__import__('os').system('true')")
    with pytest.raises(KeyError):
        cfg_mod.Cfg({"a": "%(missing)s"})["a"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
def test_every_reference_cfg_file_parses_and_expands():
    files = sorted(glob.glob(os.path.join(REF, "DistributedLeNet/cfg/*/*")) + glob.glob(os.path.join(REF, "Distributed_test/cfg/*/*")))
    assert len(files) >= 38
    for f in files:
        c = cfg_mod.load_cfg_from_file(f)
        flags = cfg_mod.train_flags(c)
        assert c["base_out_dir"].endswith(c["name"])
        assert flags["batch_size"] == c["batch_size"]
        assert "%(" not in " ".join(c["train_commands"])
        name = os.path.basename(f)
        if "interval" in name:
            assert flags["interval_method"] == "true" and int(flags["interval_ms"]) == int(name.split("_")[2])
        if "aggregate_sync" in name:
            assert int(flags["num_replicas_to_aggregate"]) == int(name.split("_")[2])
        if "time_cdf" in f:
            assert flags["worker_times_cdf_method"] == "true"
        FLAGS.reset()
        flags.update(task_id="3", job_name="worker", worker_hosts="a:1,b:2", ps_hosts="c:3")  # launch-time placeholders
        rest = FLAGS.parse([f"--{k}={v}" for k, v in flags.items()])
        assert rest == [], (f, rest)
        ev = cfg_mod.evaluate_flags(c)
        assert ev["checkpoint_dir"].endswith("train_dir")


def test_own_cfg_files_parse_and_expand():
    """The cfg files generated for the 8-GPU node (tools/make_cfgs.py) load through the same Cfg
    machinery as the reference's files and expand to flags the entry scripts accept."""
    import glob

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [f for f in glob.glob(os.path.join(root, "Distributed*", "cfg", "**", "*"), recursive=True) if os.path.isfile(f)]
    assert len(files) >= 20
    for f in files:
        c = cfg_mod.load_cfg_from_file(f)
        flags = cfg_mod.train_flags(c)
        assert flags["train_dir"].endswith("/train_dir"), f
        assert int(flags["batch_size"]) > 0
        assert c["base_out_dir"] == "runs/" + c["name"]


def test_default_grad_buckets_follow_the_gradient_size():
    from distributed_tensorflow_amd.models.lenet import lenet_param_specs
    from distributed_tensorflow_amd.models.resnet import resnet_param_specs
    from distributed_tensorflow_amd.parallel.comm import default_grad_buckets

    size = lambda specs: sum(sp.numel for sp in specs)
    assert default_grad_buckets(size(resnet_param_specs(3))) == 1      # ResNet-20: 1.1 MB, latency bound
    assert default_grad_buckets(size(resnet_param_specs(5))) == 1      # ResNet-32: 1.9 MB
    assert default_grad_buckets(size(resnet_param_specs(18))) == 3     # ResNet-110: 6.9 MB
    assert default_grad_buckets(size(lenet_param_specs())) == 3        # LeNet: 13 MB (its program has two segments at most)
    assert default_grad_buckets((4 << 20) // 4) == 1 and default_grad_buckets((4 << 20) // 4 + 1) == 3


def test_worker_hosts_on_several_machines_give_local_ranks(monkeypatch):
    """Reference-style command lines (--worker_hosts=h1:p,h2:p --task_id=K): a task's GPU index is its
    position among the workers of its own machine, and LOCAL_WORLD_SIZE < WORLD_SIZE tells the driver
    to coordinate through the process group's store."""
    import os

    from distributed_tensorflow_amd.cluster import resolve_task

    t = resolve_task("worker", 3, "a:2222", "a:2230,a:2231,b:2230,b:2231,b:2232")
    assert (t.local_rank, t.local_world, t.num_workers, t.master_addr) == (1, 3, 5, "a")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    t.export()
    assert os.environ["LOCAL_RANK"] == "1" and os.environ["LOCAL_WORLD_SIZE"] == "3" and os.environ["WORLD_SIZE"] == "5"
    one = resolve_task("worker", 1, "", "localhost:2230,127.0.0.1:2231")     # two names of one machine
    assert (one.local_rank, one.local_world) == (1, 2)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    assert resolve_task("ps", 0, "a:2222", "a:2230,b:2230").local_world == -1
