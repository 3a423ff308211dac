"""TF tensor-bundle checkpoints: format golden bytes, round trips, TensorFlow variable naming."""
import os
import struct

import numpy as np
import pytest
import torch

from distributed_tensorflow_amd import _host
from distributed_tensorflow_amd.checkpoint.saver import (Saver, get_checkpoint_state, latest_checkpoint, read_bundle,
                                                         step_of, write_bundle)
from distributed_tensorflow_amd.models.lenet import lenet_param_specs
from distributed_tensorflow_amd.models.resnet import resnet_param_specs
from distributed_tensorflow_amd.runtime.arena import ParamArena

TABLE_MAGIC = 0xDB4775248B80FB57


def test_crc32c_known_vectors_and_mask():
    # RFC 3720 test vectors
    assert _host.crc32c(b"123456789") == 0xE3069283
    assert _host.crc32c(bytes(32)) == 0x8A9136AA
    assert _host.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert _host.crc32c(bytes(range(32))) == 0x46DD794E
    for v in (0, 1, 0xDEADBEEF, 0xFFFFFFFF):
        m = _host.crc32c_mask(v)
        assert m == ((((v >> 15) | (v << 17)) + 0xA282EAD8) & 0xFFFFFFFF)
        assert _host.crc32c_unmask(m) == v


def test_table_golden_bytes():
    """Byte-exact LevelDB table for a tiny key set, derived by hand from the format definition."""
    kv = [(b"", b"H"), (b"Variable", b"AAA"), (b"Variable_1", b"BB")]
    t = _host.build_table(kv)
    # data block: entries are (shared, non_shared, value_len, key_delta, value)
    block = b"\x00\x00\x01" + b"H"
    block += b"\x00\x08\x03" + b"Variable" + b"AAA"
    block += b"\x08\x02\x02" + b"_1" + b"BB"
    block += struct.pack("<II", 0, 1)            # one restart at offset 0
    assert t[: len(block)] == block
    crc = _host.crc32c_mask(_host.crc32c(block + b"\x00"))
    assert t[len(block): len(block) + 5] == b"\x00" + struct.pack("<I", crc)
    # footer: 48 bytes, ends with the table magic
    assert len(t) >= 48 and struct.unpack("<Q", t[-8:])[0] == TABLE_MAGIC
    # metaindex block directly after the data block: empty block = restarts [0], count 1
    meta_off = len(block) + 5
    assert t[meta_off: meta_off + 8] == struct.pack("<II", 0, 1)
    # index block key is the shortest successor of the last key: "W"
    idx_off = meta_off + 8 + 5
    assert t[idx_off: idx_off + 4] == b"\x00\x01\x02" + b"W"
    assert _host.parse_table(t) == kv


def test_table_many_blocks_and_prefix_compression():
    kv = [(("key%05d" % i).encode(), os.urandom(37)) for i in range(3000)]
    t = _host.build_table(kv, block_size=4096, restart_interval=16)
    assert _host.parse_table(t) == kv
    bad = bytearray(t)
    bad[100] ^= 0xFF
    with pytest.raises(RuntimeError):
        _host.parse_table(bytes(bad))
    with pytest.raises(RuntimeError):
        _host.build_table([(b"b", b"1"), (b"a", b"2")])


def test_bundle_header_and_entry_protos(tmp_path):
    prefix = str(tmp_path / "model.ckpt-3")
    write_bundle(prefix, {"global_step": np.array(3, np.int32), "w": np.arange(6, dtype=np.float32).reshape(2, 3)})
    kv = dict(_host.parse_table(open(prefix + ".index", "rb").read()))
    # BundleHeaderProto: num_shards=1, version.producer=1 (little endian is the default, omitted)
    assert kv[b""] == bytes([0x08, 0x01, 0x1A, 0x02, 0x08, 0x01])
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    assert data[:4] == struct.pack("<i", 3) and len(data) == 4 + 24
    # BundleEntryProto of the scalar: dtype DT_INT32(3), empty shape, size 4, crc32c
    crc = _host.crc32c_mask(_host.crc32c(data[:4]))
    assert kv[b"global_step"] == bytes([0x08, 0x03, 0x12, 0x00, 0x28, 0x04, 0x35]) + struct.pack("<I", crc)
    # the matrix: dtype DT_FLOAT(1), shape {dim{size 2} dim{size 3}}, offset 4, size 24
    crc = _host.crc32c_mask(_host.crc32c(data[4:]))
    assert kv[b"w"] == bytes([0x08, 0x01, 0x12, 0x08, 0x12, 0x02, 0x08, 0x02, 0x12, 0x02, 0x08, 0x03, 0x20, 0x04, 0x28, 0x18, 0x35]) + struct.pack("<I", crc)
    back = read_bundle(prefix)
    assert back["global_step"] == 3 and back["w"].tolist() == [[0, 1, 2], [3, 4, 5]]
    # a flipped data byte is caught by the per-tensor checksum
    with open(prefix + ".data-00000-of-00001", "r+b") as f:
        f.seek(10)
        f.write(b"\x7f")
    with pytest.raises(RuntimeError):
        read_bundle(prefix)


def test_lenet_checkpoint_uses_tf_auto_names_and_adam_slots(tmp_path):
    arena = ParamArena(lenet_param_specs(), "cpu", ("m", "v"))
    arena.initialize()
    arena.slots["m"].normal_()
    arena.slots["v"].uniform_()
    saver = Saver(arena, "adam")
    prefix = saver.save(str(tmp_path), 41, updates_applied=41)
    assert os.path.basename(prefix) == "model.ckpt-41" and step_of(prefix) == 41
    names = set(_host.BundleReader(prefix).names())
    expect = {"global_step", "beta1_power", "beta2_power"}
    for i in range(8):
        v = "Variable" if i == 0 else f"Variable_{i}"
        expect |= {v, v + "/Adam", v + "/Adam_1"}
    assert names == expect
    data = read_bundle(prefix)
    assert data["Variable"].shape == (5, 5, 1, 32) and data["Variable_4"].shape == (3136, 512) and data["Variable_7"].shape == (10,)
    assert abs(float(data["beta1_power"]) - 0.9 ** 42) < 1e-7 and data["global_step"].dtype == np.int32
    # restore into a fresh arena reproduces parameters and slots bit for bit
    fresh = ParamArena(lenet_param_specs(), "cpu", ("m", "v"))
    assert Saver(fresh, "adam").restore(prefix) == 41
    assert torch.equal(fresh.params, arena.params)
    for name in arena.names():       # (alignment padding between tensors is not part of a checkpoint)
        assert torch.equal(fresh.slot("m", name), arena.slot("m", name)) and torch.equal(fresh.slot("v", name), arena.slot("v", name))
    assert get_checkpoint_state(str(tmp_path))[0] == prefix and latest_checkpoint(str(tmp_path)) == prefix
    assert open(tmp_path / "checkpoint").read().splitlines()[0] == 'model_checkpoint_path: "model.ckpt-41"'


def test_resnet_name_maps_both_conventions(tmp_path):
    n = 3
    specs = resnet_param_specs(n)
    assert len(specs) == 18 * n + 5
    assert [s.tf_name for s in specs[:4]] == ["Variable", "Variable_1", "Variable_2", "Variable_3"]
    assert [s.name for s in specs[:4]] == ["conv0/conv", "conv0/beta", "conv0/gamma", "conv1_0/conv1_in_block/conv"]
    assert specs[-1].name == "fc/fc_bias" and specs[-1].tf_name == f"Variable_{18 * n + 4}"
    arena = ParamArena(specs, "cpu", ("m",))
    arena.initialize()
    for naming in ("tf_auto", "scoped"):
        d = tmp_path / naming
        saver = Saver(arena, "momentum", names=naming)
        prefix = saver.save(str(d), 7)
        names = set(_host.BundleReader(prefix).names())
        first = "Variable" if naming == "tf_auto" else "conv0/conv"
        assert first in names and first + "/Momentum" in names and "beta1_power" not in names
        fresh = ParamArena(specs, "cpu", ("m",))
        # either convention restores into either saver
        Saver(fresh, "momentum", names="scoped" if naming == "tf_auto" else "tf_auto").restore(prefix)
        assert torch.equal(fresh.params, arena.params)
    nm = Saver(arena, "adam").name_map()
    assert nm["Variable"] == (0, (3, 3, 3, 16)) and nm["Variable_3"][1] == (3, 3, 16, 16)


def test_max_to_keep_rotates_files(tmp_path):
    arena = ParamArena(lenet_param_specs()[:2], "cpu", ())
    saver = Saver(arena, "momentum", max_to_keep=2)
    for s in (1, 2, 3):
        saver.save(str(tmp_path), s)
    files = sorted(os.listdir(tmp_path))
    assert "model.ckpt-1.index" not in files and "model.ckpt-3.index" in files and "model.ckpt-2.index" in files
    latest, allp = get_checkpoint_state(str(tmp_path))
    assert latest.endswith("model.ckpt-3") and len(allp) == 2


def test_meta_graph_file_lists_the_saved_variables(tmp_path):
    """Every checkpoint comes with a `.meta` MetaGraphDef naming the same variables as the bundle
    (VariableV2 nodes with dtype and shape, the variables / trainable_variables collections, a V2
    SaverDef). TensorFlow is not importable here: the file is decoded with the module's own
    wire-format reader, loading it with tf.train.import_meta_graph is unpinned."""
    import numpy as np

    from distributed_tensorflow_amd.checkpoint.meta_graph import DT_FLOAT, DT_INT32, build_meta_graph, read_meta
    from distributed_tensorflow_amd.checkpoint.saver import Saver, bundle_names
    from distributed_tensorflow_amd.models.lenet import lenet_param_specs
    from distributed_tensorflow_amd.runtime.arena import ParamArena

    a = ParamArena(lenet_param_specs(), "cpu", ("m", "v"))
    a.initialize(1)
    prefix = Saver(a, "adam").save(str(tmp_path), 7, 7)
    meta = read_meta(prefix + ".meta")
    assert sorted(meta["variables"]) == sorted(bundle_names(prefix))
    assert meta["variables"]["Variable"] == (DT_FLOAT, [5, 5, 1, 32]) and meta["variables"]["global_step"] == (DT_INT32, [])
    assert meta["variables"]["Variable_4/Adam_1"] == (DT_FLOAT, [3136, 512])
    assert meta["collections"]["trainable_variables"] == [f"Variable_{i}" if i else "Variable" for i in range(8)]
    assert set(meta["collections"]["variables"]) == set(meta["variables"])
    assert meta["saver"][1] == "save/Const:0" and meta["saver"][4] == 5 and meta["saver"][7] == 2
    # golden bytes of a one-variable graph: field numbers and nesting as in the .proto files
    blob = build_meta_graph([("w", DT_FLOAT, [2, 3], True)], max_to_keep=1)
    assert blob[:4] == b"\n\x1e\n\x00"                                   # MetaGraphDef.meta_info_def { meta_graph_version: "" ...
    assert b"\n\x01w\x12\nVariableV2" in blob                            # NodeDef { name: "w" op: "VariableV2"
    assert b"*\x0b\n\x05dtype\x12\x020\x01" in blob                       # attr { key: "dtype" value { type: DT_FLOAT } }
    assert b"*\x13\n\x05shape\x12\n:\x08\x12\x02\x08\x02\x12\x02\x08\x03" in blob   # shape { dim { size: 2 } dim { size: 3 } }
    assert b"\n\x03w:0\x12\x08w/Assign\x1a\x08w/read:0" in blob           # VariableDef
    assert b"\n\x0csave/Const:0" in blob and blob.count(b"\"\x02\x08\x15") == 1   # SaverDef; GraphDef.versions { producer: 21 }
