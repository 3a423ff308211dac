"""`model.ckpt-N.meta`: the MetaGraphDef file that TensorFlow's Saver writes next to every bundle.

The reference's checkpoints consist of `.index`, `.data-00000-of-00001` and `.meta` plus the
`checkpoint` state file (tf.train.Saver under a Supervisor; reference:
DistributedLeNet/src/distributed_train.py:212,234-242,370-373). The `.meta` file is a serialized
`MetaGraphDef` protobuf. There is no TensorFlow graph here, so the file carries what a checkpoint
reader can use: one `VariableV2` node per saved variable (name, dtype, shape - exactly the names in
the bundle), the `variables` / `trainable_variables` collections as serialized `VariableDef`s, and
the `SaverDef` (V2 format, max_to_keep) with the tensor names TensorFlow's Saver uses. The encoder
below writes the protobuf wire format directly (field numbers from tensorflow/core/protobuf/
meta_graph.proto, saver.proto, framework/graph.proto, node_def.proto, attr_value.proto,
tensor_shape.proto, variable.proto); `read_meta` decodes it again. TensorFlow is not installed in
this environment, so loading the file with `tf.train.import_meta_graph` is unpinned.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Iterable, List, Sequence, Tuple

DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9
VarInfo = Tuple[str, int, Sequence[int], bool]   # name, dtype enum, shape, trainable


# ---- protobuf wire format ------------------------------------------------------------------------
def _varint(n: int) -> bytes:
    if n < 0:
        n += 1 << 64
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _key(field: int, wire: int) -> bytes:
    return _varint((field << 3) | wire)


def _len(field: int, payload: bytes) -> bytes:
    return _key(field, 2) + _varint(len(payload)) + payload


def _str(field: int, s: str) -> bytes:
    return _len(field, s.encode())


def _int(field: int, v: int) -> bytes:
    return _key(field, 0) + _varint(v)


def _attr(name: str, value: bytes) -> bytes:
    """One entry of NodeDef.attr (map<string, AttrValue>)."""
    return _len(5, _str(1, name) + _len(2, value))


def _shape_proto(shape: Sequence[int]) -> bytes:
    return b"".join(_len(2, _int(1, int(d))) for d in shape)      # TensorShapeProto.dim { size }


def _variable_node(name: str, dtype: int, shape: Sequence[int]) -> bytes:
    node = _str(1, name) + _str(2, "VariableV2")
    node += _attr("container", _len(2, b""))                      # AttrValue.s
    node += _attr("dtype", _int(6, dtype))                        # AttrValue.type
    node += _attr("shape", _len(7, _shape_proto(shape)))          # AttrValue.shape
    node += _attr("shared_name", _len(2, b""))
    return node


def _variable_def(name: str) -> bytes:
    return _str(1, name + ":0") + _str(2, name + "/Assign") + _str(3, name + "/read:0")


def _collection(key: str, values: Iterable[bytes]) -> bytes:
    bytes_list = b"".join(_len(1, v) for v in values)             # BytesList.value
    return _len(4, _str(1, key) + _len(2, _len(2, bytes_list)))   # map entry: key, CollectionDef.bytes_list


def build_meta_graph(variables: Sequence[VarInfo], max_to_keep: int = 5, producer: str = "distributed_tensorflow_amd") -> bytes:
    meta_info = _str(1, "") + _str(5, producer)                   # meta_graph_version, tensorflow_version
    graph = b"".join(_len(1, _variable_node(n, dt, sh)) for n, dt, sh, _ in variables)
    graph += _len(4, _int(1, 21))                                 # versions.producer (the TF 0.12 / 1.0 graph version)
    # SaverDef: tensor names as TensorFlow's Saver creates them, keep_checkpoint_every_n_hours
    # (float, field 6) and version = V2 (field 7)
    saver = (_str(1, "save/Const:0") + _str(2, "save/control_dependency:0") + _str(3, "save/restore_all")
             + _int(4, max_to_keep) + _key(6, 5) + struct.pack("<f", 10000.0) + _int(7, 2))
    out = _len(1, meta_info) + _len(2, graph) + _len(3, saver)
    out += _collection("variables", (_variable_def(n) for n, _, _, _ in variables))
    out += _collection("trainable_variables", (_variable_def(n) for n, _, _, t in variables if t))
    return out


def write_meta(prefix: str, variables: Sequence[VarInfo], max_to_keep: int = 5) -> str:
    path = prefix + ".meta"
    with open(path + ".tmp", "wb") as f:
        f.write(build_meta_graph(variables, max_to_keep))
    os.replace(path + ".tmp", path)
    return path


# ---- decoding (tests, tools) ---------------------------------------------------------------------
def _fields(buf: bytes) -> List[Tuple[int, int, object]]:
    out, i = [], 0
    while i < len(buf):
        k, sh = 0, 0
        while True:
            b = buf[i]
            i += 1
            k |= (b & 0x7F) << sh
            sh += 7
            if not b & 0x80:
                break
        field, wire = k >> 3, k & 7
        if wire == 0:
            v, sh = 0, 0
            while True:
                b = buf[i]
                i += 1
                v |= (b & 0x7F) << sh
                sh += 7
                if not b & 0x80:
                    break
            out.append((field, wire, v))
        elif wire == 2:
            n, sh = 0, 0
            while True:
                b = buf[i]
                i += 1
                n |= (b & 0x7F) << sh
                sh += 7
                if not b & 0x80:
                    break
            out.append((field, wire, buf[i:i + n]))
            i += n
        elif wire == 5:
            out.append((field, wire, buf[i:i + 4]))
            i += 4
        elif wire == 1:
            out.append((field, wire, buf[i:i + 8]))
            i += 8
        else:
            raise ValueError(f"unsupported wire type {wire}")
    return out


def read_meta(path: str) -> Dict:
    """{'variables': {name: (dtype, shape)}, 'collections': {key: [variable names]}, 'saver': {...}}"""
    top = _fields(open(path, "rb").read())
    res: Dict = {"variables": {}, "collections": {}, "saver": {}}
    for field, _, val in top:
        if field == 2:
            for f, _, node in _fields(val):
                if f != 1:
                    continue
                name, op, dtype, shape = "", "", 0, []
                for nf, _, nv in _fields(node):
                    if nf == 1:
                        name = nv.decode()
                    elif nf == 2:
                        op = nv.decode()
                    elif nf == 5:
                        entry = dict((a, b) for a, _, b in _fields(nv))
                        key, av = entry[1].decode(), _fields(entry[2])
                        for af, _, avv in av:
                            if key == "dtype" and af == 6:
                                dtype = avv
                            if key == "shape" and af == 7:
                                shape = [dict((a, b) for a, _, b in _fields(d))[1] for df, _, d in _fields(avv) if df == 2]
                if op == "VariableV2":
                    res["variables"][name] = (dtype, shape)
        elif field == 3:
            for f, w, v in _fields(val):
                res["saver"][f] = v.decode() if w == 2 else v
        elif field == 4:
            entry = dict((a, b) for a, _, b in _fields(val))
            names = []
            for cf, _, cv in _fields(entry[2]):
                if cf == 2:
                    for bf, _, bv in _fields(cv):
                        names.append(dict((a, b) for a, _, b in _fields(bv))[1].decode().rsplit(":", 1)[0])
            res["collections"][entry[1].decode()] = names
    return res
