"""TensorBoard event files (scalar summaries) without TensorFlow.

The reference's chief writes summaries through the Supervisor and its evaluator writes
`Validation Accuracy` / `Validation Loss` scalars with a FileWriter (reference:
DistributedLeNet/src/nn_eval.py:107-110, DistributedLeNet/src/distributed_train.py:347-355,
DistributedResNet/src/resnet.py:13-20). An event file is a TFRecord stream
(len u64 | masked crc32c(len) | payload | masked crc32c(payload)) of Event protos; the few
fields needed are encoded by hand and the checksums come from the native CRC32C.
"""
from __future__ import annotations

import os
import socket
import struct
import time
from typing import Dict, Iterator, List, Tuple

from .. import _host


def _varint(v: int) -> bytes:
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _field_bytes(field: int, payload: bytes) -> bytes:
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _masked_crc(data: bytes) -> int:
    return _host.crc32c_mask(_host.crc32c(data))


def encode_scalar_event(wall_time: float, step: int, scalars: Dict[str, float]) -> bytes:
    summary = b""
    for tag, val in scalars.items():
        value = _field_bytes(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack("<f", float(val))
        summary += _field_bytes(1, value)
    ev = _varint((1 << 3) | 1) + struct.pack("<d", wall_time)
    ev += _varint((2 << 3) | 0) + _varint(step)
    ev += _field_bytes(5, summary)
    return ev


def encode_version_event(wall_time: float) -> bytes:
    return _varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _field_bytes(3, b"brain.Event:2")


def tf_record(payload: bytes) -> bytes:
    header = struct.pack("<Q", len(payload))
    return header + struct.pack("<I", _masked_crc(header)) + payload + struct.pack("<I", _masked_crc(payload))


class FileWriter:
    """tf.summary.FileWriter look-alike for scalar summaries."""

    def __init__(self, logdir: str):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "events.out.tfevents.%010d.%s" % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, "ab")
        self._f.write(tf_record(encode_version_event(time.time())))
        self._f.flush()

    def add_scalars(self, scalars: Dict[str, float], global_step: int) -> None:
        self._f.write(tf_record(encode_scalar_event(time.time(), int(global_step), scalars)))
        self._f.flush()

    def close(self) -> None:
        self._f.close()


def read_events(path: str) -> Iterator[Tuple[int, Dict[str, float]]]:
    """Parses the scalar events written above (used by tests and tools/benchmark.py)."""
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos + 12 <= len(data):
        (n,) = struct.unpack_from("<Q", data, pos)
        (crc_len,) = struct.unpack_from("<I", data, pos + 8)
        if crc_len != _masked_crc(data[pos:pos + 8]):
            raise ValueError("event file: corrupt length checksum")
        payload = data[pos + 12:pos + 12 + n]
        (crc_pl,) = struct.unpack_from("<I", data, pos + 12 + n)
        if crc_pl != _masked_crc(payload):
            raise ValueError("event file: corrupt payload checksum")
        pos += 16 + n
        step, scalars = 0, {}
        p = 0
        while p < len(payload):
            tag = payload[p]
            p += 1
            field, wt = tag >> 3, tag & 7
            if wt == 1:
                p += 8
            elif wt == 0:
                v = 0
                shift = 0
                while True:
                    b = payload[p]
                    p += 1
                    v |= (b & 0x7F) << shift
                    shift += 7
                    if not b & 0x80:
                        break
                if field == 2:
                    step = v
            elif wt == 2:
                ln = 0
                shift = 0
                while True:
                    b = payload[p]
                    p += 1
                    ln |= (b & 0x7F) << shift
                    shift += 7
                    if not b & 0x80:
                        break
                body = payload[p:p + ln]
                p += ln
                if field == 5:
                    scalars.update(_parse_summary(body))
            elif wt == 5:
                p += 4
        if scalars:
            yield step, scalars


def _parse_summary(body: bytes) -> Dict[str, float]:
    out = {}
    p = 0
    while p < len(body):
        tag = body[p]
        p += 1
        ln = body[p]
        p += 1
        if ln & 0x80:
            ln = (ln & 0x7F) | (body[p] << 7)
            p += 1
        val = body[p:p + ln]
        p += ln
        if tag >> 3 != 1:
            continue
        q = 0
        name, sv = None, None
        while q < len(val):
            t = val[q]
            q += 1
            if t == ((1 << 3) | 2):
                l2 = val[q]
                q += 1
                name = val[q:q + l2].decode()
                q += l2
            elif t == ((2 << 3) | 5):
                (sv,) = struct.unpack_from("<f", val, q)
                q += 4
            else:
                break
        if name is not None and sv is not None:
            out[name] = sv
    return out
