"""TensorBoard event files (scalar and histogram summaries) without TensorFlow.

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
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

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


_DBL_MAX = float(np.finfo(np.float64).max)
_LIMITS: Optional[np.ndarray] = None


def default_bucket_limits() -> np.ndarray:
    """Bucket limits of tf.summary.histogram: +-1e-12 * 1.1^k up to 1e20, zero, and +-DBL_MAX."""
    global _LIMITS
    if _LIMITS is None:
        pos = []
        v = 1.0e-12
        while v < 1.0e20:
            pos.append(v)
            v *= 1.1
        pos.append(_DBL_MAX)
        _LIMITS = np.array([-x for x in reversed(pos)] + [0.0] + pos, dtype=np.float64)
    return _LIMITS


def compress_histogram(limits: np.ndarray, counts: np.ndarray) -> Tuple[List[float], List[float]]:
    """Runs of empty buckets collapse into one entry carrying the last limit of the run, as in
    TensorFlow's Histogram::EncodeToProto."""
    out_l, out_c = [], []
    i, n = 0, len(counts)
    while i < n:
        end, count = float(limits[i]), float(counts[i])
        i += 1
        if count <= 0.0:
            while i < n and counts[i] <= 0.0:
                end, count = float(limits[i]), float(counts[i])
                i += 1
        out_l.append(end)
        out_c.append(count)
    return out_l, out_c


def histogram_of(values: np.ndarray) -> Dict:
    """Host statement of the histogram summary (what the device kernel is tested against)."""
    v = np.asarray(values, dtype=np.float64).ravel()
    limits = default_bucket_limits()
    idx = np.minimum(np.searchsorted(limits, v, side="right"), len(limits) - 1)
    counts = np.bincount(idx, minlength=len(limits)).astype(np.float64)
    bl, bc = compress_histogram(limits, counts)
    return {"min": float(v.min()), "max": float(v.max()), "num": float(v.size), "sum": float(v.sum()),
            "sum_squares": float((v * v).sum()), "bucket_limit": bl, "bucket": bc}


def _double_field(field: int, v: float) -> bytes:
    return _varint((field << 3) | 1) + struct.pack("<d", float(v))


def encode_histogram_proto(h: Dict) -> bytes:
    body = b"".join(_double_field(i + 1, h[k]) for i, k in enumerate(("min", "max", "num", "sum", "sum_squares")))
    body += _field_bytes(6, struct.pack("<%dd" % len(h["bucket_limit"]), *h["bucket_limit"]))
    body += _field_bytes(7, struct.pack("<%dd" % len(h["bucket"]), *h["bucket"]))
    return body


def encode_histogram_event(wall_time: float, step: int, histograms: Dict[str, Dict]) -> bytes:
    summary = b""
    for tag, h in histograms.items():
        summary += _field_bytes(1, _field_bytes(1, tag.encode()) + _field_bytes(5, encode_histogram_proto(h)))
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

    def add_histograms(self, histograms: Dict[str, Dict], global_step: int) -> None:
        """histograms: tag -> dict(min, max, num, sum, sum_squares, bucket_limit, bucket)."""
        self._f.write(tf_record(encode_histogram_event(time.time(), int(global_step), histograms)))
        self._f.flush()

    def close(self) -> None:
        self._f.close()


def _read_varint(buf: bytes, p: int) -> Tuple[int, int]:
    v = shift = 0
    while True:
        b = buf[p]
        p += 1
        v |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return v, p


def _fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    """(field, wire type, value) of a serialised proto; length-delimited values come as bytes."""
    p = 0
    while p < len(buf):
        key, p = _read_varint(buf, p)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, p = _read_varint(buf, p)
        elif wt == 1:
            v = buf[p:p + 8]
            p += 8
        elif wt == 5:
            v = buf[p:p + 4]
            p += 4
        elif wt == 2:
            n, p = _read_varint(buf, p)
            v = buf[p:p + n]
            p += n
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield field, wt, v


def _records(path: str) -> Iterator[bytes]:
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
        yield payload


def read_histograms(path: str) -> Iterator[Tuple[int, Dict[str, Dict]]]:
    """Parses the histogram events written by FileWriter.add_histograms."""
    names = ("min", "max", "num", "sum", "sum_squares")
    for payload in _records(path):
        step, out = 0, {}
        for field, wt, v in _fields(payload):
            if field == 2 and wt == 0:
                step = v
            elif field == 5 and wt == 2:
                for f2, w2, value in _fields(v):
                    if f2 != 1 or w2 != 2:
                        continue
                    tag, histo = None, None
                    for f3, w3, v3 in _fields(value):
                        if f3 == 1 and w3 == 2:
                            tag = v3.decode()
                        elif f3 == 5 and w3 == 2:
                            histo = {}
                            for f4, w4, v4 in _fields(v3):
                                if w4 == 1 and 1 <= f4 <= 5:
                                    histo[names[f4 - 1]] = struct.unpack("<d", v4)[0]
                                elif w4 == 2 and f4 in (6, 7):
                                    histo["bucket_limit" if f4 == 6 else "bucket"] = list(struct.unpack("<%dd" % (len(v4) // 8), v4))
                    if tag is not None and histo is not None:
                        out[tag] = histo
        if out:
            yield step, out


def read_events(path: str) -> Iterator[Tuple[int, Dict[str, float]]]:
    """Parses the scalar events written above (used by tests and tools/benchmark.py)."""
    for payload in _records(path):
        step, scalars = 0, {}
        for field, wt, v in _fields(payload):
            if field == 2 and wt == 0:
                step = v
            elif field == 5 and wt == 2:
                for f2, w2, value in _fields(v):
                    if f2 != 1 or w2 != 2:
                        continue
                    name, sv = None, None
                    for f3, w3, v3 in _fields(value):
                        if f3 == 1 and w3 == 2:
                            name = v3.decode()
                        elif f3 == 2 and w3 == 5:
                            (sv,) = struct.unpack("<f", v3)
                    if name is not None and sv is not None:
                        scalars[name] = sv
        if scalars:
            yield step, scalars
