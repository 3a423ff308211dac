"""Per-step timeline traces in chrome://tracing format.

With `--timeline_logging` the reference runs every step under TensorFlow's FULL_TRACE and writes
`<train_dir>/worker=<id>_timeline_iter=<step>.json` (reference:
DistributedLeNet/src/distributed_train.py:300-305,324-328). Here a step is a recorded program of
HIP kernels; the trace is built from HIP events recorded around each program segment (forward +
backward pieces, the gradient collectives, the optimizer), written with the same file naming and
the same chrome-trace JSON schema. Kernel-level detail comes from running the job under
`rocprofv3 --kernel-trace`; every span is also emitted as a roctx range (torch.cuda.nvtx is roctx
on ROCm), so `rocprofv3 --marker-trace` shows the same phase names next to the kernels.
"""
from __future__ import annotations

import json
import os
import time
from contextlib import contextmanager
from typing import Dict, List, Optional

import torch


def _roctx(push: bool, name: str = "") -> None:
    try:
        if push:
            torch.cuda.nvtx.range_push(name)
        else:
            torch.cuda.nvtx.range_pop()
    except Exception:   # a torch build without the marker library: spans still work
        pass


class StepTimeline:
    def __init__(self, worker_id: int, enabled: bool = True, use_cuda: Optional[bool] = None):
        self.worker_id = worker_id
        self.enabled = enabled
        self.use_cuda = torch.cuda.is_available() if use_cuda is None else use_cuda
        self._host_spans: List[Dict] = []
        self._gpu_spans: List = []
        self._t0 = time.perf_counter()

    def reset(self) -> None:
        self._host_spans.clear()
        self._gpu_spans.clear()
        self._t0 = time.perf_counter()

    @contextmanager
    def span(self, name: str, stream: Optional["torch.cuda.Stream"] = None):
        """Times a phase on the host and, when a stream is given, on the device."""
        if not self.enabled:
            yield
            return
        h0 = time.perf_counter()
        ev = None
        if self.use_cuda:
            _roctx(True, name)
        if self.use_cuda and stream is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(stream)
        try:
            yield
        finally:
            if self.use_cuda:
                _roctx(False)
            if ev is not None:
                ev[1].record(stream)
                self._gpu_spans.append((name, ev[0], ev[1]))
            self._host_spans.append({"name": name, "ts": (h0 - self._t0) * 1e6, "dur": (time.perf_counter() - h0) * 1e6})

    def chrome_trace(self) -> str:
        events = [
            {"name": "process_name", "ph": "M", "pid": 0, "args": {"name": f"worker {self.worker_id} host"}},
            {"name": "process_name", "ph": "M", "pid": 1, "args": {"name": f"worker {self.worker_id} MI355X stream"}},
        ]
        for s in self._host_spans:
            events.append({"name": s["name"], "cat": "host", "ph": "X", "pid": 0, "tid": 0, "ts": s["ts"], "dur": s["dur"]})
        if self._gpu_spans:
            torch.cuda.synchronize()
            base = self._gpu_spans[0][1]
            for name, e0, e1 in self._gpu_spans:
                events.append({"name": name, "cat": "gpu", "ph": "X", "pid": 1, "tid": 0,
                               "ts": base.elapsed_time(e0) * 1e3, "dur": e0.elapsed_time(e1) * 1e3})
        return json.dumps({"traceEvents": events}, indent=1)

    def write(self, train_dir: str, step: int) -> str:
        path = os.path.join(train_dir, "worker=%d_timeline_iter=%d.json" % (self.worker_id, step))
        with open(path, "w") as f:
            f.write(self.chrome_trace())
        return path
