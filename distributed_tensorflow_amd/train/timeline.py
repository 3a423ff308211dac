"""Per-step timeline traces in chrome://tracing format.

With `--timeline_logging` the reference runs every step under TensorFlow's FULL_TRACE and writes
`<train_dir>/worker=<id>_timeline_iter=<step>.json`, one event per executed op (reference:
DistributedLeNet/src/distributed_train.py:300-305,324-328). Here a step is a recorded program of
HIP kernels, and a traced step replays it launch by launch with a pair of timing events around
every launch (`Program::run_traced`, csrc/runtime/program.cc). The JSON holds, per worker:

  pid 0  host        one span per host-side phase (the whole step, the aggregation call, ...)
  pid 1  GPU stream  tid 0: phase spans measured with events on the replica's stream - the policy's
                     spans ("forward+backward", "gradient aggregation", "optimizer") and the step
                     program's own phases ("prologue", "forward", "backward", "optimizer");
                     tid 1 + lane: one event per kernel / memset / copy of the step program, named
                     after the kernel, with its grid and block size as arguments.

A traced step is slower than a normal one (no hipGraph replay, one blocking call per segment); it
is what the flag asks for, exactly as FULL_TRACE is in the reference. Every span is also emitted as
a roctx range (torch.cuda.nvtx is roctx on ROCm), so `rocprofv3 --marker-trace` shows the same
phase names next to the kernels of an untraced run.
"""
from __future__ import annotations

import json
import os
import time
from contextlib import contextmanager
from typing import Dict, List, Optional, Sequence, Tuple

import torch


def _roctx(push: bool, name: str = "") -> None:
    try:
        if push:
            torch.cuda.nvtx.range_push(name)
        else:
            torch.cuda.nvtx.range_pop()
    except Exception:   # a torch build without the marker library: spans still work
        pass


KernelEvent = Tuple[str, int, str, float, float, int, int]   # name, lane, phase, start us, duration us, blocks, threads


class StepTimeline:
    def __init__(self, worker_id: int, enabled: bool = True, use_cuda: Optional[bool] = None):
        self.worker_id = worker_id
        self.enabled = enabled
        self.use_cuda = torch.cuda.is_available() if use_cuda is None else use_cuda
        self._host_spans: List[Dict] = []
        self._gpu_spans: List = []
        self._kernels: List[KernelEvent] = []
        self._base: Optional["torch.cuda.Event"] = None
        self._t0 = time.perf_counter()

    def reset(self, stream: Optional["torch.cuda.Stream"] = None) -> None:
        """Starts a new step. With a stream, records the origin of the step's GPU time axis on it."""
        self._host_spans.clear()
        self._gpu_spans.clear()
        self._kernels.clear()
        self._base = None
        self._t0 = time.perf_counter()
        if self.enabled and self.use_cuda and stream is not None:
            self._base = torch.cuda.Event(enable_timing=True)
            self._base.record(stream)

    @property
    def base_event_handle(self) -> int:
        """Raw hipEvent of the step's time origin (0 when there is none), for `run_traced`."""
        return int(self._base.cuda_event) if self._base is not None else 0

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

    def add_kernel_events(self, events: Sequence[KernelEvent]) -> None:
        """Per-launch events of a traced segment (`StepProgramBase.run_traced`), already on this
        step's time axis when the run was given `base_event_handle`."""
        if self.enabled:
            self._kernels.extend(events)

    @property
    def num_kernel_events(self) -> int:
        return len(self._kernels)

    def chrome_trace(self) -> str:
        events = [
            {"name": "process_name", "ph": "M", "pid": 0, "args": {"name": f"worker {self.worker_id} host"}},
            {"name": "process_name", "ph": "M", "pid": 1, "args": {"name": f"worker {self.worker_id} MI355X"}},
            {"name": "thread_name", "ph": "M", "pid": 1, "tid": 0, "args": {"name": "phases"}},
        ]
        for s in self._host_spans:
            events.append({"name": s["name"], "cat": "host", "ph": "X", "pid": 0, "tid": 0, "ts": s["ts"], "dur": s["dur"]})
        if self._gpu_spans:
            torch.cuda.synchronize()
            base = self._base if self._base is not None else self._gpu_spans[0][1]
            for name, e0, e1 in self._gpu_spans:
                events.append({"name": name, "cat": "gpu", "ph": "X", "pid": 1, "tid": 0,
                               "ts": base.elapsed_time(e0) * 1e3, "dur": e0.elapsed_time(e1) * 1e3})
        if self._kernels:
            # the step program's own phases: from the first launch of a phase to the end of its last
            spans: Dict[str, List[float]] = {}
            for name, lane, phase, ts, dur, blocks, threads in self._kernels:
                if phase:
                    lo_hi = spans.setdefault(phase, [ts, ts + dur])
                    lo_hi[0] = min(lo_hi[0], ts)
                    lo_hi[1] = max(lo_hi[1], ts + dur)
            for phase, (lo, hi) in spans.items():
                events.append({"name": phase, "cat": "phase", "ph": "X", "pid": 1, "tid": 0, "ts": lo, "dur": hi - lo})
            for lane in sorted({k[1] for k in self._kernels}):
                events.append({"name": "thread_name", "ph": "M", "pid": 1, "tid": 1 + lane,
                               "args": {"name": "kernels" if lane == 0 else f"kernels (side stream {lane})"}})
            for name, lane, phase, ts, dur, blocks, threads in self._kernels:
                events.append({"name": name, "cat": "kernel", "ph": "X", "pid": 1, "tid": 1 + lane, "ts": ts, "dur": dur,
                               "args": {"phase": phase, "blocks": blocks, "threads_per_block": threads}})
        return json.dumps({"traceEvents": events}, indent=1)

    def write(self, train_dir: str, step: int) -> str:
        path = os.path.join(train_dir, "worker=%d_timeline_iter=%d.json" % (self.worker_id, step))
        with open(path, "w") as f:
            f.write(self.chrome_trace())
        return path
