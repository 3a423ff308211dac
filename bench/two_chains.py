"""Two half-batch ResNet-20 step programs on two streams of one GPU against one full-batch program:
does the start-up / drain of one chain's kernels hide behind the other chain's work?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.models.resnet_hip import ResNetHipStep  # noqa: E402
from distributed_tensorflow_amd.ops import reference as ref  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
STEPS = 100


def make(batch, stream):
    with torch.cuda.stream(stream):
        st = ResNetHipStep(3, batch)
        st.load_batch(torch.randn(batch, 32, 32, 3, device="cuda"), torch.randint(0, 10, (batch,), device="cuda"))
        st.instantiate()
        st.set_hyper(lr_t=ref.adam_lr_t(1e-3, 1))
    return st


def timed(steps_streams):
    for _ in range(10):
        for st, s in steps_streams:
            with torch.cuda.stream(s):
                st.run(-1, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(STEPS):
        for st, s in steps_streams:
            with torch.cuda.stream(s):
                st.run(-1, True)
    torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / STEPS


s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
full = make(B, s0)
print("one chain, batch %d: %.3f ms per step" % (B, timed([(full, s0)])), flush=True)
del full
a, b = make(B // 2, s0), make(B // 2, s1)
print("two chains, batch %d each: %.3f ms per pair" % (B // 2, timed([(a, s0), (b, s1)])), flush=True)
print("one chain, batch %d: %.3f ms per step" % (B // 2, timed([(a, s0)])), flush=True)
