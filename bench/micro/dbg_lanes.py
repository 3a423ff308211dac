import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["DTF_BWD_FUSED"] = sys.argv[1]; os.environ["DTF_SIDE_STREAM"] = sys.argv[2]
from distributed_tensorflow_amd.models.resnet_hip import ResNetHipStep
b = 16
g = torch.Generator().manual_seed(9)
images = torch.randn(b, 32, 32, 3, generator=g).cuda(); labels = torch.randint(0, 10, (b,), generator=g).cuda()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step = ResNetHipStep(2, b, num_buckets=2)
    step.load_batch(images, labels); step.set_hyper(lr_t=0.0)
    outs = []
    for mode in ("direct",) * 6 + ("graphseg",) * 8 + ("direct",) * 4:
        if mode == "graphseg" and not step.program.instantiated(0):
            step.instantiate(per_segment=True)
        if mode == "direct":
            step.run(graph=False)
        else:
            for i in range(step.num_segments): step.run(i, True)
        s.synchronize()
        outs.append(step.arena.grads.clone())
    a = step.arena
    for i in range(1, len(outs)):
        d = (outs[i] - outs[0]).abs()
        worst = int(d.argmax())
        name = [n for n in a.names() if a.offsets[n] <= worst][-1]
        per = {n: float((outs[i] - outs[0])[a.offsets[n]:a.offsets[n] + a.spec(n).numel].abs().max()) for n in a.names()}
        top = sorted(per.items(), key=lambda kv: -kv[1])[:3]
        print(i, "max abs diff vs first %.3e" % float(d.max()), "top:", [(n, "%.2e" % v) for n, v in top])
