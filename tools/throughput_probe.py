import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import planer_amd
from planer_amd.irgen import resnet18
ctx = planer_amd.hip.context()
B = int(os.environ.get("BATCH", "32"))
STEPS = int(os.environ.get("STEPS", "150"))
g, blob = resnet18.build()
xs = [planer_amd.asarray(np.random.default_rng(1 + i).standard_normal((B, 3, 224, 224)).astype(np.float32), ctx=ctx) for i in range(2)]
net = planer_amd.from_graph(g, blob); net.streams = os.environ.get("STREAMS", "pipe3")
plan = net.compile(xs[0], mode="throughput")
best = 0
for rep in range(4):
    for i in range(10):
        plan.feed([xs[i & 1]]); plan.launch(join=False)
    plan.join(); ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        plan.feed([xs[i & 1]]); plan.launch(join=False)
    plan.join(); ctx.synchronize()
    best = max(best, B * STEPS / (time.perf_counter() - t0))
print("%s: %.0f img/s" % (os.environ.get("TAG", ""), best))
