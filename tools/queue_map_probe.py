#!/usr/bin/env python
"""Which hardware queue a replica's stream lands on (the runtime assigns streams to its four hardware queues round robin in
creation order) -- does the pipeline's rate depend on the MAP or only on the depth?  MAP = comma list of dummy-stream counts
inserted BEFORE side stream i (i = 1..N-1), e.g. depth 6 on three queues: STREAMS=pipe6 SKIP=0,0,1,0,0 (a dummy after every
third stream skips queue 3).  LEAD = dummy streams created before the net's own.

    STREAMS=pipe6 SKIP=0,0,1,0,0 python tools/queue_map_probe.py
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import planer_amd
from planer_amd import hip
from planer_amd import net as netmod
from planer_amd.irgen import resnet18

lead = [hip.Context(0) for _ in range(int(os.environ.get("LEAD", "0")))]
ctx = hip.context()
skip = [int(v) for v in os.environ.get("SKIP", "").split(",") if v != ""]
# QMAP=0,1,0,1,...: the creation slot (mod 4) wanted for replica i; replica 0 is the net's own stream = slot LEAD
qmap = [int(v) for v in os.environ.get("QMAP", "").split(",") if v != ""]
if qmap:
    os.environ["STREAMS"] = "pipe%d" % len(qmap)
    slot = int(os.environ.get("LEAD", "0")) % 4
    assert qmap[0] == slot, "replica 0 sits in slot LEAD"
    skip = []
    for want in qmap[1:]:
        k = (want - (slot + 1)) % 4
        skip.append(k)
        slot = want
dummies = []
orig = netmod.Net._side_context
def side(self, i):
    while len(self._side) < i:
        k = len(self._side)
        for _ in range(skip[k] if k < len(skip) else 0):
            dummies.append(hip.Context(self.ctx.device))
        self._side.append(hip.Context(self.ctx.device))
    return self.ctx if i == 0 else self._side[i - 1]
netmod.Net._side_context = side
B, STEPS = 32, int(os.environ.get("STEPS", "150"))
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
print("%s LEAD=%s SKIP=%s QMAP=%s: %.0f img/s" % (net.streams, os.environ.get("LEAD", "0"), os.environ.get("SKIP", ""), os.environ.get("QMAP", ""), best))
