#!/usr/bin/env python
"""Does bringing RCCL up before the pipelined plan is compiled shift the replicas' hardware-queue slots?  A communicator of
world size 1 is created the way bench.py --gpus N does (planer_amd.dist.init), with (default) or without
(PLANER_HIP_RESERVE_STREAMS=0) the side streams reserved first; then the ResNet-18 pipeline's steady-state rate.

    [RCCL=1] [PLANER_HIP_RESERVE_STREAMS=0] STREAMS=pipe7 python tools/rccl_slot_probe.py
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import planer_amd
from planer_amd import hip, dist
from planer_amd.irgen import resnet18

ctx = hip.context()
comm = None
if os.environ.get("RCCL") == "1":
    hip.reserve_side_contexts(ctx.device, int(os.environ.get("PLANER_HIP_RESERVE_STREAMS", "14")))
    comm = dist.RcclCommunicator(ctx, 0, 1, rdzv_path="/tmp/planer_amd_rdzv_probe_%d" % os.getpid())
    comm.barrier()
B, STEPS = 32, 150
g, blob = resnet18.build()
xs = [planer_amd.asarray(np.random.default_rng(1 + i).standard_normal((B, 3, 224, 224)).astype(np.float32), ctx=ctx) for i in range(2)]
net = planer_amd.from_graph(g, blob); net.streams = os.environ.get("STREAMS", "pipe7")
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
print("%s RCCL=%s reserve=%s: %.0f img/s" % (net.streams, os.environ.get("RCCL", "0"), os.environ.get("PLANER_HIP_RESERVE_STREAMS", "14"), best))
