#!/usr/bin/env python
"""ResNet-18 batch 32, host ndarray in -> host ndarray out through net.submit with six passes in flight: images/s for pageable
and pinned batches, beside the device-resident submit loop.  Environment switches select the route (PLANER_HIP_STAGED,
PLANER_HIP_COPY_THREADS, PLANER_HIP_COPY_PRIO, PLANER_HIP_COPY_CHUNK_KB); TAG labels the line.  Run on the GPU box."""
import collections
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import planer_amd  # noqa: E402
from planer_amd.irgen import resnet18  # noqa: E402

n = int(os.environ.get("BATCH", "32"))
g, b = resnet18.build()
net = planer_amd.from_graph(g, b)
xs_host = [resnet18.make_input(n, seed=s) for s in range(2)]
xs = [planer_amd.asarray(a, ctx=net.ctx) for a in xs_host]
net.compile(xs[0], mode="throughput")
window, pend = int(os.environ.get("WINDOW", "6")), collections.deque()


def loop(k, batches):
    for i in range(k):
        pend.append(net.submit(batches[i & 1]))
        if len(pend) > window:
            pend.popleft().get()
    while pend:
        pend.popleft().get()


def rate(batches, k=int(os.environ.get("STEPS", "150"))):
    loop(20, batches)
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        loop(k, batches)
        best = max(best, n * k / (time.perf_counter() - t0))
    return best


pins = [planer_amd.hip.pinned_empty(xs_host[0].shape) for _ in range(2)]
pins[0][...] = xs_host[0]
pins[1][...] = xs_host[1]
if os.environ.get("PROFILE"):
    import cProfile
    import pstats
    loop(20, xs_host)
    pr = cProfile.Profile()
    pr.enable()
    t0 = time.perf_counter()
    loop(150, xs_host)
    dt = time.perf_counter() - t0
    pr.disable()
    print("profiled pageable loop: %.0f img/s, %.3f ms per batch" % (n * 150 / dt, dt / 150 * 1e3))
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
    sys.exit(0)
r_dev, r_host, r_pin = rate(xs), rate(xs_host), rate(pins)
t0 = time.perf_counter()
for i in range(20):
    net(xs_host[i & 1])
r_call = n * 20 / (time.perf_counter() - t0)
print("%-28s device %8.0f  pageable %8.0f  pinned %8.0f  net(x_host) one at a time %8.0f img/s" % (os.environ.get("TAG", ""), r_dev, r_host, r_pin, r_call))
