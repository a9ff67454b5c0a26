#!/usr/bin/env python
"""What ONE dependent kernel costs inside a captured forward pass at batch 1: a chain of N identical layers, compiled and
replayed like any net; (time of N = 96) - (time of N = 32) over 64 layers.  The floor every layer of a batch-1 detection net pays.

    python tools/chain_cost.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import planer_amd  # noqa: E402
from planer_amd.irgen.builder import GraphBuilder  # noqa: E402


def chain(kind, n, cin, cout, k, h):
    rng = np.random.default_rng(0)
    g = GraphBuilder(["x"])
    y, c = "x", cin
    for i in range(n):
        t = "c%d" % i
        if kind == "relu":
            y = g.op("leakyrelu", y, t + "_a", name=t + "_act", alpha=0.1)
            continue
        co = cout if i % 2 == 0 else cin
        g.init(t + "_w", (rng.standard_normal((co, c, k, k)) * np.sqrt(2.0 / (c * k * k))).astype(np.float32))
        g.init(t + "_invK", rng.uniform(0.5, 1.5, (1, co, 1, 1)).astype(np.float32))
        g.init(t + "_invB", (rng.standard_normal((1, co, 1, 1)) * 0.1).astype(np.float32))
        p = k // 2
        g.op("conv", [y, t + "_w"], t + "_c", name=t + "_conv", group=1, strides=[1, 1], dilations=[1, 1], pads=[p, p, p, p])
        g.op("batchnorm", [t + "_c", t + "_invK", t + "_invB"], t + "_b", name=t + "_bn")
        y = g.op("leakyrelu", t + "_b", t + "_a", name=t + "_act", alpha=0.1)
        c = co
    return g.finish([y])


def latency(graph, blob, x):
    net = planer_amd.from_graph(graph, blob)
    net(x)
    x.ctx.synchronize()
    ts = []
    for _ in range(60):
        t0 = time.perf_counter()
        net(x)
        x.ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    plan = net.compile(x)
    return float(np.median(ts)) * 1e6, plan.algos[0]["plan"] if plan.algos else ""


def main():
    ctx = planer_amd.hip.context()
    cases = [("relu", 4, 4, 1, 8), ("relu", 128, 128, 1, 52),
             ("conv", 256, 128, 1, 52), ("conv", 512, 256, 1, 26), ("conv", 1024, 512, 1, 13),
             ("conv", 128, 128, 3, 52), ("conv", 256, 256, 3, 26), ("conv", 512, 512, 3, 13)]
    for kind, cin, cout, k, h in cases:
        x = planer_amd.asarray(np.random.default_rng(1).standard_normal((1, cin, h, h)).astype(np.float32), ctx=ctx)
        t = {}
        for n in (32, 96):
            g, b = chain(kind, n, cin, cout, k, h)
            t[n], plan = latency(g, b, x)
        print("%-5s %4d<->%-4d k%d %2dx%-2d  32 layers %7.1f us  96 layers %7.1f us  per layer %6.2f us  [%s]"
              % (kind, cin, cout, k, h, h, t[32], t[96], (t[96] - t[32]) / 64.0, plan))


if __name__ == "__main__":
    main()
