#!/usr/bin/env python
"""BASELINE config 2 (Conv2d 3->64 k3 s1 p1 on (8,3,224,224)) three ways, to say what one replay of its captured plan is made
of: (a) the conv alone, eager launches back to back (HIP events); (b) the plan as `bench.py`'s secondary workload replays it
(K graph launches behind a blocker); (c) under `rocprofv3 --kernel-trace` (run this script through the tool): every kernel of
a replay with its duration and the gap in front of it.

    python tools/config2_trace.py                       # (a) + (b)
    rocprofv3 --kernel-trace -d out -o c2 --output-format csv -- python tools/config2_trace.py trace
    python tools/config2_trace.py digest out/c2_kernel_trace.csv
"""
import csv
import os
import re
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def build():
    import planer_amd
    from planer_amd.irgen.builder import GraphBuilder
    rng = np.random.default_rng(0)
    gb = GraphBuilder(["x"])
    gb.init("K", (rng.standard_normal((64, 3, 3, 3)) * 0.1).astype(np.float32))
    gb.init("B", rng.standard_normal(64).astype(np.float32))
    gb.op("conv", ["x", "K", "B"], "y", name="conv", group=1, strides=[1, 1], dilations=[1, 1], pads=[1, 1, 1, 1])
    g, blob = gb.finish(["y"])
    ctx = planer_amd.hip.context()
    net = planer_amd.Net(ctx)
    net.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    net.load_weights(blob)
    net.streams = "1x1"
    x = planer_amd.asarray(np.random.default_rng(7).standard_normal((8, 3, 224, 224)).astype(np.float32), ctx=ctx)
    return planer_amd, ctx, net, x


def digest(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 2:]                      # the replays at the end (the first half holds compile / tuning launches)
    stat = {}
    prev = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
        d = stat.setdefault(name, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
        if prev is not None:
            d[2] += max(0.0, (s - prev) / 1e3)
        prev = e
    for name, (n, dur, gap) in sorted(stat.items(), key=lambda kv: -kv[1][1]):
        print("%-70s x %5d  %8.2f us each  gap in front %6.2f us" % (name, n, dur / n, gap / n))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "digest":
        digest(sys.argv[2])
        sys.exit(0)
    pa, ctx, net, x = build()
    plan = net.compile(x)
    y = net(x)
    print("plan steps:", [k for _, k in getattr(plan, "fused_steps_list", [])] or plan.fused_steps, "algos:", [(a["layer"], a["algo"], a["plan"]) for a in plan.algos])
    ev = [pa.hip.Event(ctx) for _ in range(2)]

    def burst(fn, k):
        for _ in range(5):
            fn()
        ctx.synchronize()
        best = None
        for _ in range(7):
            ev[0].record()
            for _ in range(k):
                fn()
            ev[1].record()
            t = ev[0].elapsed_ms(ev[1]) / k * 1e3
            best = t if best is None else min(best, t)
        return best
    if len(sys.argv) > 1 and sys.argv[1] == "trace":
        plan.max_in_flight = 0
        for _ in range(200):
            plan.launch()
        ctx.synchronize()
        sys.exit(0)
    K = dict(zip(net.inits, net.weights))
    eager = burst(lambda: pa.Conv2d(x, K["K"], K["B"], pads=(1, 1, 1, 1)), 40)
    print("(a) layer.Conv2d eager, back to back: %.2f us per launch  [%s]" % (eager, ctx.last_conv_plan()))
    plan.max_in_flight = 0
    print("(b) plan.launch() back to back (one captured graph per replay): %.2f us per replay" % burst(plan.launch, 40))
    blocker = pa.hip.empty((256 << 20,), np.float32, ctx)

    def behind_blocker():
        pass
    best = None
    for _ in range(5):
        for _ in range(16):
            pa._lib.call("pl_memset", ctx.handle, blocker.ptr, 0, blocker.nbytes)
        ev[0].record()
        for _ in range(40):
            plan.launch()
        ev[1].record()
        t = ev[0].elapsed_ms(ev[1]) / 40 * 1e3
        best = t if best is None else min(best, t)
    print("(b') the same behind 16 memsets of 1 GiB (bench.py's secondary workload form): %.2f us per replay" % best)
