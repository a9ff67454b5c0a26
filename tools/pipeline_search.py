"""GPU: does the per-conv algorithm that wins in ISOLATION also win inside the pipelined throughput plan?
Coordinate descent over Net._algo (conv shape signature -> w_layout): for every signature try every
candidate algorithm, rebuild the three-stream plan and measure images/s; keep what is faster."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import planer_amd
from planer_amd.irgen import resnet18

ctx = planer_amd.hip.context()
g, blob = resnet18.build()
xs = [planer_amd.asarray(np.random.default_rng(1 + i).standard_normal((32, 3, 224, 224)).astype(np.float32), ctx=ctx) for i in range(2)]


def measure(algo, steps=150):
    net = planer_amd.from_graph(g, blob)
    net.streams = os.environ.get("STREAMS", "pipe3")
    net._algo_loaded = True
    net._algo = dict(algo)
    plan = net.compile(xs[0], mode="throughput")
    best = 0.0
    for rep in range(3):
        for i in range(10):
            plan.feed([xs[i & 1]]); plan.launch(join=False)
        plan.join(); ctx.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            plan.feed([xs[i & 1]]); plan.launch(join=False)
        plan.join(); ctx.synchronize()
        best = max(best, 32 * steps / (time.perf_counter() - t0))
    return best, dict(net._algo)


base, algo = measure({})
print("isolated picks: %.0f img/s" % base, {k[1][1:3] + (k[3][3],): v for k, v in algo.items()}, flush=True)
cur, cur_rate = dict(algo), base
for sig in sorted(algo, key=repr):
    for lay in (2, 5, 8, 4, 7):
        if lay == cur[sig]:
            continue
        trial = dict(cur); trial[sig] = lay
        try:
            rate, _ = measure(trial)
        except Exception as e:
            print("  sig", sig[1], "lay", lay, "failed:", str(e)[:60]); continue
        print("  C%d %dx%d res=%s: w_layout %d -> %d: %.0f img/s (%+.1f%%)" % (sig[1][1], sig[1][2], sig[1][3], sig[3][3], cur[sig], lay, rate, 100 * (rate / cur_rate - 1)), flush=True)
        if rate > cur_rate * 1.004:
            cur, cur_rate = trial, rate
print("after search: %.0f img/s (%+.1f%% over the isolated picks)" % (cur_rate, 100 * (cur_rate / base - 1)), {k[1][1:3] + (k[3][3],): v for k, v in cur.items()})
