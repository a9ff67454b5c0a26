"""GPU: how much of the pipelined rate is decided by WHICH side stream (hence hardware queue) each replica runs on?  Builds the
ResNet-18 throughput plan (pipe7 by default, STREAMS=) without the built-in probe, creates 24 side streams and measures the plan
under (a) the four rotations Net._probe_streams tries, (b) 40 random assignments, (c) a coordinate descent from the best of those
(one replica at a time to a stream of another residue mod 4).  Prints per-pass ms of short runs and the rate of a 150-step run
for the best few."""
import os, sys, time, random
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["PLANER_HIP_STREAM_PROBE"] = "0"
import planer_amd
from planer_amd import hip
from planer_amd.irgen import resnet18
ctx = hip.context()
g, blob = resnet18.build()
xs = [planer_amd.asarray(np.random.default_rng(1 + i).standard_normal((32, 3, 224, 224)).astype(np.float32), ctx=ctx) for i in range(2)]
net = planer_amd.from_graph(g, blob); net.streams = os.environ.get("STREAMS", "pipe7")
plan = net.compile(xs[0], mode="throughput")
R = len(plan.replicas)
NS = 24
hip.side_context(ctx.device, NS)

def run(steps):
    for i in range(steps):
        plan.feed([xs[i & 1]]); plan.launch(join=False)
    plan.join(); ctx.synchronize()

def measure(assign, rounds=3, reps=2):
    """assign[r-1] = creation index of the stream replica r (r >= 1) runs on"""
    hip.set_side_stream_perm(ctx.device, list(assign))
    run(R)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); run(rounds * R)
        best = min(best, (time.perf_counter() - t0) / (rounds * R) * 1e3)
    return best

def rate(assign):
    measure(assign, 1, 1)
    best = 0
    for _ in range(3):
        t0 = time.perf_counter(); run(150)
        best = max(best, 32 * 150 / (time.perf_counter() - t0))
    return best

seen = {}
def ev(a):
    a = tuple(a)
    if a not in seen:
        seen[a] = measure(a)
    return seen[a]

print("rotations:", [(s, round(ev([(i + s) % (R - 1 + 3) for i in range(R - 1)]), 4)) for s in range(4)])
rng = random.Random(0)
for _ in range(40):
    ev(rng.sample(range(NS), R - 1))
best = min(seen, key=seen.get)
print("best of rotations + 40 random: %s queues %s %.4f ms" % (best, [b % 4 for b in best], seen[best]))
improved = True
while improved:
    improved = False
    for r in range(R - 1):
        for q in range(4):
            if q == best[r] % 4:
                continue
            cand = next((i for i in range(NS) if i % 4 == q and i not in best), None)
            if cand is None:
                continue
            a = list(best); a[r] = cand
            if ev(a) < seen[best] * 0.995:
                best, improved = tuple(a), True
print("after coordinate descent: %s queues (creation index mod 4) %s %.4f ms" % (best, [b % 4 for b in best], seen[best]))
order = sorted(seen, key=seen.get)
for a in order[:4] + order[-2:]:
    print("  %s mod4 %s: %.4f ms short, %.0f img/s over 150 steps" % (a, [b % 4 for b in a], seen[a], rate(a)))
hist = {}
for a, v in seen.items():
    key = tuple(sorted([sum(1 for b in a if b % 4 == q) for q in range(4)], reverse=True))
    hist.setdefault(key, []).append(v)
for k, v in sorted(hist.items()):
    print("  histogram of side replicas over residues %s: n=%d best %.4f median %.4f worst %.4f" % (k, len(v), min(v), sorted(v)[len(v) // 2], max(v)))
