"""GPU: fill the tuning caches (PLANER_HIP_TUNE_CACHE: launch plans + conv algorithms + stream plans) for one workload shape:
the latency plan (net(x)) and the throughput plan, as tools/make_tuned_db.sh needs them for shapes bench.py does not run.
    python tools/tune_fill.py resnet18|yolov3|customnet <batch> [size]        -> one JSON line: compile seconds, misses, rates"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import planer_amd
from planer_amd.irgen import customnet, resnet18, yolov3
which, batch = sys.argv[1], int(sys.argv[2])
size = int(sys.argv[3]) if len(sys.argv) > 3 else {"yolov3": 416, "customnet": 64}.get(which, 224)
ctx = planer_amd.hip.context()
g, b = {"yolov3": yolov3, "customnet": customnet}.get(which, resnet18).build()
xs = [planer_amd.asarray(np.random.default_rng(1 + i).standard_normal((batch, 3, size, size)).astype(np.float32), ctx=ctx) for i in range(2)]
net = planer_amd.from_graph(g, b)
t0 = time.perf_counter()
y = net(xs[0]); ctx.synchronize()
lat_s = time.perf_counter() - t0
ts = []
for _ in range(20):
    t0 = time.perf_counter(); net(xs[1]); ctx.synchronize(); ts.append(time.perf_counter() - t0)
t0 = time.perf_counter()
plan = net.compile(xs[0], mode="throughput"); ctx.synchronize()
thr_s = time.perf_counter() - t0
steps = max(20, min(300, int(0.3 / max(np.median(ts), 1e-4))))
best = 0.0
for _ in range(3):
    t0 = time.perf_counter()
    for i in range(steps):
        plan.feed([xs[i & 1]]); plan.launch(join=False)
    plan.join(); ctx.synchronize()
    best = max(best, batch * steps / (time.perf_counter() - t0))
ctx.save_tune_cache(); net.save_algo_cache()
print(json.dumps({"workload": which, "batch": batch, "size": size, "latency_ms": round(float(np.median(ts)) * 1e3, 4),
                  "images_per_sec_pipelined": round(best, 1), "streams": plan.streams, "compile_s_latency": round(lat_s, 2),
                  "compile_s_throughput": round(thr_s, 2), "tune_source": net.tune_source(),
                  "algos": sorted({a["algo"].split(" ")[0] for a in plan.algos})}))
