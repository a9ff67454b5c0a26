"""GPU: one-image-at-a-time latency of a workload (device-resident input, captured plan, `net(x_dev)` then sync),
next to the pipelined rate bench.py reports -- BASELINE config 5 (YOLO-v3 @416, batch 1)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import planer_amd
from planer_amd.irgen import yolov3, resnet18

ctx = planer_amd.hip.context()
which = sys.argv[1] if len(sys.argv) > 1 else "yolov3"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g, b = (yolov3 if which == "yolov3" else resnet18).build()
size = 416 if which == "yolov3" else 224
x = planer_amd.asarray(np.random.default_rng(1).standard_normal((batch, 3, size, size)).astype(np.float32), ctx=ctx)
net = planer_amd.from_graph(g, b)
net(x); ctx.synchronize()
ctx.save_tune_cache(); net.save_algo_cache()
ts = []
for _ in range(50):
    t0 = time.perf_counter()
    y = net(x)
    ctx.synchronize()
    ts.append(time.perf_counter() - t0)
plan = net.compile(x)
kern = {}
for a in plan.algos:
    kern[a["plan"].split()[0].split("[")[0]] = kern.get(a["plan"].split()[0].split("[")[0], 0) + 1
print(json.dumps({"workload": which, "batch": batch, "latency_ms_median": round(float(np.median(ts)) * 1e3, 4),
                  "latency_ms_best": round(min(ts) * 1e3, 4), "images_per_sec_one_at_a_time": round(batch / float(np.median(ts)), 1),
                  "streams": plan.streams, "conv_plans": kern}))
