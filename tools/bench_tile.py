#!/usr/bin/env python
"""Tiled large-image inference on one MI355X: a 4096 x 4096 image through
`planer_amd.util.tile` around a small conv net (1 -> 16 -> 16 -> 1 channels, 3x3), windows of
512 with 10 % margin; per-window calls (the reference's contract) vs ONE batched call for all
windows, next to the numpy oracle timed on a bounded sample (one 1024 x 1024 corner).
Prints one JSON line (megapixels of input per second)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planer_amd as pa
from planer_amd import util
from planer_amd.irgen.builder import GraphBuilder

SIZE, WINDOW = int(os.environ.get("SIZE", "4096")), int(os.environ.get("WINDOW", "512"))
rng = np.random.default_rng(0)
gb = GraphBuilder(["x"])
chans = [1, 16, 16, 1]
for i in range(3):
    gb.init("K%d" % i, (rng.standard_normal((chans[i + 1], chans[i], 3, 3)) * (0.5 / chans[i]) ** 0.5).astype(np.float32))
    gb.init("B%d" % i, (rng.standard_normal(chans[i + 1]) * 0.1).astype(np.float32))
    gb.op("conv", ["x" if i == 0 else "r%d" % (i - 1), "K%d" % i, "B%d" % i], "c%d" % i, name="conv%d" % i,
          group=1, strides=[1, 1], dilations=[1, 1], pads=[1, 1, 1, 1])
    if i < 2:
        gb.op("relu", ["c%d" % i], "r%d" % i, name="relu%d" % i)
graph, blob = gb.finish(["c2"])
net = pa.from_graph(graph, blob)
img = rng.standard_normal((SIZE, SIZE)).astype(np.float32)
dimg = pa.asarray(img)


def f_one(win):                       # (h, w) -> (h, w, 1)
    return pa.Transpose(net(pa.Unsqueeze(win, [0, 1]))[0], [1, 2, 0])


def f_all(stack):                     # (n, h, w) -> (n, h, w, 1)
    return pa.Transpose(net(pa.Unsqueeze(stack, [1])), [0, 2, 3, 1])


def timed(fn, reps=3):
    fn(); pa.hip.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    pa.hip.synchronize()
    return (time.perf_counter() - t0) / reps, out

quiet = lambda *a: None
t_one, o1 = timed(lambda: util.tile(window=WINDOW, margin=0.1, progress=quiet)(f_one)(dimg))
t_all, o2 = timed(lambda: util.tile(window=WINDOW, margin=0.1, progress=quiet, batched=True)(f_all)(dimg))
# per-window (batch 1) and batched (batch = #windows) calls may pick different conv algorithms
# (direct / Winograd variants are chosen per shape by timing): equal to rounding, not bit for bit
a1, a2 = o1.get(), o2.get()
mode_diff = float(np.abs(a1 - a2).max() / np.abs(a1).max())
assert mode_diff < 1e-4, mode_diff

# CPU: the oracle's tile around the oracle's net on a 1024 x 1024 corner (bounded sample)
from oracle import planer_np as onp
onet = onp.OracleNet(); onet.load_json(graph["input"], graph["inits"], graph["layers"], graph["flow"]); onet.load_weights(blob)
corner = img[:1024, :1024]
t0 = time.perf_counter()
ref = onp.tile(lambda w: np.ascontiguousarray(onet(w[None, None].copy())[0].transpose(1, 2, 0)), corner, window=WINDOW, margin=0.1)
t_cpu = time.perf_counter() - t0
got = util.tile(window=WINDOW, margin=0.1, progress=quiet, batched=True)(f_all)(pa.asarray(corner)).get()
err = float(np.abs(got - ref).max() / np.abs(ref).max())
nwin = len(util.grid_slice(SIZE, SIZE, WINDOW, WINDOW, int(WINDOW * 0.1)))
print(json.dumps({"metric": "tiled inference, input megapixels/sec", "image": [SIZE, SIZE], "window": WINDOW, "windows": nwin,
                  "per_window_mpix_s": round(SIZE * SIZE / t_one / 1e6, 1), "batched_mpix_s": round(SIZE * SIZE / t_all / 1e6, 1),
                  "cpu_oracle_mpix_s": round(1024 * 1024 / t_cpu / 1e6, 2), "cpu_sample": "1024x1024 corner, %d threads" % len(os.sched_getaffinity(0)),
                  "max_rel_err_vs_oracle": err,
                  "per_window_vs_batched_rel_diff": mode_diff}))
