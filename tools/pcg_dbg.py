"""K sweep of one persistent-kernel configuration at a fixed grid: per-chunk slope and fixed cost."""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import planer_amd as pa
from planer_amd import q4, hip
ctx = hip.context()
lib = pa._lib.load()
names = []
for c in range(lib.pl_conv2d_num_configs()):
    buf = ctypes.create_string_buffer(32); lib.pl_conv2d_config_name(c, buf, 32); names.append(buf.value.decode())
rng = np.random.default_rng(3)
cfg = os.environ.get("CFG", "p128x128x16")
pts = []
for cin in (32, 64, 128, 256):
    x = rng.standard_normal((32, cin, 56, 56)).astype(np.float32)
    k = (rng.standard_normal((128, cin, 3, 3)) * 0.05).astype(np.float32)
    xq = q4.to_q4(pa.asarray(x)); kq = q4.prepare_q4_weights(pa.asarray(k))
    ctx.set_conv_config(names.index(cfg), 1)
    run = lambda: q4.ConvQ4(xq, kq, None, strides=[2, 2], pads=[1, 1, 1, 1])
    for _ in range(3): run()
    best = 1e9
    for _ in range(3):
        e0 = hip.Event(ctx).record()
        for _ in range(10): run()
        e1 = hip.Event(ctx).record()
        best = min(best, e0.elapsed_ms(e1) / 10)
    chunks = 9 * cin // 4 // 4
    pts.append((chunks, best * 1e3))
    print("Cin %3d: %3d chunks %.1f us [%s]" % (cin, chunks, best * 1e3, ctx.last_conv_plan()))
(c0, t0), (c1, t1) = pts[0], pts[-1]
slope = (t1 - t0) / (c1 - c0)
print("%s: slope %.3f us/chunk (MFMA-bound 32*64 cycles = %.3f us at 2.2 GHz), fixed %.1f us"
      % (cfg, slope, 2048 / 2200.0, t0 - slope * c0))
