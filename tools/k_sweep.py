#!/usr/bin/env python
"""Duration vs K at exactly PER_CU (env, default 1) workgroups per CU: slope = time per BK-chunk,
intercept = fixed cost per launch.  Config names starting with q run the channel-quad kernel."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planer_amd
from planer_amd import q4
ctx = planer_amd.hip.context(); lib = planer_amd._lib.load()
names = []
for c in range(lib.pl_conv2d_num_configs()):
    b = ctypes.create_string_buffer(32); lib.pl_conv2d_config_name(c, b, 32); names.append(b.value.decode())
rng = np.random.default_rng(0)
hw = 32
for cfg in sys.argv[1:] or ["t128x128x32", "t128x64x16", "t64x64x16"]:
    bm, bn, bk = [int(v) for v in cfg[1:].split("x")]
    pts = []
    for cin in (32, 64, 128, 256, 512):
        tiles = 256 * int(os.environ.get("PER_CU", "1"))
        n = tiles * bn // (hw * hw)
        x = planer_amd.asarray(rng.standard_normal((n, cin, hw, hw)).astype(np.float32))
        k = planer_amd.asarray((rng.standard_normal((bm, cin, 3, 3)) * 0.05).astype(np.float32))
        ctx.set_conv_config(names.index(cfg), 1)
        if cfg[0] == "q":
            xq, w = q4.to_q4(x), q4.prepare_q4_weights(k)
            f = lambda: q4.ConvQ4(xq, w, strides=[1, 1], pads=[1] * 4)
        else:
            w = planer_amd.prepare_conv_weights(k)
            f = lambda: planer_amd.ConvFused(x, w, strides=[1, 1], pads=[1] * 4, w_layout=1)
        for _ in range(3): f()
        e0 = planer_amd.hip.Event().record()
        for _ in range(10): f()
        e1 = planer_amd.hip.Event().record(); us = e0.elapsed_ms(e1) * 100
        pts.append((cin * 9 // bk, us))
    per = int(os.environ.get("PER_CU", "1"))
    (c0, t0), (c1, t1) = pts[1], pts[-1]
    slope = (t1 - t0) / (c1 - c0)
    print("%-12s" % cfg, " ".join("%d ch: %.1f us" % p for p in pts), "| slope %.3f us/chunk (ideal %.3f @2.2GHz), intercept %.1f us" % (
        slope, per * (bk // 2) * (bm // 32) * (bn // 32) / 4 * 64 / 2200.0, t0 - slope * c0))
