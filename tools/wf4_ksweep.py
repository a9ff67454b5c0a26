#!/usr/bin/env python
"""Device time of the fused F(4x4,3x3) kernel against the number of K chunks (Cin / 4) at a fixed output (batch 32, 64 output
channels, 56x56): time = fixed + chunks x slope separates prologue / output transform from the K loop.

    python tools/wf4_ksweep.py [--hw 56] [--cout 64] [--algo 9]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import planer_amd  # noqa: E402
from planer_amd import hip, q4  # noqa: E402
from tools.wf4_bench import PREP  # noqa: E402
from tools.wino_chain_bench import timed  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--hw", type=int, default=56)
ap.add_argument("--cout", type=int, default=64)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--algo", type=int, default=9)
ap.add_argument("--tail", action="store_true", help="scale + shift + residual + ReLU (the tail a ResNet block carries)")
args = ap.parse_args()
ctx = hip.context()
rng = np.random.default_rng(0)
pts = []
for cin in (4, 8, 16, 32, 64, 128, 256):
    x = q4.to_q4(planer_amd.asarray(rng.standard_normal((args.batch, cin, args.hw, args.hw)).astype(np.float32)))
    k = planer_amd.asarray((rng.standard_normal((args.cout, cin, 3, 3)) * 0.05).astype(np.float32))
    u = PREP[args.algo](k)
    if args.tail:
        sc = planer_amd.asarray(rng.standard_normal(args.cout).astype(np.float32))
        res = q4.to_q4(planer_amd.asarray(rng.standard_normal((args.batch, args.cout, args.hw, args.hw)).astype(np.float32)))
        us = timed(ctx, lambda: q4.ConvQ4(x, u, None, sc, sc, res, pads=(1, 1, 1, 1), act=1, w_layout=args.algo))
    else:
        us = timed(ctx, lambda: q4.ConvQ4(x, u, pads=(1, 1, 1, 1), w_layout=args.algo))
    pts.append((cin // 4, us))
    print("Cin %4d  chunks %3d  %8.2f us  [%s]" % (cin, cin // 4, us, ctx.last_conv_plan()))
(c0, t0), (c1, t1) = pts[-3], pts[-1]
slope = (t1 - t0) / (c1 - c0)
print("slope %.3f us per chunk, fixed %.2f us (from the last three points)" % (slope, t1 - slope * c1))
