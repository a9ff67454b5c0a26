#!/usr/bin/env python
"""Device time of one 3x3 conv (bn + residual + relu tail) at ResNet-18's layer shapes, batch 32, per algorithm:
direct (2), fused 1-D F(4,3) (8), staged F(4x4,3x3) (7), fully fused F(4x4,3x3) (9).  HIP events, best of 5 bursts of 20.

    python tools/wf4_bench.py [--batch 32] [--shapes 64x56,128x28,256x14,512x7] [--algos 2,8,7,9]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import planer_amd  # noqa: E402
from planer_amd import hip, q4  # noqa: E402
from tools.wino_chain_bench import timed  # noqa: E402

PREP = {2: q4.prepare_q4_weights, 8: q4.prepare_w1d4_q4_weights, 4: q4.prepare_winograd_q4_weights,
        7: q4.prepare_winograd4_q4_weights, 9: q4.prepare_wf4_q4_weights}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--shapes", default="64x56,128x28,256x14,512x7")
    ap.add_argument("--algos", default="2,8,7,9")
    args = ap.parse_args()
    ctx = hip.context()
    rng = np.random.default_rng(0)
    for spec in args.shapes.split(","):
        c, h = [int(v) for v in spec.split("x")]
        n = args.batch
        x = q4.to_q4(planer_amd.asarray(rng.standard_normal((n, c, h, h)).astype(np.float32)))
        res = q4.to_q4(planer_amd.asarray(rng.standard_normal((n, c, h, h)).astype(np.float32)))
        k = planer_amd.asarray((rng.standard_normal((c, c, 3, 3)) * (2.0 / (9 * c)) ** 0.5).astype(np.float32))
        sc = planer_amd.asarray(rng.uniform(0.5, 1.5, (1, c, 1, 1)).astype(np.float32))
        flops = 2.0 * n * c * c * 9 * h * h
        for lay in [int(v) for v in args.algos.split(",")]:
            u = PREP[lay](k)
            us = timed(ctx, lambda: q4.ConvQ4(x, u, None, sc, sc, res, pads=(1, 1, 1, 1), act=1, w_layout=lay))
            ext = ctx.last_conv_extents()
            exe = 2.0 * ext[0] * ext[1] * ext[2] * ext[3]
            print("%4dx%-3d w_layout %d  %7.2f us  %6.1f TFLOP/s direct-equivalent  %6.1f executed (%.2f of 157.3)  [%s]"
                  % (c, h, lay, us, flops / us / 1e6, exe / us / 1e6, exe / us / 1e6 / 157.3, ctx.last_conv_plan()))


if __name__ == "__main__":
    main()
