#!/usr/bin/env python
"""Device time (HIP events, best of 5 bursts of 20) of the F(4x4,3x3) stages at ResNet-18's layer2-4 shapes, batch 32:
input transform, GEMMs, output transform (with bn + residual + relu), the chained kernel with and without writing y.
Run once as is and once with PLANER_HIP_WINO_LDS=0 (register transform kernels) to compare the two families.

    python tools/wino_chain_bench.py [--batch 32]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import planer_amd  # noqa: E402
from planer_amd import hip, q4  # noqa: E402


def timed(ctx, fn, bursts=5, reps=20):
    for _ in range(3):
        fn()
    best = None
    for _ in range(bursts):
        e0 = hip.Event(ctx).record()
        for _ in range(reps):
            fn()
        e1 = hip.Event(ctx).record()
        ctx.synchronize()
        t = e0.elapsed_ms(e1) / reps * 1e3
        best = t if best is None else min(best, t)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--shapes", default="128x28,256x14,512x7")
    args = ap.parse_args()
    ctx = hip.context()
    rng = np.random.default_rng(0)
    print("PLANER_HIP_WINO_LDS=%s G=%s BD=%s" % (os.environ.get("PLANER_HIP_WINO_LDS", "1"), os.environ.get("PLANER_HIP_WINO_G"),
                                                  os.environ.get("PLANER_HIP_WINO_BD")))
    for spec in args.shapes.split(","):
        c, h = [int(v) for v in spec.split("x")]
        n = args.batch
        x = q4.to_q4(planer_amd.asarray(rng.standard_normal((n, c, h, h)).astype(np.float32)))
        res = q4.to_q4(planer_amd.asarray(rng.standard_normal((n, c, h, h)).astype(np.float32)))
        k = planer_amd.asarray((rng.standard_normal((c, c, 3, 3)) * (2.0 / (9 * c)) ** 0.5).astype(np.float32))
        u = q4.prepare_winograd4_q4_weights(k)
        sc = planer_amd.asarray(rng.uniform(0.5, 1.5, (1, c, 1, 1)).astype(np.float32))
        v = q4.Wino4In(x)
        m = q4.Wino4Gemm(v, u)
        mb = m.nbytes / 1e6
        xb = x.nbytes / 1e6
        rows = [("in", lambda: q4.Wino4In(x), xb + mb),
                ("gemm", lambda: q4.Wino4Gemm(v, u), 2 * mb),
                ("out (bn,res,relu)", lambda: q4.Wino4Out(m, None, sc, sc, res, act=1), mb + 2 * xb),
                ("out (bn,relu)", lambda: q4.Wino4Out(m, None, sc, sc, None, act=1), mb + xb)]
        if q4.wino4_chain_supported((n, c, h, h), ctx):
            rows += [("chain keep y (bn,res,relu)", lambda: q4.Wino4Chain(m, None, sc, sc, res, act=1, keep_y=True), 2 * mb + 2 * xb),
                     ("chain no y (bn,relu)", lambda: q4.Wino4Chain(m, None, sc, sc, None, act=1, keep_y=False), 2 * mb)]
        rows += [("one call conv (bn,res,relu)",
                  lambda: q4.ConvQ4(x, u, None, sc, sc, res, pads=(1, 1, 1, 1), act=1, w_layout=7), 0)]
        for name, fn, mbytes in rows:
            us = timed(ctx, fn)
            print("%4dx%-3d %-28s %7.2f us  %6.1f MB  %5.2f TB/s" % (c, h, name, us, mbytes, mbytes / us if us else 0))


if __name__ == "__main__":
    main()
