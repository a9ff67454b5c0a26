#!/usr/bin/env python
"""Does it matter that a batch-1 forward streams every layer's filter from HBM?  Device time of one conv at YOLO-v3's
layer shapes (batch 1) with the SAME filter every launch (warm: it sits in L2 / the memory-side cache) against a rotation of
filters whose total exceeds the 256 MB memory-side cache (cold: what a forward pass through 62 M parameters sees).

    python tools/cold_weights.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import planer_amd  # noqa: E402
from planer_amd import hip, q4  # noqa: E402

CASES = [  # cin, cout, h, k, w_layout
    (256, 128, 52, 1, 2), (128, 256, 52, 3, 7),
    (512, 256, 26, 1, 2), (256, 512, 26, 3, 7),
    (1024, 512, 13, 1, 2), (512, 1024, 13, 3, 4), (512, 1024, 13, 3, 7), (512, 1024, 13, 3, 2),
]
PREP = {2: q4.prepare_q4_weights, 4: q4.prepare_winograd_q4_weights, 7: q4.prepare_winograd4_q4_weights}


def burst(ctx, fns, reps=24):
    for f in fns:
        f()
    best = None
    for _ in range(5):
        e0 = hip.Event(ctx).record()
        for i in range(reps):
            fns[i % len(fns)]()
        e1 = hip.Event(ctx).record()
        ctx.synchronize()
        t = e0.elapsed_ms(e1) / reps * 1e3
        best = t if best is None else min(best, t)
    return best


def main():
    ctx = hip.context()
    rng = np.random.default_rng(0)
    for cin, cout, h, k, lay in CASES:
        x = q4.to_q4(planer_amd.asarray(rng.standard_normal((1, cin, h, h)).astype(np.float32)))
        sc = planer_amd.asarray(rng.uniform(0.5, 1.5, (1, cout, 1, 1)).astype(np.float32))
        w0 = (rng.standard_normal((cout, cin, k, k)) * (2.0 / (k * k * cin)) ** 0.5).astype(np.float32)
        u0 = PREP[lay](planer_amd.asarray(w0))
        nsets = max(2, int(700e6 // u0.nbytes) + 1)
        us = [u0] + [PREP[lay](planer_amd.asarray(w0)) for _ in range(min(nsets, 160) - 1)]
        p = (k // 2,) * 4

        def call(u):
            return lambda: q4.ConvQ4(x, u, None, sc, sc, None, pads=p, act=2, alpha=0.1, w_layout=lay)
        warm = burst(ctx, [call(u0)])
        plan = ctx.last_conv_plan()
        cold = burst(ctx, [call(u) for u in us], reps=len(us))
        print("%4d->%4d %2dx%-2d k%d w_layout %d  filter %6.1f MB x %3d  warm %6.2f us  cold %6.2f us  [%s]"
              % (cin, cout, h, h, k, lay, u0.nbytes / 1e6, len(us), warm, cold, plan))
        del us


if __name__ == "__main__":
    main()
