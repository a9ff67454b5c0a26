#!/usr/bin/env python
"""ResNet-18's stem at batch 32 (3 -> 64, 7x7 / s2 / p3 on 224x224, bn + relu) followed by maxpool(3x3 / s2 / p1): device time of
the one-kernel form (ConvPoolQ4, conv_stem_pool_kernel) against conv kernel + pool kernel, packed image prepared once (as a
plan's feed does).  `--reps N --only fused|pair` runs one form N times (for rocprofv3 passes: tools/stem_pool_pmc.sh)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import planer_amd as pa  # noqa: E402
from planer_amd import hip, q4  # noqa: E402
from tools.wino_chain_bench import timed  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--reps", type=int, default=0)
ap.add_argument("--only", default="")
args = ap.parse_args()
ctx = hip.context()
rng = np.random.default_rng(0)
x = pa.asarray(rng.standard_normal((args.batch, 3, 224, 224)).astype(np.float32))
K = q4.prepare_rowpack_weights(pa.asarray((rng.standard_normal((64, 3, 7, 7)) * 0.1).astype(np.float32)))
sc = pa.asarray(rng.uniform(0.5, 1.5, (1, 64, 1, 1)).astype(np.float32))
para = dict(strides=[2, 2], pads=[3, 3, 3, 3], dilations=[1, 1], group=1)
q4.pack_rows(x, geom=(7, 2, 3, 3))                      # x.packed: both forms read the packed image
fused = lambda: q4.ConvPoolQ4(x, K, None, sc, sc, act=1, **para)
conv = lambda: q4.ConvQ4(x, K, None, sc, sc, None, act=1, w_layout=6, **para)
y = conv()
pool = lambda: q4.MaxpoolQ4(y, (3, 3), (1, 1, 1, 1), (2, 2))
if args.reps:
    for _ in range(args.reps):
        if args.only != "pair":
            fused()
        if args.only != "fused":
            conv(); pool()
    ctx.synchronize()
else:
    tf, tc, tp = timed(ctx, fused), timed(ctx, conv), timed(ctx, pool)
    print("batch %d: stem + maxpool in one kernel %.1f us [%s]; conv %.1f us + pool %.1f us = %.1f us"
          % (args.batch, tf, "fused", tc, tp, tc + tp))
