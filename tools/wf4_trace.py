#!/usr/bin/env python
"""Probe: per-wave time stamps of two K steps of the fused F(4x4,3x3) kernel (library built with -DWF4X_TRACE=<block>; the
kernel writes them over the tail of its input).  Prints, per wave, cycles since the earliest stamp:
step start / after the transform-first / after the MFMAs / before the barrier / after the barrier, for chunks 8 and 9."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import planer_amd
from planer_amd import hip, q4
rng = np.random.default_rng(0)
n, c, h = 32, 64, 56
x = q4.to_q4(planer_amd.asarray(rng.standard_normal((n, c, h, h)).astype(np.float32)))
k = planer_amd.asarray((rng.standard_normal((c, c, 3, 3)) * 0.05).astype(np.float32))
u = q4.prepare_wf4_q4_weights(k)
for _ in range(3):
    y = q4.ConvQ4(x, u, pads=(1, 1, 1, 1), w_layout=9)
hip.context().synchronize()
raw = x.get().view(np.uint32).ravel()
dbg = raw[-256:].reshape(16, 16)[:12, :10].astype(np.int64)
t0 = dbg.min()
for w in range(12):
    print("wave %2d  " % w + "  ".join("%6d" % (v - t0) for v in dbg[w]))
