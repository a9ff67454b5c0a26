#!/usr/bin/env python
"""Where a block of the fused F(4x4,3x3) kernel spends its time: s_memtime stamps of waves 0 and 4 of every block (probe
library built with -DWF4_STAMP: planer_amd/build/ab/libstamp.so, selected through PLANER_HIP_LIB) at entry, chunk 0 landed,
prologue done, K loop done, last output row issued, stores drained.  One launch per shape (after warm-up launches), ResNet-18's
layer1 / layer2 shapes at batch 32 with the bn + residual + relu tail; s_memtime ticks at the shader clock (~0.5 ns while this kernel runs).

    PLANER_HIP_LIB=$PWD/planer_amd/build/ab/libstamp.so python tools/wf4_stamp.py
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import planer_amd  # noqa: E402
from planer_amd import hip, q4  # noqa: E402

ctx = hip.context()
lib = planer_amd._lib.load()
lib.pl_debug_wf4_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rng = np.random.default_rng(0)
NAMES = ["entry -> chunk 0 landed", "-> prologue done (transform, patch 1)", "-> K loop done", "-> last output row issued", "-> stores drained"]
for c, h, res in ((64, 56, False), (64, 56, True), (128, 28, True)):
    n = 32
    x = q4.to_q4(planer_amd.asarray(rng.standard_normal((n, c, h, h)).astype(np.float32)))
    r = q4.to_q4(planer_amd.asarray(rng.standard_normal((n, c, h, h)).astype(np.float32))) if res else None
    k = planer_amd.asarray((rng.standard_normal((c, c, 3, 3)) * (2.0 / (9 * c)) ** 0.5).astype(np.float32))
    sc = planer_amd.asarray(rng.uniform(0.5, 1.5, (1, c, 1, 1)).astype(np.float32))
    u = q4.prepare_wf4_q4_weights(k)
    for _ in range(5):
        q4.ConvQ4(x, u, None, sc, sc, r, pads=(1, 1, 1, 1), act=1, w_layout=9)
    ctx.synchronize()
    e0, e1 = hip.Event(ctx).record(), None
    q4.ConvQ4(x, u, None, sc, sc, r, pads=(1, 1, 1, 1), act=1, w_layout=9)
    e1 = hip.Event(ctx).record()
    ctx.synchronize()
    plan = ctx.last_conv_plan()
    blocks = int(plan.split("blocks=")[1])
    buf = np.zeros(4096 * 2 * 8, np.uint64)
    assert lib.pl_debug_wf4_stamps(buf.ctypes.data, buf.nbytes) == 0
    st = buf.reshape(4096, 2, 8)[:blocks].astype(np.int64)
    t0 = st[:, :, 0].min()
    seg = np.diff(st[:, :, :6], axis=2)                      # (blocks, 2 waves, 5 segments)
    tick_us = float(os.environ.get("TICK_US", "0.0005"))      # (s_memtime here: ~2 GHz while the kernel runs)
    print("%dx%d res=%s  [%s]  launch by HIP events %.2f us" % (c, h, res, plan, e0.elapsed_ms(e1) * 1e3))
    print("   entry skew over blocks: first %.2f us, last %.2f us after the earliest;  last stamp %.2f us" % (
        (st[:, :, 0].min() - t0) * tick_us, (st[:, :, 0].max() - t0) * tick_us, (st[:, :, 5].max() - t0) * tick_us))
    for i, name in enumerate(NAMES):
        for w, tag in ((0, "wave 0"), (1, "wave 4")):
            v = seg[:, w, i] * tick_us
            print("   %-40s %s: mean %6.2f  min %6.2f  max %6.2f us" % (name, tag, v.mean(), v.min(), v.max()))
    sb = np.zeros(512 * 8 * 2 * 8, np.uint64)
    lib.pl_debug_wf4_step_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert lib.pl_debug_wf4_step_stamps(sb.ctypes.data, sb.nbytes) == 0
    ss = sb.reshape(512, 8, 2, 8)[:min(blocks, 512)].astype(np.int64)          # (block, wave, step 6 / 7, mark)
    base = ss[:, :, :, 0].min(axis=1, keepdims=True)                            # earliest wave's entry into the step, per block
    print("   K steps 6 and 7, ticks after the step's first wave entered (mean over blocks): entry | transform_first done | MFMAs done | transform_last done | barrier passed")
    for w in range(8):
        for j in (0, 1):
            v = (ss[:, w, j, :5] - base[:, 0, j, None]).mean(axis=0)
            print("      wave %d step %d: %s" % (w, 6 + j, "  ".join("%6.0f" % t for t in v)))
    print("      step length (barrier to barrier): %.0f ticks" % (ss[:, :, 1, 4] - ss[:, :, 0, 4]).mean())
    tot = (st[:, :, 5] - st[:, :, 0]) * tick_us
    print("   block total (entry -> drained): mean %.2f  min %.2f  max %.2f us" % (tot.mean(), tot.min(), tot.max()))
