#!/usr/bin/env python
"""Conv throughput, NCHW tap-major kernel vs channel-quad (Q4) kernel:
 (a) perfectly balanced grids at k workgroups per CU (no tile quantisation), forced configs;
 (b) the ResNet-18 batch-32 conv shapes with each path's own autotuned plan."""
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


def timeit(f, reps=10):
    for _ in range(3): f()
    e0 = planer_amd.hip.Event().record()
    for _ in range(reps): f()
    e1 = planer_amd.hip.Event().record()
    return e0.elapsed_ms(e1) / reps


if "--grid" in sys.argv or len(sys.argv) == 1:
    cin, hw = 128, 32
    for cfg in ["128x128x16", "128x64x16", "64x64x16", "128x128x32", "64x128x32"]:
        bm, bn = [int(v) for v in cfg.split("x")[:2]]
        for kind in "tq":
            row = []
            for per_cu in (1, 2, 3, 4, 8):
                tiles = 256 * per_cu; cout = bm; n = tiles * bn // (hw * hw)
                if n * hw * hw != tiles * bn or n < 1:
                    row.append("   n/a   "); continue
                x = planer_amd.asarray(rng.standard_normal((n, cin, hw, hw)).astype(np.float32))
                k = planer_amd.asarray((rng.standard_normal((cout, cin, 3, 3)) * 0.05).astype(np.float32))
                ctx.set_conv_config(names.index(kind + cfg), 1)
                if kind == "t":
                    w = planer_amd.prepare_conv_weights(k)
                    f = lambda: planer_amd.ConvFused(x, w, strides=[1, 1], pads=[1] * 4, w_layout=1)
                else:
                    xq, w = q4.to_q4(x), q4.prepare_q4_weights(k)
                    f = lambda: q4.ConvQ4(xq, w, strides=[1, 1], pads=[1] * 4)
                ms = timeit(f)
                row.append("%d/CU %5.1f TF" % (per_cu, 2.0 * n * cout * hw * hw * cin * 9 / ms / 1e9))
            print("%-12s" % (kind + cfg), " | ".join(row), flush=True)
    ctx.set_conv_config(-1, 0)

if "--resnet" in sys.argv or len(sys.argv) == 1:
    N = int(os.environ.get("BATCH", "32"))
    shapes = [("stem 7x7/2", (N, 3, 224, 224), (64, 3, 7, 7), 2, 3),
              ("l1 3x3", (N, 64, 56, 56), (64, 64, 3, 3), 1, 1),
              ("l2 3x3/2", (N, 64, 56, 56), (128, 64, 3, 3), 2, 1),
              ("l2 3x3", (N, 128, 28, 28), (128, 128, 3, 3), 1, 1),
              ("l2 1x1/2", (N, 64, 56, 56), (128, 64, 1, 1), 2, 0),
              ("l3 3x3/2", (N, 128, 28, 28), (256, 128, 3, 3), 2, 1),
              ("l3 3x3", (N, 256, 14, 14), (256, 256, 3, 3), 1, 1),
              ("l4 3x3/2", (N, 256, 14, 14), (512, 256, 3, 3), 2, 1),
              ("l4 3x3", (N, 512, 7, 7), (512, 512, 3, 3), 1, 1)]
    for label, xs, ks, s, p in shapes:
        x = planer_amd.asarray(rng.standard_normal(xs).astype(np.float32))
        k = planer_amd.asarray((rng.standard_normal(ks) * 0.05).astype(np.float32))
        kw = dict(strides=[s, s], pads=[p] * 4)
        ho = (xs[2] + 2 * p - ks[2]) // s + 1
        # the tail the ResNet plan really fuses: folded batchnorm (+ residual on stride-1 3x3) + relu
        sc = planer_amd.asarray(rng.uniform(0.5, 1.5, (1, ks[0], 1, 1)).astype(np.float32))
        sh = planer_amd.asarray(rng.standard_normal((1, ks[0], 1, 1)).astype(np.float32))
        res = planer_amd.asarray(rng.standard_normal((xs[0], ks[0], ho, ho)).astype(np.float32)) if s == 1 and ks[2] == 3 else None
        resq = q4.to_q4(res) if res is not None else None
        if ks[1] % 16 == 0:
            w = planer_amd.prepare_conv_weights(k)
            t_old = timeit(lambda: planer_amd.ConvFused(x, w, None, sc, sh, res, act=1, w_layout=1, **kw))
        else:
            t_old = timeit(lambda: planer_amd.ConvFused(x, k, None, sc, sh, res, act=1, **kw))
        xq, wq = q4.to_q4(x), q4.prepare_q4_weights(k)
        t_q4 = timeit(lambda: q4.ConvQ4(xq, wq, None, sc, sh, resq, act=1, **kw))
        fl = 2.0 * xs[0] * ks[0] * ho * ho * ks[1] * ks[2] * ks[3]
        extra = ""
        if q4.rowpack_eligible(ks, **kw):
            wr = q4.prepare_rowpack_weights(k)
            t_r = timeit(lambda: q4.ConvQ4(x, wr, None, sc, sh, resq, act=1, w_layout=6, **kw))
            extra = " | row-packed (incl. input re-layout) %7.1f us %6.1f TF" % (t_r * 1e3, fl / t_r / 1e9)
        if q4.w1d_q4_eligible(ks, **kw):
            u8 = q4.prepare_w1d4_q4_weights(k)
            t_8 = timeit(lambda: q4.ConvQ4(xq, u8, None, sc, sh, resq, act=1, w_layout=8, **kw))
            extra = " | 1-D F(4,3) %7.1f us %6.1f TF" % (t_8 * 1e3, fl / t_8 / 1e9)
            if xs[0] <= 32:
                u2 = q4.prepare_winograd_q4_weights(k)
                t_2 = timeit(lambda: q4.ConvQ4(xq, u2, None, sc, sh, resq, act=1, w_layout=4, **kw))
                extra += " | winograd-2d %7.1f us" % (t_2 * 1e3)
            if ks[0] % 4 == 0:
                u4 = q4.prepare_winograd4_q4_weights(k)
                t_4 = timeit(lambda: q4.ConvQ4(xq, u4, None, sc, sh, resq, act=1, w_layout=7, **kw))
                extra += " | F(4,3) %7.1f us" % (t_4 * 1e3)
        print("%-11s nchw %7.1f us %6.1f TF | q4 %7.1f us %6.1f TF | x%.2f%s" %
              (label, t_old * 1e3, fl / t_old / 1e9, t_q4 * 1e3, fl / t_q4 / 1e9, t_old / t_q4, extra), flush=True)
