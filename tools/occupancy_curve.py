#!/usr/bin/env python
"""Efficiency vs. workgroups-per-CU for the conv tile configs on perfectly
balanced grids (tiles = k*256), i.e. with no tile-quantization loss."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planer_amd
ctx = planer_amd.hip.context(); lib = planer_amd._lib.load()
names = []
for c in range(lib.pl_conv2d_num_configs()):
    b = ctypes.create_string_buffer(32); lib.pl_conv2d_config_name(c, b, 32); names.append(b.value.decode())
rng = np.random.default_rng(0)
cin, hw = 128, 32
for cfg in sys.argv[1:] or ["t64x64x16", "t64x64x32", "t128x64x16", "t128x64x32", "t128x128x16", "t128x128x32", "t64x128x16"]:
    bm, bn = [int(v) for v in cfg[1:].split("x")[:2]]
    row = []
    for per_cu in (1, 2, 3, 4, 6, 8):
        tiles = 256 * per_cu
        cout = bm                      # one m-tile
        n = tiles * bn // (hw * hw)    # cols = n*hw*hw = tiles*bn
        if n * hw * hw != tiles * bn or n < 1:
            row.append("  n/a "); continue
        x = planer_amd.asarray(rng.standard_normal((n, cin, hw, hw)).astype(np.float32))
        w = planer_amd.prepare_conv_weights(planer_amd.asarray((rng.standard_normal((cout, cin, 3, 3)) * 0.05).astype(np.float32)))
        ctx.set_conv_config(names.index(cfg), 1)
        f = lambda: planer_amd.ConvFused(x, w, strides=[1, 1], pads=[1] * 4, w_layout=1)
        for _ in range(3): f()
        e0 = planer_amd.hip.Event().record()
        for _ in range(10): f()
        e1 = planer_amd.hip.Event().record(); ms = e0.elapsed_ms(e1) / 10
        row.append("%d/CU %5.1f TF (%.0f us)" % (per_cu, 2.0 * n * cout * hw * hw * cin * 9 / ms / 1e9, ms * 1e3))
    print("%-12s" % cfg, " | ".join(row), flush=True)
