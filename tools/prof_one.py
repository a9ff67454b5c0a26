#!/usr/bin/env python
"""Run ONE conv shape with ONE tile config a few times (for rocprofv3).
   python tools/prof_one.py N Cin H Cout k stride pad cfgname split reps"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planer_amd

n, cin, h, cout, k, st, pd = [int(v) for v in sys.argv[1:8]]
cfgname, split, reps = sys.argv[8], int(sys.argv[9]), int(sys.argv[10])
ctx = planer_amd.hip.context()
lib = planer_amd._lib.load()
names = []
for c in range(lib.pl_conv2d_num_configs()):
    buf = ctypes.create_string_buffer(32)
    lib.pl_conv2d_config_name(c, buf, 32)
    names.append(buf.value.decode())
cfg = names.index(cfgname)
tap = cfgname.startswith("t")
rng = np.random.default_rng(0)
x = planer_amd.asarray(rng.standard_normal((n, cin, h, h)).astype(np.float32))
w = planer_amd.asarray((rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32))
if tap:
    w = planer_amd.prepare_conv_weights(w)
sc = planer_amd.asarray(rng.uniform(0.5, 1.5, (1, cout, 1, 1)).astype(np.float32))
ctx.set_conv_config(cfg, split)
for _ in range(reps):
    y = planer_amd.ConvFused(x, w, None, sc, sc, None, strides=[st, st], pads=[pd] * 4, act=1, w_layout=int(tap))
ctx.synchronize()
