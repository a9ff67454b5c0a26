#!/usr/bin/env python
"""Experiment: ResNet-18 batch 32 as S independent sub-batch graphs on S streams (contexts) of one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import planer_amd
from planer_amd.hip import Context
from planer_amd.irgen import resnet18

g, blob = resnet18.build()
total = 32
for S in (2, 4):
    ctxs = [Context(0) for _ in range(S)]
    nets, plans, xs = [], [], []
    for c in ctxs:
        net = planer_amd.Net(c); net.load_json(g["input"], g["inits"], g["layers"], g["flow"]); net.load_weights(blob)
        x = planer_amd.asarray(resnet18.make_input(total // S), ctx=c)
        plans.append(net.compile(x)); nets.append(net); xs.append(x)
    def step():
        for p in plans: p.launch()
    for _ in range(10): step()
    for c in ctxs: c.synchronize()
    t0 = time.perf_counter()
    K = 50
    for _ in range(K): step()
    for c in ctxs: c.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("streams %d x batch %d: %.3f ms per 32 images -> %.0f img/s   inner streams %s" % (
        S, total // S, dt * 1e3, total / dt, [p.streams for p in plans]), flush=True)
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(K): step()
        for c in ctxs: c.synchronize()
        print("   repeat: %.3f ms" % ((time.perf_counter() - t0) / K * 1e3), flush=True)
    del plans, nets, xs
