#!/usr/bin/env python
"""Achieved HBM GB/s of the stand-alone HBM-bound layer kernels (BatchNorm, ReLU, LeakyReLU, Sigmoid, Add, Maxpool,
UpSample, Concatenate, GlobalAveragePool: layer.py:44-51, 61-64, 71-72, 77-82, 90-95, 125-127) at the shapes BASELINE's
configs run them at when nothing is fused: config 1 (CustomNet, (1,3,64,64)), ResNet-18 at batch 32, YOLO-v3 at batch 1.
In a compiled plan BatchNorm / ReLU / Add are epilogues of the conv kernels; these are the kernels `Net.forward` (one kernel per
layer, the reference's own execution model) and any graph the fuser cannot touch run.

    python tools/pointwise_gbs.py [--manifest m.json]        # HIP-event table (best of 5 bursts of 20) as markdown
    tools/pointwise_prof.sh r04                              # the same cases under rocprofv3 --kernel-trace -> profiles/

Algorithmic bytes = every input read once + the output written once (fp32).  With --manifest the launches of each case are
recorded (case, launches); three small memsets in a row separate the cases, so tools/pointwise_digest.py can cut a kernel
trace into one segment per case.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import planer_amd as pa  # noqa: E402
from planer_amd import hip  # noqa: E402

PEAK = 8000.0


def cases(rng):
    def t(*shape):
        return pa.asarray(rng.standard_normal(shape).astype(np.float32))
    k2 = np.array([1, 1, 2, 2], np.float32)
    out = []

    def add(net, name, fn, nbytes):
        out.append((net, name, fn, float(nbytes)))
    # ---- config 1: CustomNet on (1,3,64,64): relu -> maxpool k2s2 -> upsample x2 -> concat -> sigmoid
    a = t(1, 64, 64, 64); b = t(1, 64, 32, 32); c = t(1, 128, 64, 64)
    add("config1", "relu (1,64,64,64)", lambda: pa.ReLU(a), 2 * a.nbytes)
    add("config1", "maxpool k2 s2 (1,64,64,64)", lambda: pa.Maxpool(a, (2, 2), (0, 0, 0, 0), (2, 2)), a.nbytes + b.nbytes)
    add("config1", "upsample x2 (1,64,32,32)", lambda: pa.UpSample(b, k2), b.nbytes + a.nbytes)
    add("config1", "concat axis 1 2x(1,64,64,64)", lambda: pa.Concatenate(a, a, axis=1), 4 * a.nbytes)
    add("config1", "sigmoid (1,128,64,64)", lambda: pa.Sigmoid(c), 2 * c.nbytes)
    # ---- ResNet-18, batch 32 (bn 9.9 MB r+w per image, relu 9.2, add 6+3, maxpool 3.2+0.8, gap 0.1)
    s = t(32, 64, 112, 112); sc = t(1, 64, 1, 1)
    l1 = t(32, 64, 56, 56); l1b = t(32, 64, 56, 56)
    l4 = t(32, 512, 7, 7)
    add("resnet18 b32", "batchnorm (32,64,112,112)", lambda: pa.BatchNorm(s, sc, sc), 2 * s.nbytes)
    add("resnet18 b32", "relu (32,64,112,112)", lambda: pa.ReLU(s), 2 * s.nbytes)
    add("resnet18 b32", "maxpool k3 s2 p1 (32,64,112,112)", lambda: pa.Maxpool(s, (3, 3), (1, 1, 1, 1), (2, 2)), s.nbytes + l1.nbytes)
    add("resnet18 b32", "batchnorm (32,64,56,56)", lambda: pa.BatchNorm(l1, sc, sc), 2 * l1.nbytes)
    add("resnet18 b32", "add (32,64,56,56)", lambda: pa.Add(l1, l1b), 3 * l1.nbytes)
    add("resnet18 b32", "relu (32,64,56,56)", lambda: pa.ReLU(l1), 2 * l1.nbytes)
    add("resnet18 b32", "add (32,512,7,7)", lambda: pa.Add(l4, l4), 3 * l4.nbytes)
    add("resnet18 b32", "gap (32,512,7,7)", lambda: pa.GlobalAveragePool(l4), l4.nbytes + 32 * 512 * 4)
    # ---- YOLO-v3 @416, batch 1
    y0 = t(1, 32, 416, 416); s32 = t(1, 32, 1, 1)
    y2 = t(1, 128, 104, 104); s128 = t(1, 128, 1, 1)
    u13 = t(1, 256, 13, 13); u26 = t(1, 256, 26, 26); c26 = t(1, 512, 26, 26)
    u52 = t(1, 128, 52, 52); c52 = t(1, 256, 52, 52); v26 = t(1, 128, 26, 26)
    add("yolov3 b1", "batchnorm (1,32,416,416)", lambda: pa.BatchNorm(y0, s32, s32), 2 * y0.nbytes)
    add("yolov3 b1", "leakyrelu (1,32,416,416)", lambda: pa.LeakyReLU(y0, 0.1), 2 * y0.nbytes)
    add("yolov3 b1", "batchnorm (1,128,104,104)", lambda: pa.BatchNorm(y2, s128, s128), 2 * y2.nbytes)
    add("yolov3 b1", "leakyrelu (1,128,104,104)", lambda: pa.LeakyReLU(y2, 0.1), 2 * y2.nbytes)
    add("yolov3 b1", "add (1,128,104,104)", lambda: pa.Add(y2, y2), 3 * y2.nbytes)
    add("yolov3 b1", "upsample x2 (1,256,13,13)", lambda: pa.UpSample(u13, k2), 5 * u13.nbytes)
    add("yolov3 b1", "concat (1,256,26,26)+(1,512,26,26)", lambda: pa.Concatenate(u26, c26, axis=1), 2 * (u26.nbytes + c26.nbytes))
    add("yolov3 b1", "upsample x2 (1,128,26,26)", lambda: pa.UpSample(v26, k2), 5 * v26.nbytes)
    add("yolov3 b1", "concat (1,128,52,52)+(1,256,52,52)", lambda: pa.Concatenate(u52, c52, axis=1), 2 * (u52.nbytes + c52.nbytes))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--manifest")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    ctx = hip.context()
    rng = np.random.default_rng(0)
    rows, manifest = [], []
    mark = hip.zeros((64,), np.float32, ctx)
    for net, name, fn, nbytes in cases(rng):
        for _ in range(3):                      # three fill-buffer dispatches in a row separate the cases in a kernel trace
            pa._lib.call("pl_memset", ctx.handle, mark.ptr, 0, mark.nbytes)
        for _ in range(3):
            fn()
        ctx.synchronize()
        best = None
        for _ in range(5):
            e0 = hip.Event(ctx).record()
            for _ in range(args.reps):
                fn()
            e1 = hip.Event(ctx).record()
            ctx.synchronize()
            us = e0.elapsed_ms(e1) / args.reps * 1e3
            best = us if best is None else min(best, us)
        rows.append((net, name, nbytes, best))
        manifest.append({"net": net, "case": name, "bytes": nbytes, "launches": 3 + 5 * args.reps})
    if args.manifest:
        with open(args.manifest, "w") as f:
            json.dump(manifest, f)
    print("| workload | layer (shape) | algorithmic MB | us (HIP events) | GB/s | of 8 TB/s |")
    print("|---|---|---|---|---|---|")
    for net, name, nbytes, us in rows:
        gbs = nbytes / us / 1e3
        print("| %s | %s | %.2f | %.2f | %.0f | %.3f |" % (net, name, nbytes / 1e6, us, gbs, gbs / PEAK))


if __name__ == "__main__":
    main()
