#!/usr/bin/env python
"""Sweep the implicit-GEMM conv kernel's tile configs / split-K over the conv
shapes of a model and print achieved TFLOP/s (HIP events, on the GPU box).

    python tools/tune_conv.py [--model resnet18|yolov3] [--batch 32] [--reps 20]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import planer_amd  # noqa: E402
from planer_amd import _lib  # noqa: E402
from planer_amd.irgen import resnet18, yolov3  # noqa: E402


def conv_shapes(g, x_shape):
    """Distinct (N,Cin,H,W,Cout,kh,kw,stride,pad) with multiplicity, by shape inference."""
    kinds = {n: (k, p) for n, k, p in g["layers"]}
    shapes = {k: tuple(s) for k, s, _ in g["inits"]}
    shapes[g["input"][0]] = x_shape
    out = {}
    for src, names, dst in g["flow"]:
        kind, para = kinds[names[0]]
        srcs = src if isinstance(src, list) else [src]
        s0 = shapes.get(srcs[0])
        if kind == "conv":
            n, cin, h, w = s0
            cout, _, kh, kw = shapes[srcs[1]]
            st, pd = para["strides"], para["pads"]
            ho = (h + 2 * pd[0] - kh) // st[0] + 1
            wo = (w + 2 * pd[1] - kw) // st[1] + 1
            shapes[dst] = (n, cout, ho, wo)
            key = (n, cin, h, w, cout, kh, kw, st[0], pd[0])
            out[key] = out.get(key, 0) + 1
        elif kind == "maxpool":
            n, c, h, w = s0
            k, st, pd = para["w"], para["strides"], para["pads"]
            shapes[dst] = (n, c, (h + 2 * pd[0] - k[0]) // st[0] + 1, (w + 2 * pd[1] - k[1]) // st[1] + 1)
        elif kind == "upsample":
            n, c, h, w = s0
            shapes[dst] = (n, c, h * 2, w * 2)
        elif kind == "concat":
            a, b = shapes[srcs[0]], shapes[srcs[1]]
            shapes[dst] = (a[0], a[1] + b[1], a[2], a[3])
        elif kind in ("gap",):
            shapes[dst] = s0[:2] + (1, 1)
        elif kind == "flatten":
            shapes[dst] = (s0[0], int(np.prod(s0[1:])))
        elif kind == "dense":
            shapes[dst] = (s0[0], shapes[srcs[1]][0])
        elif kind == "return":
            pass
        else:
            shapes[dst] = s0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="resnet18")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--splits", default="1,2,3,4,6,8")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    ctx = planer_amd.hip.context()
    lib = _lib.load()
    ncfg = lib.pl_conv2d_num_configs()
    names = []
    for c in range(ncfg):
        buf = ctypes.create_string_buffer(32)
        lib.pl_conv2d_config_name(c, buf, 32)
        names.append(buf.value.decode())
    if args.model == "resnet18":
        g, _ = resnet18.build()
        xs = (args.batch, 3, 224, 224)
    else:
        g, _ = yolov3.build()
        xs = (args.batch, 3, 416, 416)
    rng = np.random.default_rng(0)
    results = []
    splits = [int(s) for s in args.splits.split(",")]
    for key, mult in conv_shapes(g, xs).items():
        n, cin, h, w, cout, kh, kw, st, pd = key
        x = planer_amd.asarray(rng.standard_normal((n, cin, h, w)).astype(np.float32))
        k = planer_amd.asarray((rng.standard_normal((cout, cin, kh, kw)) * 0.05).astype(np.float32))
        sc = planer_amd.asarray(rng.uniform(0.5, 1.5, (1, cout, 1, 1)).astype(np.float32))
        ho = (h + 2 * pd - kh) // st + 1
        flops = 2.0 * n * cout * ho * ho * cin * kh * kw
        rows = []

        kt = planer_amd.prepare_conv_weights(k) if cin % 16 == 0 else None
        state = {"tap": 0}

        def run():
            return planer_amd.ConvFused(x, kt if state["tap"] else k, None, sc, sc, None, strides=[st, st],
                                        pads=[pd] * 4, act=1, w_layout=state["tap"])

        def timeit():
            run()
            e0 = planer_amd.hip.Event().record()
            for _ in range(args.reps):
                run()
            e1 = planer_amd.hip.Event().record()
            return e0.elapsed_ms(e1) / args.reps

        ctx.set_conv_config(-1, 0)
        state["tap"] = 1 if kt is not None else 0
        auto_ms = timeit()
        for c in range(ncfg):
            tap = names[c].startswith("t")
            if tap and (kt is None or cin % int(names[c].split("x")[-1])):
                continue
            state["tap"] = int(tap)
            for s in splits:
                if s > 1 and cin * kh * kw // s < 64:
                    continue
                ctx.set_conv_config(c, s)
                try:
                    ms = timeit()
                except Exception as e:       # noqa: BLE001
                    print("fail", key, names[c], s, e)
                    continue
                rows.append((ms, names[c], s))
        ctx.set_conv_config(-1, 0)
        rows.sort()
        best = rows[0]
        print("%-44s x%d  auto %.3f ms %5.1f TF | best %s s%d %.3f ms %5.1f TF | next: %s" % (
            str(key), mult, auto_ms, flops / auto_ms / 1e9, best[1], best[2], best[0], flops / best[0] / 1e9,
            ", ".join("%s/s%d %.1f" % (r[1], r[2], flops / r[0] / 1e9) for r in rows[1:5])), flush=True)
        results.append({"shape": key, "mult": mult, "auto_ms": auto_ms, "flops": flops,
                        "rows": [(r[0], r[1], r[2]) for r in rows]})
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(results, f)


if __name__ == "__main__":
    main()
