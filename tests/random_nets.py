"""Seeded random CNN graphs in planer's IR (shared by the GPU fuzz tests and the CPU plan-compiler tests)."""
import numpy as np

from planer_amd.irgen.builder import GraphBuilder


def random_net(seed):
    r = np.random.default_rng(seed)
    gb = GraphBuilder(["x"])
    n, c0, h0 = int(r.choice([1, 2, 3])), int(r.choice([1, 3, 4, 8])), int(r.choice([8, 12, 16, 20]))
    live = {"x": (c0, h0, h0)}
    uid = [0]

    def new(prefix):
        uid[0] += 1
        return "%s%d" % (prefix, uid[0])

    def pick(pred=lambda s: True):
        names = [k for k, s in live.items() if pred(s)]
        return names[int(r.integers(len(names)))] if names else None

    for _ in range(int(r.integers(4, 12))):
        kind = r.choice(["conv", "conv", "conv", "conv", "pool", "up", "concat", "add", "sigmoid", "relu"])
        src = pick()
        c, h, w = live[src]
        if kind == "conv":
            k = int(r.choice([1, 3, 3, 5])); st = int(r.choice([1, 1, 2])); co = int(r.choice([3, 4, 8, 12, 16, 20, 32]))
            if h < 3 and st == 2:
                st = 1
            K = gb.init(new("K"), (r.standard_normal((co, c, k, k)) * np.sqrt(2.0 / (c * k * k))).astype(np.float32))
            ins = [src, K]
            if r.random() < 0.5:
                ins.append(gb.init(new("B"), (r.standard_normal(co) * 0.1).astype(np.float32)))
            ho = (h + 2 * (k // 2) - k + st) // st
            cur = gb.op("conv", ins, new("c"), group=1, strides=[st, st], dilations=[1, 1], pads=[k // 2] * 4)
            shape = (co, ho, ho)
            if r.random() < 0.7:
                sc = gb.init(new("s"), r.uniform(0.5, 1.5, (1, co, 1, 1)).astype(np.float32))
                sh = gb.init(new("t"), (r.standard_normal((1, co, 1, 1)) * 0.1).astype(np.float32))
                cur = gb.op("batchnorm", [cur, sc, sh], new("b"))
            other = pick(lambda s: s == shape)
            if other is not None and r.random() < 0.5:
                cur = gb.op("add", [cur, other], new("a"))
            act = r.choice(["relu", "leakyrelu", "none"])
            if act == "relu":
                cur = gb.op("relu", cur, new("r"))
            elif act == "leakyrelu":
                cur = gb.op("leakyrelu", cur, new("l"), alpha=0.1)
            live[cur] = shape
        elif kind == "pool" and h >= 4:
            k, st, pd = [(2, 2, 0), (3, 2, 1)][int(r.integers(2))]
            op = "maxpool" if r.random() < 0.7 else "averagepool"
            ho = (h + 2 * pd - k + st) // st
            live[gb.op(op, src, new("p"), w=[k, k], pads=[pd] * 4, strides=[st, st])] = (c, ho, ho)
        elif kind == "up" and h <= 16:
            f = gb.init(new("f"), np.array([1, 1, 2, 2], np.float32))
            live[gb.op("upsample", [src, f], new("u"), mode="nearest")] = (c, 2 * h, 2 * w)
        elif kind == "concat":
            other = pick(lambda s: s[1:] == (h, w))
            live[gb.op("concat", [src, other], new("k"), axis=1)] = (c + live[other][0], h, w)
        elif kind == "add":
            other = pick(lambda s: s == (c, h, w))
            live[gb.op("add", [src, other], new("a"))] = (c, h, w)
        elif kind == "sigmoid":
            live[gb.op("sigmoid", src, new("g"))] = (c, h, w)
        elif kind == "relu" and src != "x":
            live[gb.op("relu", src, new("r"))] = (c, h, w)     # in place in the reference: later readers of `src` see it
    names = [k for k in live if k != "x"] or ["x"]
    outs = [names[-1]]
    if len(names) > 2 and r.random() < 0.5:
        outs.append(names[int(r.integers(len(names) - 1))])
    if r.random() < 0.7:
        src = names[int(r.integers(len(names)))]
        c = live[src][0]
        gp = gb.op("gap", src, new("q"))
        fl = gb.op("flatten", gp, new("v"))
        W = gb.init(new("W"), (r.standard_normal((10, c)) * 0.2).astype(np.float32))
        Bd = gb.init(new("D"), r.standard_normal(10).astype(np.float32))
        outs.append(gb.op("dense", [fl, W, Bd], new("y"), shp=[c, 10]))
    g, blob = gb.finish(outs)
    xs = [r.standard_normal((n, c0, h0, h0)).astype(np.float32) for _ in range(2)]
    return g, blob, xs
