"""BASELINE.json config 5: YOLO-v3 (Darknet-53 + 3 heads) as planer IR.

75 conv (38 3x3 incl. 5 stride-2, 37 1x1), 72 folded batchnorm + 72
leakyrelu(0.1), 23 residual adds, 2 nearest x2 upsamples (scale tensor
[1,1,2,2] as an init, layer.py:80-82), 2 concat(axis 1), three linear 1x1
heads with bias, `return` of (1,255,13,13),(1,255,26,26),(1,255,52,52) at 416.
247 flow steps; no detection post-processing.
"""
import numpy as np

from .builder import GraphBuilder


class _Gen:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.g = GraphBuilder(["x"])
        self.n = 0

    def cbl(self, src, cin, cout, k, s=1, gain=1.0):
        """conv(no bias) + folded BN + leakyrelu(0.1)"""
        rng, g = self.rng, self.g
        t = "c%d" % self.n
        self.n += 1
        w = rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (cin * k * k))
        gamma = rng.uniform(0.5, 1.5, cout) * gain
        beta = rng.standard_normal(cout) * 0.1
        mean = rng.standard_normal(cout) * 0.1
        var = rng.uniform(0.5, 1.5, cout)
        inv = gamma / np.sqrt(var + 1e-5)
        g.init(t + "_w", w.astype(np.float32))
        g.init(t + "_invK", inv.reshape(1, -1, 1, 1).astype(np.float32))
        g.init(t + "_invB", (beta - mean * inv).reshape(1, -1, 1, 1).astype(np.float32))
        p = k // 2
        g.op("conv", [src, t + "_w"], t + "_c", name=t + "_conv", group=1,
             strides=[s, s], dilations=[1, 1], pads=[p, p, p, p])
        g.op("batchnorm", [t + "_c", t + "_invK", t + "_invB"], t + "_b", name=t + "_bn")
        return g.op("leakyrelu", t + "_b", t + "_a", name=t + "_act", alpha=0.1)

    def head(self, src, cin, tag, classes_ch=255):
        rng, g = self.rng, self.g
        w = rng.standard_normal((classes_ch, cin, 1, 1)) * np.sqrt(1.0 / cin)
        g.init(tag + "_w", w.astype(np.float32))
        g.init(tag + "_bias", (rng.standard_normal(classes_ch) * 0.1).astype(np.float32))
        return g.op("conv", [src, tag + "_w", tag + "_bias"], tag, name=tag + "_conv",
                    group=1, strides=[1, 1], dilations=[1, 1], pads=[0, 0, 0, 0])

    def res(self, src, ch):
        y = self.cbl(src, ch, ch // 2, 1)
        y = self.cbl(y, ch // 2, ch, 3, gain=0.5)
        t = "r%d" % self.n
        return self.g.op("add", [src, y], t, name=t + "_add")


def build(seed=0):
    m = _Gen(seed)
    g = m.g
    g.init("scales", np.array([1, 1, 2, 2], np.float32))
    y = m.cbl("x", 3, 32, 3)
    feats = []
    for ch, reps in [(64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)]:
        y = m.cbl(y, ch // 2, ch, 3, s=2)
        for _ in range(reps):
            y = m.res(y, ch)
        feats.append(y)
    f52, f26, f13 = feats[2], feats[3], feats[4]

    def neck(src, cin, ch):
        y = m.cbl(src, cin, ch, 1)
        y = m.cbl(y, ch, ch * 2, 3)
        y = m.cbl(y, ch * 2, ch, 1)
        y = m.cbl(y, ch, ch * 2, 3)
        return m.cbl(y, ch * 2, ch, 1)

    n13 = neck(f13, 1024, 512)
    o13 = m.head(m.cbl(n13, 512, 1024, 3), 1024, "out13")
    u = m.cbl(n13, 512, 256, 1)
    u = g.op("upsample", [u, "scales"], "up26", name="up26", mode="nearest")
    c26 = g.op("concat", [u, f26], "cat26", name="cat26", axis=1)
    n26 = neck(c26, 768, 256)
    o26 = m.head(m.cbl(n26, 256, 512, 3), 512, "out26")
    u = m.cbl(n26, 256, 128, 1)
    u = g.op("upsample", [u, "scales"], "up52", name="up52", mode="nearest")
    c52 = g.op("concat", [u, f52], "cat52", name="cat52", axis=1)
    n52 = neck(c52, 384, 128)
    o52 = m.head(m.cbl(n52, 128, 256, 3), 256, "out52")
    return g.finish([o13, o26, o52])


def make_input(n=1, seed=1, size=416):
    return np.random.default_rng(seed).standard_normal((n, 3, size, size)).astype(np.float32)
