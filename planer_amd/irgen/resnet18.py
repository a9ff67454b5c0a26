"""BASELINE.json configs 3/4: ResNet-18 as planer IR with seeded weights.

Layout as read_onnx would emit it for a torchvision export:
conv (no bias) -> batchnorm (folded K,B of shape (1,C,1,1), eps 1e-5,
io.py:76-91) -> relu; maxpool 3x3 s2 p1; 8 BasicBlocks with add+relu; three
1x1 s2 downsample conv+bn; gap, flatten, dense, return.
70 layers / 70 flow steps / 62 inits / 11,689,512 parameters.
"""
import numpy as np

from .builder import GraphBuilder

BLOB_BYTES = 46758048


class _Gen:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.g = GraphBuilder(["x"])
        self.i = 0

    def conv_bn(self, src, cin, cout, k, s, p, relu, tag):
        rng, g = self.rng, self.g
        w = rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (cin * k * k))
        gamma = rng.uniform(0.5, 1.5, cout)
        beta = rng.standard_normal(cout) * 0.1
        mean = rng.standard_normal(cout) * 0.1
        var = rng.uniform(0.5, 1.5, cout)
        inv = gamma / np.sqrt(var + 1e-5)
        g.init(tag + "_w", w.astype(np.float32))
        g.init(tag + "_invK", inv.reshape(1, -1, 1, 1).astype(np.float32))
        g.init(tag + "_invB", (beta - mean * inv).reshape(1, -1, 1, 1).astype(np.float32))
        g.op("conv", [src, tag + "_w"], tag + "_c", name=tag + "_conv", group=1,
             strides=[s, s], dilations=[1, 1], pads=[p, p, p, p])
        out = g.op("batchnorm", [tag + "_c", tag + "_invK", tag + "_invB"],
                   tag + "_b", name=tag + "_bn")
        if relu:
            out = g.op("relu", out, tag + "_r", name=tag + "_relu")
        return out

    def block(self, src, cin, cout, stride, tag):
        y = self.conv_bn(src, cin, cout, 3, stride, 1, True, tag + "a")
        y = self.conv_bn(y, cout, cout, 3, 1, 1, False, tag + "b")
        if stride != 1 or cin != cout:
            src = self.conv_bn(src, cin, cout, 1, stride, 0, False, tag + "d")
        s = self.g.op("add", [y, src], tag + "_s", name=tag + "_add")
        return self.g.op("relu", s, tag + "_o", name=tag + "_out")


def build(seed=0, classes=1000):
    m = _Gen(seed)
    y = m.conv_bn("x", 3, 64, 7, 2, 3, True, "stem")
    y = m.g.op("maxpool", y, "pool", name="maxpool", w=[3, 3],
               pads=[1, 1, 1, 1], strides=[2, 2])
    cin = 64
    for li, (cout, stride) in enumerate([(64, 1), (128, 2), (256, 2), (512, 2)], 1):
        for bi in range(2):
            y = m.block(y, cin, cout, stride if bi == 0 else 1, "l%d%d" % (li, bi))
            cin = cout
    y = m.g.op("gap", y, "gap", name="gap")
    y = m.g.op("flatten", y, "flat", name="flatten")
    m.g.init("fc_w", (m.rng.standard_normal((classes, 512)) * 0.03).astype(np.float32))
    m.g.init("fc_b", (m.rng.standard_normal(classes) * 0.1).astype(np.float32))
    y = m.g.op("dense", [y, "fc_w", "fc_b"], "logits", name="fc", shp=[512, classes])
    return m.g.finish([y])


def make_input(n, seed=1, size=224):
    return np.random.default_rng(seed).standard_normal((n, 3, size, size)).astype(np.float32)
