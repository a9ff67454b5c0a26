"""BASELINE.json config 1: the README's CustomNet in the current 4-arg IR.

readme.md:25-55 shows a class-style API that no longer exists; SURVEY §8(d)
re-expresses it against layer.py's function API: Conv2d 3->64 k3 s1 pad 1 +
ReLU -> Maxpool 2x2 -> UpSample x2 -> Concat(axis 1) -> Sigmoid -> return.
"""
import numpy as np

from .builder import GraphBuilder


def build(seed=0):
    rng = np.random.default_rng(seed)
    g = GraphBuilder(["x"])
    g.init("K", (rng.standard_normal((64, 3, 3, 3)) * 0.1).astype(np.float32))
    g.init("B", rng.standard_normal(64).astype(np.float32))
    g.init("k", np.array([1, 1, 2, 2], np.float32))
    g.op("conv", ["x", "K", "B"], "c", name="conv", group=1, strides=[1, 1],
         dilations=[1, 1], pads=[1, 1, 1, 1])
    g.op("relu", "c", "r", name="relu")
    g.op("maxpool", "r", "p", name="pool", w=[2, 2], pads=[0, 0, 0, 0],
         strides=[2, 2])
    g.op("upsample", ["p", "k"], "u", name="up", mode="nearest")
    g.op("concat", ["r", "u"], "z", name="concat", axis=1)
    g.op("sigmoid", "z", "s", name="sigmoid")
    return g.finish(["s"])


def make_input(n=1, size=64, seed=0):
    rng = np.random.default_rng(seed + 1000)
    return rng.standard_normal((n, 3, size, size)).astype(np.float32)
