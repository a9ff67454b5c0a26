"""Seeded random networks through the whole stack on a real MI355X: IR -> plan compiler (epilogue fusion, channel-quad
layout assignment, algorithm choice) -> captured hipGraph, against the oracle's interpreter of the same IR.

Each case draws a small CNN: convs (1x1 / 3x3 / 5x5, stride 1 / 2, bias or not) optionally followed by batchnorm,
a residual add and an activation (the chains the plan compiler folds into conv epilogues), max / average pooling,
nearest upsampling, concat, add, sigmoid, with tensors re-used by several readers, one to three outputs, one of them
usually gap -> flatten -> dense.  Checked: graph replay, eager interpretation, and a second replay with another input
(plans must not bake data in)."""
import numpy as np
import pytest

from oracle import planer_np as onp
from tests.random_nets import random_net
from tests.conftest import assert_close

pytestmark = pytest.mark.gpu
TOL = 3e-4          # deep random nets: a few fused roundings per layer on top of the 1e-4 per-layer budget


@pytest.fixture(scope="module")
def pa():
    import planer_amd
    planer_amd.hip.context()
    return planer_amd


@pytest.mark.parametrize("seed", range(60))
def test_random_network_plan_vs_oracle(pa, seed):
    g, blob, xs = random_net(40000 + seed)
    ref = onp.OracleNet()
    ref.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    ref.load_weights(blob)
    net = pa.from_graph(g, blob)

    def check(got, want, what):
        got = got if isinstance(got, tuple) else (got,)
        want = want if isinstance(want, tuple) else (want,)
        assert len(got) == len(want), what
        for i, (a, b) in enumerate(zip(got, want)):
            assert a.shape == b.shape, (what, i, a.shape, b.shape)
            assert_close(a, np.ascontiguousarray(b), TOL, "%s seed %d output %d kinds %s"
                         % (what, seed, i, [l[1] for l in g["layers"]]))
    want0, want1 = ref(xs[0].copy()), ref(xs[1].copy())
    check(net(xs[0].copy()), want0, "graph")
    check(net(xs[1].copy()), want1, "graph, second input")
    net.use_graph = False
    check(net(xs[0].copy()), want0, "eager")
    plain = pa.from_graph(g, blob)
    plain.use_q4 = plain.use_fusion = False
    check(plain(xs[1].copy()), want1, "unfused NCHW")
    quad = pa.from_graph(g, blob)
    quad.use_q4 = "force"                      # these maps are small: by its cost estimate the plan would often stay NCHW
    check(quad(xs[0].copy()), want0, "forced channel-quad plan")
    check(quad(xs[1].copy()), want1, "forced channel-quad plan, second input")


@pytest.mark.parametrize("seed", range(0, 60, 3))
def test_random_network_multi_stream_plans(pa, seed):
    """The same random graphs through the multi-stream plans: sub-batch streams (the batch cut in two, each half its own
    captured graph on its own stream) and the throughput pipeline (two whole-batch replicas used round robin)."""
    g, blob, xs = random_net(40000 + seed)
    ref = onp.OracleNet()
    ref.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    ref.load_weights(blob)
    want = [ref(x.copy()) for x in xs]
    want = [w if isinstance(w, tuple) else (w,) for w in want]
    n = xs[0].shape[0]
    if n % 2 == 0:
        net = pa.from_graph(g, blob)
        net.streams = "2x2"
        for x, w in zip(xs, want):
            got = net(x.copy())
            got = got if isinstance(got, tuple) else (got,)
            for a, b in zip(got, w):
                assert_close(a, np.ascontiguousarray(b), TOL, "2x2 seed %d" % seed)
    net = pa.from_graph(g, blob)
    net.streams = "pipe2"
    dev = [pa.asarray(x) for x in xs]
    plan = net.compile(dev[0], mode="throughput")
    held = []
    for d in dev:                                   # both replicas in flight
        plan.feed([d])
        plan.launch(join=False)
        held.append(plan.outputs)
    plan.join()
    net.ctx.synchronize()
    for h, w in zip(held, want):
        h = h if isinstance(h, tuple) else (h,)
        for a, b in zip(h, w):
            assert_close(a.get(), np.ascontiguousarray(b), TOL, "pipe2 seed %d" % seed)
