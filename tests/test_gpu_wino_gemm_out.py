"""Winograd F(4x4,3x3) GEMM stage + output transform in one kernel (csrc/wino4_gemm_out_kernel.h, plan.fuse_wino_gemm_out) on a
real MI355X.

Contract: y = Wino4Out(Wino4Gemm(V, U), tail) -- the 3x3 conv of layer.Conv2d (layer.py:22-26 -> util.conv_for util.py:17-44)
with its BatchNorm / LeakyReLU / Add tail (layer.py:125-127, 48-51, 93-95).  The fused kernel sums K in its own order (one fmaf
chain over ascending k-quads instead of the tiled kernel's chunks), so y is compared to 3e-5 of max|y| with the two-kernel path
and to 1e-4 of max|ref| with the oracle, like every conv test."""
import numpy as np
import pytest

from oracle import planer_np as onp
from tests.conftest import RTOL, assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import planer_amd
    planer_amd.hip.context()
    return planer_amd


SHAPES = [  # n, cin, h, w, cout
    (1, 128, 52, 52, 256), (1, 256, 26, 26, 512), (1, 64, 104, 104, 128), (1, 512, 13, 13, 1024),
    (2, 24, 7, 5, 36), (1, 12, 6, 10, 4), (3, 40, 9, 23, 44), (1, 8, 1, 1, 8), (1, 4, 4, 8, 20), (1, 36, 17, 3, 16)]
TAILS = [(), ("b",), ("bn", "leaky"), ("bn", "res", "relu"), ("b", "bn", "leaky", "res", "after")]


def _tail(rng, pa, n, cout, h, w, tail):
    from planer_amd import plan, q4
    host = {"b": None, "scale": None, "shift": None, "res": None}
    if "b" in tail:
        host["b"] = rng.standard_normal(cout).astype(np.float32)
    if "bn" in tail:
        host["scale"] = rng.uniform(0.5, 1.5, (1, cout, 1, 1)).astype(np.float32)
        host["shift"] = (rng.standard_normal((1, cout, 1, 1)) * 0.1).astype(np.float32)
    if "res" in tail:
        host["res"] = rng.standard_normal((n, cout, h, w)).astype(np.float32)
    dev = {k: (None if v is None else pa.asarray(v)) for k, v in host.items()}
    dev["resq"] = None if dev["res"] is None else q4.to_q4(dev["res"])
    act = plan.ACT_RELU if "relu" in tail else plan.ACT_LEAKY if "leaky" in tail else plan.ACT_NONE
    return host, dev, act | (plan.ACT_RES_AFTER if "after" in tail else 0)


@pytest.mark.parametrize("shape", SHAPES, ids=["x".join(map(str, s)) for s in SHAPES])
def test_gemm_out_matches_the_two_stages_and_the_oracle(pa, shape):
    from planer_amd import q4
    n, cin, h, w, cout = shape
    rng = np.random.default_rng(abs(hash(shape)) % (1 << 31))
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    k = (rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).astype(np.float32)
    u = q4.prepare_winograd4_q4_weights(pa.asarray(k))
    v = q4.Wino4In(q4.to_q4(pa.asarray(x)))
    for tail in TAILS:
        host, dev, act = _tail(rng, pa, n, cout, h, w, tail)
        args = (dev["b"], dev["scale"], dev["shift"], dev["resq"])
        got = q4.Wino4GemmOut(v, u, *args, act=act, alpha=0.1)
        assert "gemm+out" in pa.hip.context().last_conv_plan(), pa.hip.context().last_conv_plan()
        want = q4.Wino4Out(q4.Wino4Gemm(v, u), *args, act=act, alpha=0.1)
        assert got.shape == want.shape and got.chan == want.chan
        assert_close(got.get(), want.get(), 3e-5, "one kernel vs two stages %s %s" % (shape, tail))
        ref = onp.conv2d(x, k, host["b"], pads=(1, 1, 1, 1))
        if host["scale"] is not None:
            ref = onp.batchnorm(ref, host["scale"], host["shift"])
        if host["res"] is not None and "after" not in tail:
            ref = onp.add(ref, host["res"])
        ref = onp.relu(np.ascontiguousarray(ref)) if "relu" in tail else onp.leakyrelu(ref, alpha=0.1) if "leaky" in tail else ref
        if host["res"] is not None and "after" in tail:
            ref = onp.add(ref, host["res"])
        assert_close(q4.from_q4(got).get(), np.ascontiguousarray(ref), RTOL, "vs oracle %s %s" % (shape, tail))
    again = q4.Wino4GemmOut(v, u, *args, act=act, alpha=0.1)
    np.testing.assert_array_equal(again.get(), got.get())          # deterministic summation order


def test_plan_uses_the_fused_kernel_on_small_maps(pa, monkeypatch):
    """YOLO-v3 at batch 1 / 160 px: the unchained staged convs of the compiled program run as wino4_gemm_out steps (no wino4_gemm /
    wino4_out pair left for them); the heads equal the plan without the fusion to 1e-5."""
    from planer_amd.irgen import yolov3
    g, b = yolov3.build()
    x = yolov3.make_input(1, size=160)
    outs = {}
    for flag in ("1024", "0"):
        monkeypatch.setenv("PLANER_HIP_WINO_GEMM_OUT", flag)
        net = pa.from_graph(g, b)
        outs[flag] = net(x)
        plan = net.compile(pa.asarray(x, ctx=net.ctx))
        kinds = [a["kind"] for a in plan.algos]
        if flag == "0":
            assert "wino4_gemm_out" not in kinds and net.wino_gemm_out == 0
        else:
            assert net.wino_gemm_out > 0 and kinds.count("wino4_gemm_out") == net.wino_gemm_out
            assert "wino4_gemm" not in kinds
    for a, c in zip(outs["1024"], outs["0"]):
        assert_close(a, c, 1e-5, "fused vs unfused heads")
