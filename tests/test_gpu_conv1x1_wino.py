"""1x1 conv + Winograd input transform in one kernel (csrc/conv1x1_wino_in_kernel.h, plan.fuse_conv1x1_wino_in) on a real MI355X.

Contract: V = Wino4In(ConvQ4(x, K1, tail)) -- the 1x1 conv of layer.Conv2d (layer.py:22-26) with its BatchNorm / LeakyReLU tail
(layer.py:125-127, 48-51), then B^T d B of the 6x6 patches of the zero-padded result.  The 1x1 conv's K summation order is the
fused kernel's own (eight K slices in wave order), so V is compared to 1e-5 of max|V| with the two-kernel path and the whole
1x1 -> 3x3 pair to 1e-4 of max|ref| with the oracle."""
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
    (1, 256, 52, 52, 128), (1, 512, 26, 26, 256), (1, 1024, 13, 13, 512), (1, 128, 104, 104, 64),
    (2, 24, 7, 5, 36), (1, 12, 6, 10, 4), (3, 40, 9, 23, 44), (1, 8, 1, 1, 8), (1, 64, 4, 8, 32)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("tail", ["bn+leaky", "b", "bn+relu", "none"])
def test_conv1x1_wino_in_matches_conv_then_transform(pa, shape, tail):
    from planer_amd import plan, q4
    n, cin, h, w, cout = shape
    rng = np.random.default_rng(abs(hash((shape, tail))) % (1 << 31))
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    k1 = (rng.standard_normal((cout, cin, 1, 1)) * (2.0 / cin) ** 0.5).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) if "b" in tail.split("+") else None
    sc = rng.uniform(0.5, 1.5, (1, cout, 1, 1)).astype(np.float32) if "bn" in tail else None
    sh = (rng.standard_normal((1, cout, 1, 1)) * 0.1).astype(np.float32) if "bn" in tail else None
    act = plan.ACT_LEAKY if "leaky" in tail else plan.ACT_RELU if "relu" in tail else plan.ACT_NONE
    dev = lambda a: None if a is None else pa.asarray(a)
    xq, kq = q4.to_q4(dev(x)), q4.prepare_q4_weights(dev(k1))
    db, dsc, dsh = dev(b), dev(sc), dev(sh)
    y = q4.ConvQ4(xq, kq, db, dsc, dsh, None, act=act, alpha=0.1, w_layout=2)
    want = q4.Wino4In(y).get()
    got = q4.Conv1x1WinoIn(xq, kq, db, dsc, dsh, act=act, alpha=0.1, wino=4)
    assert got.meta == (n, cout, h, w)
    got = got.get()
    assert got.shape == want.shape
    assert_close(got, want, 1e-5, "V %s %s" % (shape, tail))
    # the pair 1x1 -> 3x3 through the fused kernel, against the oracle
    k2 = (rng.standard_normal((cout, cout, 3, 3)) * (2.0 / (9 * cout)) ** 0.5).astype(np.float32)
    u = q4.prepare_winograd4_q4_weights(dev(k2))
    v = q4.Conv1x1WinoIn(xq, kq, db, dsc, dsh, act=act, alpha=0.1, wino=4)
    out = q4.from_q4(q4.Wino4Out(q4.Wino4Gemm(v, u))).get()
    r = onp.conv2d(x, k1, b)
    if sc is not None:
        r = onp.batchnorm(r, sc, sh)
    r = onp.leakyrelu(r, 0.1) if act == plan.ACT_LEAKY else onp.relu(r) if act == plan.ACT_RELU else r
    ref = onp.conv2d(r, k2, None, pads=(1, 1, 1, 1))
    assert_close(out, ref, RTOL, "pair %s %s" % (shape, tail))


def test_plan_uses_the_fused_kernel_for_darknet_blocks(pa, monkeypatch):
    """YOLO-v3 at batch 1 / 160 px: the compiled program holds conv1x1_wino_in steps (and no input transform for those convs);
    its heads equal the plan without the fusion to 1e-5 and the reference fixtures within the usual bar (test_gpu_nets)."""
    from planer_amd.irgen import yolov3
    g, b = yolov3.build()
    x = yolov3.make_input(1, size=160)
    outs = {}
    for flag in ("4096", "0"):
        monkeypatch.setenv("PLANER_HIP_CONV1X1_WINO", flag)
        net = pa.from_graph(g, b)
        outs[flag] = net(x)
        prog = net._fuse({k: a.shape for k, a in zip(net.inits, net.weights)} | _shapes(net, x))[0]
        kinds = [o.name for o in prog.objs.values()]
        if flag == "0":
            assert "conv1x1_wino_in" not in kinds
        else:
            assert net.conv_wino_fused == kinds.count("conv1x1_wino_in")
            outs["n"] = net.conv_wino_fused
    for a, c in zip(outs["4096"], outs["0"]):
        assert_close(a, c, 1e-5, "fused vs unfused heads")


def _shapes(net, x):
    shapes = {k: a.shape for k, a in zip(net.inits, net.weights)}
    shapes[net.input[0]] = x.shape
    net._interpret(net._program, [net_asarray(net, x)], shapes=shapes)
    return shapes


def net_asarray(net, x):
    import planer_amd
    return planer_amd.asarray(x, ctx=net.ctx)
