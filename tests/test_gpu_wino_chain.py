"""Winograd F(4x4,3x3) stage by stage and chained (csrc/wino4_chain_kernel.h, plan.chain_winograd) on a real MI355X.

The chained kernel must produce exactly what the one-call pipeline produces -- the stages run the same
arithmetic in the same order -- so every comparison with `ConvQ4(w_layout=7)` here is bit for bit; the
comparison with the oracle (util.conv_for, util.py:17-44, + layer.BatchNorm / Add / ReLU) is to 1e-4 of
max|ref| like every conv test."""
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


def _operands(pa, rng, n, cin, h, w, cout, tail):
    from planer_amd import q4
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    k = (rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).astype(np.float32)
    host = {"x": x, "k": k, "b": None, "scale": None, "shift": None, "res": None}
    if "b" in tail:
        host["b"] = rng.standard_normal(cout).astype(np.float32)
    if "bn" in tail:
        host["scale"] = rng.uniform(0.5, 1.5, (1, cout, 1, 1)).astype(np.float32)
        host["shift"] = (rng.standard_normal((1, cout, 1, 1)) * 0.1).astype(np.float32)
    if "res" in tail:
        host["res"] = rng.standard_normal((n, cout, h, w)).astype(np.float32)
    dev = {key: (None if v is None else pa.asarray(v)) for key, v in host.items()}
    dev["xq"] = q4.to_q4(dev["x"])
    dev["resq"] = None if dev["res"] is None else q4.to_q4(dev["res"])
    dev["u"] = q4.prepare_winograd4_q4_weights(dev["k"])
    return host, dev


def _act(tail):
    from planer_amd import plan
    act = plan.ACT_RELU if "relu" in tail else plan.ACT_LEAKY if "leaky" in tail else plan.ACT_NONE
    return act | (plan.ACT_RES_AFTER if "after" in tail else 0)


def _mono(dev, tail, xq=None):
    from planer_amd import q4
    return q4.ConvQ4(dev["xq"] if xq is None else xq, dev["u"], dev["b"], dev["scale"], dev["shift"], dev["resq"],
                     pads=(1, 1, 1, 1), act=_act(tail), alpha=0.1, w_layout=7)


def _oracle(host, tail, x=None):
    y = onp.conv2d(host["x"] if x is None else x, host["k"], host["b"], pads=(1, 1, 1, 1))
    if host["scale"] is not None:
        y = onp.batchnorm(y, host["scale"], host["shift"])
    if host["res"] is not None and "after" not in tail:
        y = onp.add(y, host["res"])
    if "relu" in tail:
        y = onp.relu(np.ascontiguousarray(y))
    elif "leaky" in tail:
        y = onp.leakyrelu(y, alpha=0.1)
    if host["res"] is not None and "after" in tail:
        y = onp.add(y, host["res"])
    return np.ascontiguousarray(y)


SHAPES = [(2, 128, 28, 28, 128), (2, 256, 14, 14, 256), (2, 512, 7, 7, 512), (1, 8, 13, 13, 12), (3, 64, 26, 26, 32),
          (1, 16, 5, 9, 16), (1, 4, 1, 1, 4), (2, 12, 4, 4, 8), (1, 32, 52, 52, 16), (5, 20, 3, 17, 24)]
TAILS = [(), ("b",), ("bn", "relu"), ("bn", "res", "relu"), ("b", "bn", "leaky", "res", "after")]


@pytest.mark.parametrize("shape", SHAPES, ids=["x".join(map(str, s)) for s in SHAPES])
@pytest.mark.parametrize("lds", ["1", "2"], ids=["register-transforms", "lds-transforms"])
def test_stages_equal_the_one_call_pipeline_bit_for_bit(pa, shape, lds, monkeypatch):
    """PLANER_HIP_WINO_LDS=2 sends lone input / output transforms through the LDS kernel too (default: chained ones only)."""
    from planer_amd import q4
    monkeypatch.setenv("PLANER_HIP_WINO_LDS", lds)
    n, cin, h, w, cout = shape
    rng = np.random.default_rng(sum(shape))
    for tail in TAILS:
        host, dev = _operands(pa, rng, n, cin, h, w, cout, tail)
        want = _mono(dev, tail).get()
        v = q4.Wino4In(dev["xq"])
        m = q4.Wino4Gemm(v, dev["u"])
        assert v.meta == (n, cin, h, w) and m.meta == (n, cout, h, w)
        got = q4.Wino4Out(m, dev["b"], dev["scale"], dev["shift"], dev["resq"], act=_act(tail), alpha=0.1)
        assert q4.logical_shape(got) == (n, cout, h, w)
        np.testing.assert_array_equal(got.get(), want, err_msg="%s %s" % (shape, tail))
        assert_close(q4.from_q4(got).get(), _oracle(host, tail), 3e-5, "%s %s" % (shape, tail))


@pytest.mark.parametrize("shape", SHAPES, ids=["x".join(map(str, s)) for s in SHAPES])
def test_chained_pair_equals_two_one_call_convs_bit_for_bit(pa, shape):
    from planer_amd import q4
    n, cin, h, w, cout = shape
    rng = np.random.default_rng(7 + sum(shape))
    for tail in TAILS:
        host1, dev1 = _operands(pa, rng, n, cin, h, w, cout, tail)
        host2, dev2 = _operands(pa, rng, n, cout, h, w, cout, ("bn", "relu"))
        y1_want = _mono(dev1, tail)
        y2_want = _mono(dev2, ("bn", "relu"), xq=y1_want).get()
        if not q4.wino4_chain_supported((n, cout, h, w), dev1["x"].ctx):
            with pytest.raises(NotImplementedError):
                q4.Wino4Chain(q4.Wino4Gemm(q4.Wino4In(dev1["xq"]), dev1["u"]))
            continue
        for keep in (True, False):
            m1 = q4.Wino4Gemm(q4.Wino4In(dev1["xq"]), dev1["u"])
            out = q4.Wino4Chain(m1, dev1["b"], dev1["scale"], dev1["shift"], dev1["resq"], act=_act(tail), alpha=0.1, keep_y=keep)
            y1, v2 = out if keep else (None, out)
            if keep:
                np.testing.assert_array_equal(y1.get(), y1_want.get(), err_msg="y1 %s %s" % (shape, tail))
            y2 = q4.Wino4Out(q4.Wino4Gemm(v2, dev2["u"]), dev2["b"], dev2["scale"], dev2["shift"], None, act=_act(("relu",)))
            np.testing.assert_array_equal(y2.get(), y2_want, err_msg="y2 %s %s keep=%s" % (shape, tail, keep))
        want = _oracle(host2, ("bn", "relu"), x=_oracle(host1, tail))
        assert_close(q4.from_q4(y2).get(), want, RTOL, "%s %s" % (shape, tail))


def test_lds_transforms_refuse_maps_that_do_not_fit(pa):
    from planer_amd import q4
    ctx = pa.hip.context()
    assert q4.wino4_chain_supported((32, 128, 28, 28), ctx)
    assert q4.wino4_chain_supported((32, 512, 7, 7), ctx)
    assert not q4.wino4_chain_supported((32, 64, 112, 112), ctx)
    assert not q4.wino4_chain_supported((1, 6, 8, 8), ctx)          # C % 4


@pytest.mark.parametrize("shape", [(32, 128, 28, 28), (32, 256, 14, 14), (32, 512, 7, 7), (32, 64, 56, 56)],
                         ids=["layer2", "layer3", "layer4", "layer1"])
def test_forced_chain_at_the_real_resnet18_layer_shapes(pa, shape, monkeypatch):
    """A BasicBlock's two 3x3 convs (conv -> bn -> relu -> conv -> bn -> add(identity) -> relu) at batch 32 through the
    plan compiler with F(4x4,3x3) forced: chained where the plane fits the LDS kernel (layer2-4), staged otherwise
    (layer1: 56x56 planes), against the oracle and against the plan without chaining."""
    import planer_amd
    from planer_amd.irgen.builder import GraphBuilder
    n, c, h, w = shape
    rng = np.random.default_rng(c + h)
    gb = GraphBuilder(["x"])
    for i in (1, 2):
        gb.init("K%d" % i, (rng.standard_normal((c, c, 3, 3)) * (2.0 / (9 * c)) ** 0.5).astype(np.float32))
        gb.init("s%d" % i, rng.uniform(0.5, 1.5, (1, c, 1, 1)).astype(np.float32))
        gb.init("t%d" % i, (rng.standard_normal((1, c, 1, 1)) * 0.1).astype(np.float32))
    conv = dict(group=1, strides=[1, 1], dilations=[1, 1], pads=[1, 1, 1, 1])
    gb.op("relu", ["x"], "x0", name="r0")           # in-place on a copy of the input inside the plan
    gb.op("conv", ["x0", "K1"], "c1", name="c1", **conv)
    gb.op("batchnorm", ["c1", "s1", "t1"], "b1", name="b1")
    gb.op("relu", ["b1"], "r1", name="r1")
    gb.op("conv", ["r1", "K2"], "c2", name="c2", **conv)
    gb.op("batchnorm", ["c2", "s2", "t2"], "b2", name="b2")
    gb.op("add", ["b2", "x0"], "a2", name="a2")
    gb.op("relu", ["a2"], "r2", name="r2")
    g, blob = gb.finish(["r2"])
    x = rng.standard_normal(shape).astype(np.float32)
    ref = onp.OracleNet()
    ref.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    ref.load_weights(blob)
    want = ref(x.copy())
    want = want[0] if isinstance(want, tuple) else want
    outs = {}
    for mode in ("1", "stages", "0"):
        monkeypatch.setenv("PLANER_HIP_WINO_CHAIN", mode)
        net = planer_amd.from_graph(g, blob)
        net.force_algo, net.streams, net.use_q4 = 7, "1x1", "force"      # (a lone 56x56 block would stay NCHW by the cost estimate)
        got = net(planer_amd.asarray(x.copy()))
        got = got[0] if isinstance(got, tuple) else got
        outs[mode] = got.get()
        assert_close(outs[mode], want, RTOL, "mode %s" % mode)
        if mode == "1":
            assert net.wino_chains == (1 if h <= 28 else 0)
    np.testing.assert_array_equal(outs["1"], outs["0"])
    np.testing.assert_array_equal(outs["stages"], outs["0"])


def test_resnet18_chained_plan_equals_unchained_plan(pa, monkeypatch):
    import planer_amd
    from planer_amd.irgen import resnet18
    g, b = resnet18.build()
    x = resnet18.make_input(4)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PLANER_HIP_WINO_CHAIN", mode)
        net = planer_amd.from_graph(g, b)
        net.force_algo, net.streams = 7, "1x1"
        outs[mode] = net(planer_amd.asarray(x.copy())).get()
        if mode == "1":
            assert net.wino_chains >= 6, net.wino_chains
    np.testing.assert_array_equal(outs["1"], outs["0"])
    ref = onp.OracleNet()
    ref.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    ref.load_weights(b)
    assert_close(outs["1"], ref(x.copy()), RTOL, "resnet18 chained")
