"""Host-side plan compiler (planer_amd/plan.py) checked on the CPU: the fused
flow, interpreted with the oracle's ops plus a numpy `conv_fused`, must give
the same tensors as the flow as written."""
import numpy as np

from oracle import planer_np as onp
from planer_amd.irgen import customnet, resnet18, yolov3
from planer_amd.plan import ACT_LEAKY, ACT_RELU, fuse_flow
from tests.conftest import assert_close


def conv_fused_np(x, K, B=None, scale=None, shift=None, res=None, act=0, alpha=0.0, **conv):
    y = np.ascontiguousarray(onp.conv2d(x, K, B, **conv))
    if scale is not None:
        y = y * scale.reshape(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.reshape(1, -1, 1, 1)
    if res is not None:
        y = y + res
    if act == ACT_RELU:
        y = onp.relu(y)
    elif act == ACT_LEAKY:
        y = onp.leakyrelu(y, alpha)
    return y


def run_flow(graph, blob, x, body, flow, record=None):
    net = onp.OracleNet()
    ops = dict(onp.OPS, conv_fused=conv_fused_np)
    saved = dict(onp.OPS)
    onp.OPS.update(ops)
    try:
        net.load_json(graph["input"], graph["inits"], body, flow)
    finally:
        onp.OPS.clear()
        onp.OPS.update(saved)
    net.load_weights(blob)
    if record is not None:
        env_shapes = record
        orig = net.forward

        def fwd(*xs):
            out = orig(*xs)
            return out
        net.forward = fwd
    return net(x)


def shapes_of(graph, blob, x):
    """One oracle pass recording the shape of every tensor key."""
    shapes = {k: tuple(s) for k, s, _ in graph["inits"]}
    shapes[graph["input"][0]] = x.shape
    net = onp.OracleNet()
    net.load_json(graph["input"], graph["inits"], graph["layers"], graph["flow"])
    net.load_weights(blob)
    env = dict(zip(net.inits, net.weights))
    env[graph["input"][0]] = x.copy()
    env["None"] = None
    for src, names, dst in graph["flow"]:
        kind, fn, para = net._ops[names[0]]
        args = [env[src]] if isinstance(src, str) else [env.get(k) for k in src]
        val = fn(*args, **para)
        if isinstance(dst, str):
            env[dst] = val
            shapes[dst] = getattr(val, "shape", None)
        else:
            for k, v in zip(dst, val):
                env[k], shapes[k] = v, v.shape
    return shapes


def _check(mod, x, expect_fused):
    g, b = mod.build()
    shapes = shapes_of(g, b, x)
    body, flow, nf = fuse_flow(g["layers"], g["flow"], [i[0] for i in g["inits"]], shapes)
    assert nf == expect_fused
    assert len(flow) == len(g["flow"]) - nf
    ref = run_flow(g, b, x.copy(), g["layers"], g["flow"])
    got = run_flow(g, b, x.copy(), body, flow)
    ref = ref if isinstance(ref, tuple) else (ref,)
    got = got if isinstance(got, tuple) else (got,)
    for r, o in zip(ref, got):
        assert_close(o, r, 1e-5)
    return body, flow


def test_fuse_customnet():
    # conv(+bias) -> relu : the relu is absorbed; its output is read twice afterwards
    body, flow = _check(customnet, customnet.make_input(1), 1)
    assert flow[0][1] == ["conv+"] and flow[0][2] == "r"


def test_fuse_resnet18():
    # 20 conv+bn, 9 relu directly after bn, 8 add, 8 relu after add -> 45 absorbed steps
    body, flow = _check(resnet18, resnet18.make_input(1, size=64), 45)
    kinds = [b[1] for b in body]
    assert kinds.count("conv_fused") == 20 and "batchnorm" not in kinds and "add" not in kinds
    # the downsample branch is produced BEFORE the conv that consumes it as residual
    names = [f[1][0] for f in flow]
    assert names.index("l20d_conv+") < names.index("l20b_conv+")


def test_fuse_yolov3():
    body, flow = _check(yolov3, yolov3.make_input(1, size=64), 144)
    kinds = [b[1] for b in body]
    assert kinds.count("conv_fused") == 72 and kinds.count("conv") == 3
    assert kinds.count("add") == 23          # darknet adds follow a leakyrelu, not a conv


def test_no_fusion_when_intermediate_has_two_readers():
    layers = [["c", "conv", {"pads": [1, 1, 1, 1]}], ["r", "relu", {}], ["a", "add", {}], ["return", "return", {}]]
    flow = [[["x", "K"], ["c"], "t"], ["t", ["r"], "u"], [["t", "u"], ["a"], "v"], [["v"], ["return"], "plrst"]]
    body, out, nf = fuse_flow(layers, flow, ["K"], {})
    assert nf == 0 and [f[1][0] for f in out] == ["c", "r", "a", "return"]


def test_broadcast_add_is_not_fused():
    layers = [["c", "conv", {}], ["a", "add", {}]]
    flow = [[["x", "K"], ["c"], "t"], [["t", "b"], ["a"], "v"]]
    _, out, nf = fuse_flow(layers, flow, ["K", "b"], {"t": (1, 4, 5, 5), "b": (1, 4, 1, 1)})
    assert nf == 0
