"""Host-side plan compiler (planer_amd/plan.py) checked on the CPU: the fused
flow, interpreted with the oracle's ops plus a numpy `conv_fused`, must give
the same tensors as the flow as written."""
import numpy as np

from oracle import planer_np as onp
from planer_amd.irgen import customnet, resnet18, yolov3
from planer_amd.plan import ACT_LEAKY, ACT_RELU, ACT_RES_AFTER, assign_layouts, fuse_flow
from tests.conftest import assert_close


def conv_fused_np(x, K, B=None, scale=None, shift=None, res=None, act=0, alpha=0.0, **conv):
    y = np.ascontiguousarray(onp.conv2d(x, K, B, **conv))
    if scale is not None:
        y = y * scale.reshape(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.reshape(1, -1, 1, 1)
    post, act = act & ACT_RES_AFTER, act & 15
    if res is not None and not post:
        y = y + res
    if act == ACT_RELU:
        y = onp.relu(y)
    elif act == ACT_LEAKY:
        y = onp.leakyrelu(y, alpha)
    if res is not None and post:
        y = y + res
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
    # darknet's 23 residual adds follow the leakyrelu: folded as "residual after activation"
    body, flow = _check(yolov3, yolov3.make_input(1, size=64), 144 + 23)
    kinds = [b[1] for b in body]
    assert kinds.count("conv_fused") == 72 and kinds.count("conv") == 3 and "add" not in kinds
    post = [b for b in body if b[1] == "conv_fused" and b[2]["act"] & ACT_RES_AFTER]
    assert len(post) == 23 and all(b[2]["act"] & 15 == ACT_LEAKY for b in post)


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


def test_add_after_activation_needs_a_free_residual_slot():
    # conv -> add -> relu -> add: the first add takes the residual slot, the second stays a layer
    layers = [["c", "conv", {}], ["a1", "add", {}], ["r", "relu", {}], ["a2", "add", {}]]
    flow = [[["x", "K"], ["c"], "t"], [["t", "p"], ["a1"], "u"], ["u", ["r"], "v"], [["v", "q"], ["a2"], "w"]]
    shp = {k: (1, 4, 5, 5) for k in "tpuvqw"}
    body, out, nf = fuse_flow(layers, flow, ["K"], shp)
    assert nf == 2 and [f[1][0] for f in out] == ["c+", "a2"]
    assert body[0][2]["act"] == ACT_RELU


# ---- activation layout assignment (channel-quad plans) ----------------------------------------
def _layout_program(mod, x):
    g, b = mod.build()
    shapes = shapes_of(g, b, x)
    inits = [i[0] for i in g["inits"]]
    body, flow, _ = fuse_flow(g["layers"], g["flow"], inits, shapes)
    body, flow, nq4 = assign_layouts(body, flow, inits, shapes)
    kinds = {b_[0]: b_[1] for b_ in body}
    paras = {b_[0]: b_[2] for b_ in body}
    return [(kinds[f[1][0]], f[0], f[2], paras[f[1][0]]) for f in flow], nq4


def test_layouts_resnet18_one_conversion_in_none_out():
    steps, nq4 = _layout_program(resnet18, resnet18.make_input(1, size=64))
    kinds = [k for k, _, _, _ in steps]
    # the 3-channel stem reads the NCHW input itself (row-packed kernel): no conversion step at all
    assert "to_q4" not in kinds and "from_q4" not in kinds
    assert kinds[0] == "conv_q4" and steps[0][3].get("rowpack") and steps[0][1][0] == "x"
    assert sum(1 for s_ in steps if s_[3].get("rowpack")) == 1
    assert kinds.count("conv_q4") == 20 and "conv_fused" not in kinds
    assert "maxpool_q4" in kinds and "gap_q4" in kinds          # gap hands NCHW to flatten/dense
    assert kinds[-3:] == ["flatten", "dense", "return"]
    # residual operands are read in Q4 directly (no conversions around them)
    assert all(not s.endswith("@nchw") for _, srcs, _, _ in steps for s in srcs)


def test_layouts_yolov3_outputs_are_converted_back():
    steps, _ = _layout_program(yolov3, yolov3.make_input(1, size=64))
    kinds = [k for k, _, _, _ in steps]
    assert "to_q4" not in kinds and steps[0][3].get("rowpack")      # 3 -> 32 first conv: row-packed
    assert "upsample_q4" in kinds and "concat_q4" in kinds
    # the three detection heads (255 channels -> padded quads) come back as NCHW for `return`
    assert kinds.count("from_q4") == 3 and kinds[-1] == "return"
    ret = steps[-1][1]
    assert all(s.endswith("@nchw") for s in ret)


def test_layouts_inplace_relu_invalidates_converted_copies():
    layers = [["c", "conv", {}], ["f", "flatten", {}], ["r", "relu", {}], ["g", "flatten", {}],
              ["return", "return", {}]]
    flow = [[["x", "K"], ["c"], "t"], [["t"], ["f"], "a"], [["t"], ["r"], "u"], [["t"], ["g"], "b"],
            [["a", "b"], ["return"], "plrst"]]
    shp = {"x": (1, 4, 5, 5), "K": (8, 4, 3, 3), "t": (1, 8, 3, 3), "u": (1, 8, 3, 3)}
    body, out, _ = assign_layouts(layers, flow, ["K"], shp, force=True)
    kinds = {b_[0]: b_[1] for b_ in body}
    seq = [kinds[f[1][0]] for f in out]
    # t is converted for the first flatten, mutated in place by relu_q4, and converted AGAIN for the second
    assert seq == ["to_q4", "conv_q4", "from_q4", "flatten", "relu_q4", "from_q4", "flatten", "return"]


def test_layouts_lone_conv_with_a_large_output_stays_nchw():
    # BASELINE config 2: one 3->64 conv on (8,3,224,224); converting its 103 MB output back would
    # cost more than the Q4 kernel saves
    layers = [["c", "conv", {"pads": [1, 1, 1, 1]}]]
    flow = [[["x", "K", "B"], ["c"], "y"]]
    shp = {"x": (8, 3, 224, 224), "K": (64, 3, 3, 3), "B": (64,), "y": (8, 64, 224, 224)}
    body, out, nq4 = assign_layouts(layers, flow, ["K", "B"], shp)
    assert nq4 == 0 and [b_[1] for b_ in body] == ["conv"] and out == [[["x", "K", "B"], ["c"], "y"]]
    body, out, nq4 = assign_layouts(layers, flow, ["K", "B"], shp, force=True)
    assert nq4 == 1 and len(out) == 2 and body[0][2].get("rowpack")           # conv (NCHW in) + from_q4


# ---- random graphs: the plan compiler's rewrites keep the meaning of the flow (CPU, numpy stand-ins for the kernels) ----
def _q4_standins():
    """numpy stand-ins for the plan-internal kinds: a layout conversion is a COPY (the Q4 tensor is another buffer, so
    an in-place ReLU on one side must not leak to the other unless the plan says so), a *_q4 layer is its NCHW op."""
    ops = {"to_q4": lambda x: x.copy(), "from_q4": lambda x: x.copy(),
           "conv_q4": lambda x, K, B=None, scale=None, shift=None, res=None, rowpack=False, w_layout=2, **kw:
               conv_fused_np(x, K, B, scale, shift, res, **kw),
           "conv_fused": conv_fused_np}
    for kind in ("maxpool", "averagepool", "gap", "upsample", "batchnorm", "relu", "leakyrelu", "sigmoid", "add", "concat"):
        ops[kind + "_q4"] = onp.OPS[kind]
    return ops


def _run(graph, blob, x, body, flow):
    saved = dict(onp.OPS)
    onp.OPS.update(_q4_standins())
    try:
        net = onp.OracleNet()
        net.load_json(graph["input"], graph["inits"], body, flow)
    finally:
        onp.OPS.clear()
        onp.OPS.update(saved)
    net.load_weights(blob)
    out = net(x)
    return out if isinstance(out, tuple) else (out,)


def test_random_graphs_fusion_and_layout_assignment_preserve_the_flow():
    import pytest
    from tests.random_nets import random_net
    fused_total = q4_total = 0
    for seed in range(150):
        g, b, xs = random_net(40000 + seed)
        x = xs[0]
        shapes = shapes_of(g, b, x)
        inits = [i[0] for i in g["inits"]]
        want = _run(g, b, x.copy(), g["layers"], g["flow"])
        body, flow, nf = fuse_flow(g["layers"], g["flow"], inits, shapes)
        got = _run(g, b, x.copy(), body, flow)
        body2, flow2, _ = assign_layouts(body, flow, inits, shapes)
        got2 = _run(g, b, x.copy(), body2, flow2)
        body3, flow3, nq4 = assign_layouts(body, flow, inits, shapes, force=True)     # tiny maps: the cost model says no
        got3 = _run(g, b, x.copy(), body3, flow3)
        fused_total += nf
        q4_total += nq4
        for what, outs in (("fused", got), ("fused + layouts", got2), ("fused + forced layouts", got3)):
            assert len(outs) == len(want)
            for o, w in zip(outs, want):
                assert o.shape == w.shape, (seed, what)
                assert_close(np.ascontiguousarray(o), np.ascontiguousarray(w), 1e-5, "seed %d %s" % (seed, what))
    assert fused_total > 150 and q4_total > 300          # the generator does exercise both passes


def test_upsample_concat_peephole_and_its_guards():
    """Net._fuse_upsample_concat (host logic): upsample_q4 -> concat_q4 becomes one upconcat_q4 step, unless the
    upsampled tensor has another reader, is not the concat's first input, or its source is touched in between."""
    from planer_amd.net import Net

    def prog(extra_reader=False, first=True, touch=False):
        body = {"up": ["up", "upsample_q4", {"mode": "nearest"}], "cat": ["cat", "concat_q4", {"axis": 1}],
                "r": ["r", "relu_q4", {}], "s": ["s", "sigmoid_q4", {}]}
        flow = [[["a", "k"], ["up"], "u"]]
        if touch:
            flow.append(["a", ["r"], "a2"])                 # in-place ReLU on the upsample's source before the concat
        flow.append([["u", "b"] if first else ["b", "u"], ["cat"], "c"])
        if extra_reader:
            flow.append(["u", ["s"], "z"])
        return body, flow

    body, flow = prog()
    out = Net._fuse_upsample_concat(body, flow)
    assert out == [[["a", "k", "b"], ["cat"], "c"]] and body["cat"][1] == "upconcat_q4"
    for kw in ({"extra_reader": True}, {"first": False}, {"touch": True}):
        body, flow = prog(**kw)
        assert Net._fuse_upsample_concat(body, flow) == flow and body["cat"][1] == "concat_q4", kw


# ---- Winograd chaining (plan.chain_winograd): pure host logic ------------------------------------------
def _wino_prog():
    """Two BasicBlocks' worth of F(4x4,3x3) convs: c1 -> c2 (+ identity x) -> c3 -> c4 (+ identity y2), then a pool."""
    w7 = {"w_layout": 7, "act": 1, "alpha": 0.0, "pads": [1, 1, 1, 1]}
    body = [["c%d" % i, "conv_q4", dict(w7)] for i in (1, 2, 3, 4)] + [["pool", "maxpool_q4", {}], ["out", "from_q4", {}]]
    flow = [[["x", "U1", "None", "s1", "t1", "None"], ["c1"], "y1"],
            [["y1", "U2", "None", "s2", "t2", "x"], ["c2"], "y2"],
            [["y2", "U3", "None", "s3", "t3", "None"], ["c3"], "y3"],
            [["y3", "U4", "None", "s4", "t4", "y2"], ["c4"], "y4"],
            [["y4"], ["pool"], "p"],
            [["p"], ["out"], "o"]]
    return body, flow


def test_chain_winograd_merges_output_and_input_transforms():
    from planer_amd.plan import chain_winograd
    body, flow = _wino_prog()
    b, f, n = chain_winograd(body, flow)
    kinds = {e[0]: e for e in b}
    assert n == 3
    seq = [(names[0], kinds[names[0]][1]) for _, names, _ in f]
    assert seq == [("c1@in", "wino4_in"), ("c1@gemm", "wino4_gemm"), ("c1@chain", "wino4_chain"),
                   ("c2@gemm", "wino4_gemm"), ("c2@chain", "wino4_chain"), ("c3@gemm", "wino4_gemm"),
                   ("c3@chain", "wino4_chain"), ("c4@gemm", "wino4_gemm"), ("c4@out", "wino4_out"),
                   ("pool", "maxpool_q4"), ("out", "from_q4")]
    # y1 and y3 feed only the next conv: never written; y2 is also c4's residual: written AND transformed
    assert kinds["c1@chain"][2]["keep_y"] is False and f[2][2] == "c2@V"
    assert kinds["c2@chain"][2]["keep_y"] is True and f[4][2] == ["y2", "c3@V"]
    assert kinds["c3@chain"][2]["keep_y"] is False
    assert f[4][0] == ["c2@M", "None", "s2", "t2", "x"]            # the tail's operands travel with the chain step
    assert f[8][0] == ["c4@M", "None", "s4", "t4", "y2"] and kinds["c4@out"][2]["act"] == 1
    # every key is produced before it is read
    have = {"x", "None"} | {"U%d" % i for i in range(1, 5)} | {"s%d" % i for i in range(1, 5)} | {"t%d" % i for i in range(1, 5)}
    for src, _, dst in f:
        assert all(k in have for k in src), (src, sorted(have))
        have |= set(dst if isinstance(dst, list) else [dst])


def test_chain_winograd_stages_only_and_unsupported_maps():
    from planer_amd.plan import chain_winograd
    body, flow = _wino_prog()
    b, f, n = chain_winograd(body, flow, chain=False)
    assert n == 0 and [e[1] for e in b].count("wino4_in") == 4 and [e[1] for e in b].count("wino4_out") == 4
    b, f, n = chain_winograd(body, flow, supported=lambda key: key != "y2")
    assert n == 2 and any(e[0] == "c2@out" for e in b) and any(e[0] == "c3@in" for e in b)


def test_chain_winograd_keeps_an_input_transform_behind_an_in_place_reader():
    """relu_q4 rewrites its input in place (layer.py:46): a conv that reads the tensor AFTER such a step must see the
    rewritten values, so its input transform is not hoisted over it."""
    from planer_amd.plan import chain_winograd
    w7 = {"w_layout": 7, "act": 0, "alpha": 0.0}
    body = [["c1", "conv_q4", dict(w7)], ["r", "relu_q4", {}], ["c2", "conv_q4", dict(w7)], ["a", "add_q4", {}]]
    flow = [[["x", "U1"], ["c1"], "y1"], [["y1"], ["r"], "z"], [["y1", "U2"], ["c2"], "y2"], [["y2", "z"], ["a"], "o"]]
    b, f, n = chain_winograd(body, flow)
    assert n == 0 and [names[0] for _, names, _ in f] == ["c1@in", "c1@gemm", "c1@out", "r", "c2@in", "c2@gemm", "c2@out", "a"]
    # a pure reader in between is fine
    body[1] = ["r", "leakyrelu_q4", {}]
    b, f, n = chain_winograd(body, flow)
    assert n == 1 and f[2][2] == ["y1", "c2@V"]


def test_chain_winograd_leaves_other_convs_alone():
    from planer_amd.plan import chain_winograd
    body = [["c1", "conv_q4", {"w_layout": 8}], ["c2", "conv_q4", {"w_layout": 2}]]
    flow = [[["x", "U1"], ["c1"], "y1"], [["y1", "U2"], ["c2"], "y2"]]
    b, f, n = chain_winograd(body, flow)
    assert n == 0 and b == body and f == [[["x", "U1"], ["c1"], "y1"], [["y1", "U2"], ["c2"], "y2"]]


# ---- sibling convs in one launch (plan.pair_sibling_convs): pure host logic ----------------------------------
def _fork_prog(stride=2, k2=(128, 64, 1, 1)):
    d2 = {"w_layout": 2, "strides": [stride, stride], "pads": [1, 1, 1, 1], "act": 1, "alpha": 0.0, "group": 1, "dilations": [1, 1]}
    dd = dict(d2, pads=[0, 0, 0, 0], act=0)
    body = [["a", "conv_q4", d2], ["b", "conv_q4", {"w_layout": 7}], ["d", "conv_q4", dd], ["s", "add_q4", {}]]
    flow = [[["x", "Ka@q4g1", "None", "sa", "ta", "None"], ["a"], "ya"],
            [["ya", "Kb@wino4q4", "None", "sb", "tb", "None"], ["b"], "yb"],
            [["x", "Kd@q4g1", "None", "sd", "td", "None"], ["d"], "yd"],
            [["yb", "yd"], ["s"], "out"]]
    shapes = {"Ka": (128, 64, 3, 3), "Kb": (128, 128, 3, 3), "Kd": k2}
    return body, flow, (lambda key: shapes.get(key.split("@")[0]))


def test_pair_sibling_convs_merges_the_projection_pattern():
    from planer_amd.plan import pair_sibling_convs
    body, flow, kshape = _fork_prog()
    b, f, n = pair_sibling_convs(body, flow, kshape)
    assert n == 1
    assert f[0] == [["x", "Ka@q4g1", "None", "sa", "ta", "Kd@q4g1", "None", "sd", "td"], ["a&d"], ["ya", "yd"]]
    assert [names[0] for _, names, _ in f] == ["a&d", "b", "s"]
    pair = [e for e in b if e[0] == "a&d"][0]
    assert pair[1] == "conv_q4_pair" and pair[2]["para1"]["act"] == 1 and pair[2]["para2"]["pads"] == [0, 0, 0, 0]


def test_pair_sibling_convs_leaves_other_forks_alone():
    from planer_amd.plan import pair_sibling_convs
    for kw in (dict(stride=1), dict(k2=(128, 64, 3, 3))):                     # stride 1; two 3x3 convs
        body, flow, kshape = _fork_prog(**kw)
        b, f, n = pair_sibling_convs(body, flow, kshape)
        assert n == 0 and [names[0] for _, names, _ in f] == ["a", "b", "d", "s"]
    # the projection is the flow's LAST step (the program's result, net.py:72): hoisting it would make another step last
    body, flow, kshape = _fork_prog()
    b, f, n = pair_sibling_convs(body[:3], flow[:3], kshape)
    assert n == 0 and [names[0] for _, names, _ in f] == ["a", "b", "d"]
    # an in-place ReLU on the shared input between the two convs: the second one must see the rewritten tensor
    body, flow, kshape = _fork_prog()
    body.insert(1, ["r", "relu_q4", {}])
    flow.insert(1, [["x"], ["r"], "xr"])
    b, f, n = pair_sibling_convs(body, flow, kshape)
    assert n == 0
    # a residual on one of them: not a plain fused tail
    body, flow, kshape = _fork_prog()
    flow[2][0][5] = "ya"
    assert pair_sibling_convs(body, flow, kshape)[2] == 0


# ---- 1x1 conv -> staged Winograd conv (plan.fuse_conv1x1_wino_in): pure host logic ---------------------
def _darknet_pair(extra=(), act1=2, res="x"):
    """x -> 1x1 conv (w_layout 2) -> y1 -> 3x3 conv (F(4x4,3x3) staged) + residual x -> y2 [-> extra steps]."""
    body = [["c1", "conv_q4", {"w_layout": 2, "strides": [1, 1], "pads": [0, 0, 0, 0], "act": act1, "alpha": 0.1}],
            ["c2", "conv_q4", {"w_layout": 7, "strides": [1, 1], "pads": [1, 1, 1, 1], "act": 2 | 16, "alpha": 0.1}]]
    flow = [[["x", "K1", "None", "S1", "H1", "None"], ["c1"], "y1"],
            [["y1", "K2", "None", "S2", "H2", res], ["c2"], "y2"]]
    for b, f in extra:
        body.append(b)
        flow.append(f)
    return body, flow


KSHAPES = {"K1": (64, 128, 1, 1), "K2": (128, 64, 3, 3)}


def test_conv1x1_feeding_a_staged_winograd_conv_writes_its_v():
    from planer_amd.plan import chain_winograd, fuse_conv1x1_wino_in
    body, flow = _darknet_pair()
    b, f, _ = chain_winograd(body, flow)
    b, f, n = fuse_conv1x1_wino_in(b, f, KSHAPES.get)
    assert n == 1
    assert [s[1][0] for s in f] == ["c1@v4", "c2@gemm", "c2@out"]
    kinds = {e[0]: e for e in b}
    assert kinds["c1@v4"][1] == "conv1x1_wino_in" and kinds["c1@v4"][2] == {"act": 2, "alpha": 0.1, "wino": 4}
    assert f[0] == [["x", "K1", "None", "S1", "H1"], ["c1@v4"], "c2@V"]
    assert f[1][0] == ["c2@V", "K2"]
    assert "c1" not in kinds and "c2@in" not in kinds


def test_conv1x1_wino_in_guards():
    from planer_amd.plan import chain_winograd, fuse_conv1x1_wino_in

    def fused(body, flow, kshape=KSHAPES.get, small=lambda key: True):
        b, f, _ = chain_winograd(body, flow)
        return fuse_conv1x1_wino_in(b, f, kshape, small)[2]
    # a second reader of the 1x1 conv's output keeps it
    body, flow = _darknet_pair(extra=[(["r", "leakyrelu_q4", {}], [["y1"], ["r"], "z"])])
    assert fused(body, flow) == 0
    # ... and so does being the program's result
    body, flow = _darknet_pair()
    flow.append([["y2", "y1"], ["cat"], "out"])
    body.append(["cat", "concat_q4", {"axis": 1}])
    assert fused(body, flow) == 0
    # strided / padded / grouped / residual-carrying / 3x3 producers, or a map the caller calls large
    for para in ({"strides": [2, 2]}, {"pads": [1, 1, 1, 1]}, {"group": 2}, {"dilations": [2, 2]}, {"w_layout": 7}, {"act": 2 | 16}):
        body, flow = _darknet_pair()
        body[0][2] = dict(body[0][2], **para)
        assert fused(body, flow) == 0, para
    body, flow = _darknet_pair()
    flow[0][0][5] = "x"
    assert fused(body, flow) == 0
    body, flow = _darknet_pair()
    assert fused(body, flow, kshape={"K1": (64, 128, 3, 3), "K2": (128, 64, 3, 3)}.get) == 0
    assert fused(body, flow, kshape={"K1": (62, 128, 1, 1), "K2": (128, 62, 3, 3)}.get) == 0
    assert fused(body, flow, small=lambda key: False) == 0
    assert fused(body, flow) == 1
