"""Whole-net parity on a real MI355X against vectors captured from the
reference (tests/golden) and against the oracle at BASELINE's full sizes."""
import numpy as np
import pytest

from oracle import planer_np as onp
from planer_amd.irgen import customnet, resnet18, save_model, yolov3
from tests.cases import sample_index
from tests.conftest import RTOL, assert_close, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import planer_amd
    planer_amd.hip.context()
    return planer_amd


def check_packed(outs, z, tol=RTOL):
    for i, o in enumerate(outs):
        o = np.ascontiguousarray(o)
        assert tuple(z["out%d_shape" % i]) == o.shape
        idx = sample_index(o.size)
        err = np.abs(o.reshape(-1)[idx] - z["out%d_sample" % i]).max() / float(z["out%d_absmax" % i])
        assert err <= tol, err
        s = z["out%d_sum" % i]
        assert abs(o.astype(np.float64).sum() - s[0]) <= tol * s[1]


@pytest.mark.parametrize("graph_mode", [False, True])
def test_customnet_config1(pa, graph_mode):
    g, b = customnet.build()
    net = pa.from_graph(g, b)
    net.use_graph = graph_mode
    x = customnet.make_input(1)
    y = net(x)                                   # host in -> host out
    assert isinstance(y, np.ndarray) and y.shape == (1, 128, 64, 64)
    check_packed([y], load_golden("customnet_b1.npz"))
    # without the trailing `return`, a batch-1 output loses its batch dim (net.py:101)
    g2 = dict(g, layers=g["layers"][:-1], flow=g["flow"][:-1])
    net2 = pa.from_graph(g2, b)
    net2.use_graph = graph_mode
    y2 = net2(x)
    assert y2.shape == (128, 64, 64)
    check_packed([y2], load_golden("customnet_b1_noreturn.npz"))


@pytest.mark.parametrize("n", [1, 2])
@pytest.mark.parametrize("mode", ["eager", "fused_graph", "unfused_graph"])
def test_resnet18_logits(pa, n, mode):
    g, b = resnet18.build()
    net = pa.from_graph(g, b)
    net.use_graph = mode != "eager"
    net.use_fusion = mode == "fused_graph"
    x = resnet18.make_input(n)
    ref = load_golden("resnet18_b%d.npz" % n)["logits"]
    for _ in range(2):                           # second call replays the captured graph
        y = net(x)
        assert y.shape == (n, 1000)
        assert_close(y, ref, RTOL)
    d = net(pa.asarray(x))                       # device in -> device out, no sync
    assert isinstance(d, pa.DeviceArray)
    assert_close(d.get(), ref, RTOL)


def test_resnet18_stage_activations(pa):
    g, b = resnet18.build()
    z = load_golden("resnet18_b2_stages.npz")
    x = resnet18.make_input(2)
    for key in ("stem_r", "pool", "l11_o", "l21_o", "l31_o", "l41_o", "gap"):
        cut = [i for i, f in enumerate(g["flow"]) if f[2] == key][0]
        net = pa.from_graph(dict(g, flow=g["flow"][:cut + 1]), b)
        o = net.forward(pa.asarray(x)).get()
        assert o.shape == tuple(z[key + "_shape"])
        idx = sample_index(o.size)
        err = np.abs(o.reshape(-1)[idx] - z[key + "_sample"]).max() / float(z[key + "_absmax"])
        assert err <= RTOL, (key, err)


@pytest.mark.parametrize("streams", ["1x1", "2x2", "2x4", "4x4"])
def test_sub_batch_streams_give_identical_results(pa, streams):
    """Fanning a batch out over side streams (concurrent sub-batch graphs) must not change a bit
    relative to... the oracle tolerance, and replays must be stable."""
    g, b = resnet18.build()
    x = resnet18.make_input(8, size=96)
    ref = onp.OracleNet()
    ref.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    ref.load_weights(b)
    want = ref(x.copy())
    net = pa.from_graph(g, b)
    net.streams = streams
    d = pa.asarray(x)
    y1 = net(d).get()
    y2 = net(d).get()
    assert net.compile(d).streams == streams
    assert_close(y1, want, RTOL)
    np.testing.assert_array_equal(y1, y2)
    # tuple outputs (YOLO heads) through the multi-stream path
    if streams == "2x2":
        gy, by = yolov3.build()
        ny = pa.from_graph(gy, by)
        ny.streams = "2x2"
        xy = yolov3.make_input(2, size=96)
        outs = ny(xy)
        ny1 = pa.from_graph(gy, by)
        ny1.streams = "1x1"
        for a, c in zip(outs, ny1(xy)):
            assert_close(a, c, RTOL)


@pytest.mark.parametrize("streams", ["pipe2", "pipe3", "pipe7"])
def test_call_on_a_pipelined_plan_sees_each_new_input(pa, streams, monkeypatch):
    """`net(x)` with PLANER_HIP_STREAMS=pipeR: consecutive calls run on different replicas, and every call must run on
    the input it was given (not on what an older call left in that replica's static buffers)."""
    g, b = resnet18.build()
    ref = onp.OracleNet()
    ref.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    ref.load_weights(b)
    net = pa.from_graph(g, b)
    net.streams = streams
    for seed in range(5 if streams != "pipe7" else 9):        # more calls than replicas: every replica is reused
        x = resnet18.make_input(2, seed=40 + seed, size=64)
        got = net(pa.asarray(x))
        assert_close(got.get(), ref(x.copy()), RTOL, "device call %d" % seed)
        assert_close(net(x), ref(x.copy()), RTOL, "host call %d" % seed)


@pytest.mark.parametrize("streams", ["1x1", "pipe2"])
def test_feeding_a_plan_relays_the_batch_into_the_stem_image(pa, streams, monkeypatch):
    """Where the stem conv is the only reader of a graph input, feeding the plan IS the stem's first pass over the batch:
    (round 5) the stem + max-pool kernel reads the caller's NCHW batch itself and runs at feed time, in front of the captured
    pass (`static.prefed`; W % 4 == 0); with PLANER_HIP_STEM_NCHW=0 the plan keeps a row-packed image beside the static
    input and `feed` re-lays each new batch straight into it (`static.packed`).  Same results as the plan that does everything
    inside the graph (PLANER_HIP_FEED_PACK=0) for batches fed one after another; re-feeding a batch reproduces its result bit
    for bit; batches the caller drops right after feeding are read before their memory is reused."""
    g, b = resnet18.build()
    xs = [pa.asarray(resnet18.make_input(2, seed=70 + s, size=64)) for s in range(4)]
    outs = {}
    for flag, nchw in (("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("PLANER_HIP_FEED_PACK", flag)
        monkeypatch.setenv("PLANER_HIP_STEM_NCHW", nchw)
        net = pa.from_graph(g, b)
        net.streams = streams
        plan = net.compile(xs[0], mode="throughput")
        statics = (plan.replicas[0] if hasattr(plan, "replicas") else plan).inputs
        assert (statics[0].prefed is not None) == (flag == "1" and nchw == "1")
        assert (statics[0].packed is not None) == (flag == "1" and nchw == "0")
        assert any(a["kind"] == "conv_pool_q4" and a["plan"].startswith("stem+maxpool(nchw)" if nchw == "1" else "stem+maxpool ")
                   for a in plan.algos), plan.algos
        got = []
        for x in xs + xs[:2]:
            plan.feed([x])
            plan.launch(join=False)
            held = plan.outputs
            plan.join()
            net.ctx.synchronize()
            got.append((held[0] if isinstance(held, tuple) else held).get())
        outs[flag + nchw] = got
        y = net(xs[3])                                   # the latency path feeds the same way
        np.testing.assert_array_equal((y[0] if isinstance(y, tuple) else y).get(), got[3])
        # asynchronous submits of host batches: the temporaries are dropped at once, the feed kernel must have read them
        pend = [net.submit(resnet18.make_input(2, seed=70 + s, size=64)) for s in range(4)]
        for s, pd in enumerate(pend):
            r = pd.get()
            np.testing.assert_array_equal(r[0] if isinstance(r, tuple) else r, got[s])
    for key in ("10", "01"):
        for a, c in zip(outs["11"], outs[key]):          # (other nets: each times its own launch plans for these small shapes,
            assert_close(a, c, 1e-5, "fed vs " + key)    #  so split-K / algorithm picks -- the summation order -- may differ)
    np.testing.assert_array_equal(outs["11"][0], outs["11"][4])
    assert not np.array_equal(outs["11"][0], outs["11"][1])


@pytest.mark.parametrize("streams", ["pipe2", "pipe3", "auto"])
def test_pipelined_batches_keep_their_own_results(pa, streams):
    """Throughput plans pipeline consecutive batches over several streams (one full-batch graph per
    stream, used round robin).  Batches in flight together must not disturb each other: every one
    matches the oracle, and re-feeding the same batch reproduces its result bit for bit."""
    g, b = resnet18.build()
    ref = onp.OracleNet()
    ref.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    ref.load_weights(b)
    net = pa.from_graph(g, b)
    net.streams = streams
    xs = [resnet18.make_input(4, seed=10 + s, size=96) for s in range(5)]
    want = [ref(x.copy()) for x in xs]
    dev = [pa.asarray(x) for x in xs]
    plan = net.compile(dev[0], mode="throughput")
    if streams != "auto":
        assert plan.streams == streams
    R = len(plan.replicas) if hasattr(plan, "replicas") else 1
    outs = []
    for rnd in range(2):
        for i in range(0, 5, R):
            held = []
            for d in dev[i:i + R]:             # R batches in flight together, no join in between
                plan.feed([d])
                plan.launch(join=False)
                held.append(plan.outputs)      # the replica just launched keeps this batch until reused
            plan.join()
            net.ctx.synchronize()
            outs += [(h[0] if isinstance(h, tuple) else h).get() for h in held]
    for i, o in enumerate(outs):
        assert_close(o, want[i % 5], RTOL, "batch %d" % i)
    for i in range(5):
        np.testing.assert_array_equal(outs[i], outs[i + 5])


def test_resnet18_batch32_vs_oracle(pa):
    """BASELINE config 3 at full size: batch 32, fused + graph path vs the CPU oracle."""
    g, b = resnet18.build()
    x = resnet18.make_input(32)
    net = pa.from_graph(g, b)
    y = net(x)
    ref = onp.OracleNet()
    ref.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    ref.load_weights(b)
    assert_close(y, ref(x.copy()), RTOL)
    # shards are independent (BN is folded): rows of a batch equal single-image runs
    one = net(x[5:6])
    assert_close(one, y[5:6], RTOL)


@pytest.mark.parametrize("size,gold", [(160, "yolov3_b1_160.npz"), (416, "yolov3_b1.npz")])
def test_yolov3_heads(pa, size, gold):
    g, b = yolov3.build()
    net = pa.from_graph(g, b)
    x = yolov3.make_input(1, size=size)
    y = net(x)
    assert isinstance(y, tuple) and len(y) == 3
    check_packed(list(y), load_golden(gold))
    net.use_graph = False
    check_packed(list(net(x)), load_golden(gold))


def test_read_net_formats_and_api(pa, tmp_path, capsys):
    g, b = customnet.build()
    x = customnet.make_input(1)
    save_model(str(tmp_path / "cn"), g, b)
    save_model(str(tmp_path / "cz"), g, b, pla=True)
    n1 = pa.read_net(str(tmp_path / "cn"))
    n2 = pa.InferenceSession(str(tmp_path / "cz"))
    y1, y2 = n1(x), n2({"x": x})                 # dict input (net.py:95)
    np.testing.assert_array_equal(y1, y2)
    assert n1.run(None, {"x": x})[0].shape == (1, 128, 64, 64)   # onnxruntime shim (net.py:79-81)
    assert pa.read_net(str(tmp_path / "nope")) is None
    assert "not found" in capsys.readouterr().out
    n1.use_graph = False
    n1.timeit("start")
    n1(x)
    assert set(n1.timer) == {"conv", "relu", "maxpool", "upsample", "concat", "sigmoid", "return"}
    n1.profile = True
    n1(x)
    assert n1.device_timer["conv"] > 0


def test_relu_mutates_its_input_like_the_reference(pa):
    x = np.random.default_rng(1).standard_normal((2, 3, 4, 4)).astype(np.float32)
    d = pa.asarray(x)
    r = pa.ReLU(d)
    assert r is d
    np.testing.assert_array_equal(d.get(), onp.relu(x.copy()))


def test_postprocessing_flow_with_data_dependent_shapes(pa):
    """conv -> sigmoid -> reshape -> TopK / Greater -> NonZero -> ScatterND in one flow (what an ONNX detection head
    appends to the trunk): NonZero's output shape depends on the data, so the flow cannot live in a hipGraph --
    the net must notice and launch the same kernels one by one, with the oracle's results."""
    rng = np.random.default_rng(5)
    inits = [("K", rng.standard_normal((12, 8, 1, 1)).astype(np.float32)), ("B", rng.standard_normal(12).astype(np.float32)),
             ("shp", np.array([1, 12, -1], np.int64)), ("kk", np.array([5], np.int64)), ("thr", np.array([0.8], np.float32)),
             ("sidx", np.array([[[0, 3], [0, 7], [0, 3]]], np.int64)), ("supd", rng.standard_normal((1, 3, 100)).astype(np.float32))]
    graph = {"input": ["x"], "inits": [[n, list(a.shape), str(a.dtype)] for n, a in inits],
             "layers": [["conv", "conv", {"group": 1, "strides": [1, 1], "dilations": [1, 1], "pads": [0, 0, 0, 0]}],
                        ["sig", "sigmoid", {}], ["rs", "reshape", {}], ["topk", "topk", {"axis": -1, "largest": 1, "sorted": 1}],
                        ["gt", "greater", {}], ["nz", "nonzero", {}], ["sc", "scatternd", {}], ["return", "return", {}]],
             "flow": [[["x", "K", "B"], ["conv"], "c"], ["c", ["sig"], "s"], [["s", "shp"], ["rs"], "r"],
                      [["r", "kk"], ["topk"], ["tv", "ti"]], [["r", "thr"], ["gt"], "m"], ["m", ["nz"], "nzi"],
                      [["r", "sidx", "supd"], ["sc"], "scat"], [["tv", "ti", "nzi", "scat"], ["return"], "plrst"]]}
    blob = np.concatenate([a.reshape(-1).view(np.uint8) for _, a in inits])
    x = rng.standard_normal((1, 8, 10, 10)).astype(np.float32)
    ref = onp.OracleNet()
    ref.load_json(graph["input"], graph["inits"], graph["layers"], graph["flow"])
    ref.load_weights(blob)
    want = ref(x.copy())
    net = pa.from_graph(graph, blob)
    assert net.use_graph
    for rnd in range(2):
        got = net(x)
        assert net.use_graph is False                      # noticed on the first call, eager from then on
        assert len(got) == 4
        assert_close(got[0], want[0], RTOL, "topk values")
        np.testing.assert_array_equal(got[1], want[1])
        assert got[2].dtype == np.int64 and got[2].shape == want[2].shape and want[2].shape[1] > 0
        np.testing.assert_array_equal(got[2], want[2])
        assert_close(got[3], want[3], RTOL, "scatternd")


def test_eager_fallback_rebuilds_its_program_for_a_new_input_shape(pa):
    """A flow that cannot be captured (NonZero) runs its FUSED program eagerly; that program is specialised for the input
    shapes it was fused for (conv algorithm picks, Winograd chaining, row packing), so a second image size must get a
    program of its own -- detection nets are called with varying sizes (round-3 advisor finding)."""
    rng = np.random.default_rng(11)
    inits = [("K1", (rng.standard_normal((16, 8, 3, 3)) * 0.2).astype(np.float32)),
             ("K2", (rng.standard_normal((16, 16, 3, 3)) * 0.2).astype(np.float32)),
             ("thr", np.array([0.5], np.float32))]
    conv = {"group": 1, "strides": [1, 1], "dilations": [1, 1], "pads": [1, 1, 1, 1]}
    graph = {"input": ["x"], "inits": [[n, list(a.shape), str(a.dtype)] for n, a in inits],
             "layers": [["c1", "conv", conv], ["r1", "relu", {}], ["c2", "conv", conv], ["sig", "sigmoid", {}],
                        ["gt", "greater", {}], ["nz", "nonzero", {}], ["return", "return", {}]],
             "flow": [[["x", "K1"], ["c1"], "a"], ["a", ["r1"], "b"], [["b", "K2"], ["c2"], "c"], ["c", ["sig"], "s"],
                      [["s", "thr"], ["gt"], "m"], ["m", ["nz"], "nzi"], [["s", "nzi"], ["return"], "plrst"]]}
    blob = np.concatenate([a.reshape(-1).view(np.uint8) for _, a in inits])
    ref = onp.OracleNet()
    ref.load_json(graph["input"], graph["inits"], graph["layers"], graph["flow"])
    ref.load_weights(blob)
    net = pa.from_graph(graph, blob)
    for shape in ((1, 8, 12, 12), (2, 8, 40, 36), (1, 8, 12, 12), (1, 8, 64, 64)):
        x = rng.standard_normal(shape).astype(np.float32)
        want = ref(x.copy())
        got = net(pa.hip.asarray(x, ctx=net.ctx))
        assert net.use_graph is False
        assert_close(got[0].get(), want[0], RTOL, "sigmoid map %s" % (shape,))
        # positions may only differ where the sigmoid sits within rounding distance of the threshold
        if np.abs(want[0] - 0.5).min() > 1e-5:
            np.testing.assert_array_equal(got[1].get(), want[1])
    assert len(net._eager_prog) == 3


def test_submit_keeps_batches_in_flight_and_matches_call(pa):
    """Net.submit = asynchronous net(x): handles of several batches in flight (more than the plan has replicas) each
    hold their own batch's result, bit-identical to net(x); host arrays in -> host arrays out; the submit loop runs at
    the plan API's rate (what bench.py times)."""
    import time
    g, b = resnet18.build()
    net = pa.from_graph(g, b)
    xs = [resnet18.make_input(8, size=64, seed=s) for s in range(7)]
    ds = [pa.asarray(x) for x in xs]
    want = [net(d).get() for d in ds]
    hs = [net.submit(d) for d in ds]                      # 7 in flight over (at most) 3 replicas
    got = [h.result() for h in reversed(hs)][::-1]        # consumed out of order
    for i, (y, w) in enumerate(zip(got, want)):
        assert isinstance(y, pa.hip.DeviceArray) and y.shape == w.shape
        np.testing.assert_array_equal(y.get(), w, "batch %d" % i)
    yh = net.submit(xs[3]).result()
    assert isinstance(yh, np.ndarray)
    np.testing.assert_array_equal(yh, want[3])
    np.testing.assert_array_equal(net.submit({"x": ds[5]}).get(), want[5])
    # rate at BASELINE's size (GPU-bound): a loop of submits against the plan API (feed + launch of the same plan)
    ds = [pa.asarray(resnet18.make_input(32, seed=s)) for s in range(2)]
    plan = net.compile(ds[0], mode="throughput")

    def loop_plan(k):
        for i in range(k):
            plan.feed([ds[i & 1]])
            plan.launch(join=False)
        plan.join()
        net.ctx.synchronize()

    def loop_submit(k):
        hs = [net.submit(ds[i & 1]) for i in range(k)]
        hs[-1].done()
        net.ctx.synchronize()
    rates = {}
    for name, loop in (("plan", loop_plan), ("submit", loop_submit)):
        loop(50)
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            loop(300)
            best = max(best, 300 / (time.perf_counter() - t0))
        rates[name] = best
    print("passes/s: plan API %.0f, submit %.0f (%.2f)" % (rates["plan"], rates["submit"], rates["submit"] / rates["plan"]))
    assert rates["submit"] >= 0.93 * rates["plan"]


def test_submit_with_temporaries_dropped_right_away(pa):
    """Round-4 advisor finding: the feed of a submitted pass runs asynchronously on the replica's stream, so an input the
    caller drops right after `submit` (a host array's device temporary, the result of a preprocessing kernel) must stay
    alive until the replica has read it -- under GPU-bound load the next allocation on the net's context would take
    its block and overwrite it.  A loop of submits with fresh host arrays and with dropped device temporaries, each
    followed by an allocation of the same size that is filled with garbage, against net(x) one call at a time."""
    g, b = resnet18.build()
    net = pa.from_graph(g, b)
    n, size, rounds = 16, 96, 24
    xs = [resnet18.make_input(n, size=size, seed=s) for s in range(4)]
    want = [net(pa.asarray(x)).get() for x in xs]
    junk = np.full((n, 3, size, size), 1e6, np.float32)
    hs = []
    for i in range(rounds):
        if i & 1:
            hs.append(net.submit(xs[i % 4].copy()))                 # host array in: the device temporary is ours to keep alive
        else:
            d = pa.asarray(xs[i % 4], ctx=net.ctx)
            e = d.copy()                                            # stands for a preprocessing kernel's output
            del d
            hs.append(net.submit(e))
            del e                                                   # a dropped device temporary
        t = pa.asarray(junk, ctx=net.ctx)                           # takes a free block of that size and overwrites it
        del t
    for i, h in enumerate(hs):
        got = h.get()
        np.testing.assert_array_equal(got, want[i % 4], "submit %d" % i)


def test_submit_on_a_sub_batch_plan_copies_on_the_main_stream(pa, monkeypatch):
    """Round-4 advisor finding: with streams="2x2" a submitted pass used to copy its rows on the side streams BEFORE they
    forked behind the net's stream; an input still being produced there could be read early.  Inputs produced by a
    kernel on the main stream right before the submit, many times in a row."""
    g, b = resnet18.build()
    net = pa.from_graph(g, b)
    x = resnet18.make_input(8, size=64, seed=3)
    want = net(pa.asarray(x)).get()
    net2 = pa.from_graph(g, b)
    net2.streams = "2x2"
    big = pa.asarray(np.zeros((64 << 20,), np.float32), ctx=net2.ctx)
    for i in range(6):
        d = pa.hip.empty(x.shape, np.float32, net2.ctx)
        pa._lib.call("pl_memset", net2.ctx.handle, big.ptr, 0, big.nbytes)      # keeps the main stream busy for a while
        d.set(x) if i == 0 else d.copy_from(pa.asarray(x, ctx=net2.ctx))
        h = net2.submit(d)
        assert_close(h.get(), want, 1e-5, "pass %d" % i)      # (batch-4 sub-graphs pick their own kernels: not bit-equal)
    assert net2.compile(pa.asarray(x, ctx=net2.ctx), mode="throughput").streams in ("2x2", "1x1")


def test_two_host_threads_two_contexts(pa):
    """The library is thread-compatible: one context (stream + pool) per host thread, shared launch-plan cache behind a
    mutex, thread-local error text.  Two threads run different nets on their own contexts at the same time."""
    import threading
    jobs = [(customnet, customnet.make_input(1)), (resnet18, resnet18.make_input(2, size=64))]
    want, got, errs = [], [None, None], []
    for mod, x in jobs:
        g, b = mod.build()
        ref = onp.OracleNet()
        ref.load_json(g["input"], g["inits"], g["layers"], g["flow"])
        ref.load_weights(b)
        want.append(ref(x.copy()))

    def work(i):
        try:
            ctx = pa.hip.Context(0)
            mod, x = jobs[i]
            g, b = mod.build()
            net = pa.from_graph(g, b, ctx=ctx)
            for _ in range(10):
                y = net(x.copy())
            got[i] = y
        except Exception as e:                     # noqa: BLE001 -- reported by the main thread
            errs.append(repr(e))
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for y, w in zip(got, want):
        assert_close(y, w, RTOL, "threaded")


def test_launch_without_feed_runs_the_feed_time_part(pa):
    """Round-5 advisor finding: where the stem + max-pool kernel runs at feed time (in front of the captured graph), a caller
    that writes `plan.inputs[0]` itself and calls `plan.launch()` got the logits of the PREVIOUS batch's pooled tensor.
    launch() now runs the feed-time part from the static tensor when no feed preceded it: latency and pipeline plans."""
    g, b = resnet18.build()
    net = pa.from_graph(g, b)
    xs = [resnet18.make_input(4, size=64, seed=s) for s in (1, 2, 3)]
    for mode in ("latency", "throughput"):
        plan = net.compile(pa.asarray(xs[0], ctx=net.ctx), mode=mode)
        reps = getattr(plan, "replicas", [plan])
        fed_at_feed_time = any(s.prefed is not None or s.packed is not None for rp in reps for s in rp.inputs)
        want = []
        for x in xs:                                       # the documented way: feed, then launch
            plan.feed([pa.asarray(x, ctx=net.ctx)])
            plan.launch()
            net.ctx.synchronize()
            want.append(plan.outputs[0].get() if isinstance(plan.outputs, tuple) else plan.outputs.get())
        for x, w in zip(xs[::-1], want[::-1]):             # writing the static input directly, no feed
            rp = reps[getattr(plan, "turn", 0)] if hasattr(plan, "replicas") else plan
            rp.inputs[0].set(x)
            plan.launch()
            net.ctx.synchronize()
            out = plan.outputs[0] if isinstance(plan.outputs, tuple) else plan.outputs
            np.testing.assert_array_equal(out.get(), w, "mode %s (feed-time stem: %s)" % (mode, fed_at_feed_time))
