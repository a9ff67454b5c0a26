"""Parity of exactly what bench.py times, at BASELINE's full sizes, on a real MI355X.

* the throughput plan (pipe2 / pipe3 / pipe7, batch 32, 224x224, algorithms picked by timing): every
  replica's logits vs the oracle (BASELINE configs[2]);
* every 3x3 algorithm (direct, fused 1-D F(2,3) / F(4,3), F(2x2,3x3), F(4x4,3x3)) forced at every
  real ResNet-18 stride-1 layer shape at batch 32, with the real fused tail (bn + residual + relu),
  and forced through the whole net;
* BASELINE configs[3]: batch 256 cut into 8 x 32 by dist.shard_range, each shard on its own
  "virtual rank" (SURVEY 8(e)), assembled (256, 1000) vs the oracle and vs one batch-256 pass.

Tolerance: per tensor max|y - ref| <= 1e-4 * max|ref| (north_star; tests/conftest.RTOL); the
elementwise figure SURVEY 8(c) also asks for -- allclose(rtol=1e-4, atol=1e-4*max|ref|) -- is
asserted and its worst ratio printed.
"""
import numpy as np
import pytest

from oracle import planer_np as onp
from planer_amd.irgen import resnet18
from tests.conftest import RTOL, assert_close, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import planer_amd
    planer_amd.hip.context()
    return planer_amd


@pytest.fixture(scope="module")
def r18():
    g, b = resnet18.build()
    ref = onp.OracleNet()
    ref.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    ref.load_weights(b)
    return g, b, ref


def elementwise_ratio(y, ref):
    """max over elements of |y-ref| / (1e-4*|ref| + 1e-4*max|ref|): <= 1 means
    allclose(rtol=1e-4, atol=1e-4*max|ref|) holds."""
    y, ref = np.asarray(y, np.float64), np.asarray(ref, np.float64)
    return float((np.abs(y - ref) / (RTOL * np.abs(ref) + RTOL * np.abs(ref).max())).max())


@pytest.mark.parametrize("streams", ["pipe2", "pipe3", "pipe7"])      # pipe7: what the shipped database picks for this workload
def test_throughput_plan_batch32_full_size(pa, r18, streams):
    g, b, ref = r18
    net = pa.from_graph(g, b)
    net.streams = streams
    R = int(streams[4:])
    xs = [resnet18.make_input(32, seed=40 + i) for i in range(R)]
    want = [ref(x.copy()) for x in xs]
    dev = [pa.asarray(x) for x in xs]
    plan = net.compile(dev[0], mode="throughput")
    assert plan.streams == streams and len(plan.replicas) == R
    # what ran is on record: one entry per conv / dense step with its kernel family and tile plan
    # (wino4_gemm: a staged F(4x4,3x3) conv; conv_q4_pair: two sibling convs in one launch; conv_pool_q4: stem conv + max-pool)
    convs = [a for a in plan.algos if a["kind"] in ("conv_q4", "wino4_gemm", "wino43_gemm", "conv_q4_pair", "conv_pool_q4")]
    assert sum(2 if a["kind"] == "conv_q4_pair" else 1 for a in convs) == 20 and all(a["plan"] for a in convs), plan.algos
    for rnd in range(2):
        held = []
        for d in dev:                          # R batches in flight together
            plan.feed([d])
            plan.launch(join=False)
            held.append(plan.outputs)
        plan.join()
        net.ctx.synchronize()
        for i, h in enumerate(held):
            y = (h[0] if isinstance(h, tuple) else h).get()
            assert y.shape == (32, 1000)
            assert_close(y, want[i], RTOL, "%s replica %d round %d" % (streams, i, rnd))
            r = elementwise_ratio(y, want[i])
            print("%s replica %d: rel err %.2e, elementwise ratio %.3f" % (streams, i, rel_err(y, want[i]), r))
            assert r <= 1.0


def test_throughput_plans_take_the_pipeline_judged_picks(pa, r18):
    """Round 5: throughput plans ask the database's `algo_throughput` table first (picks made under the seven-replica pipeline,
    tools/pipeline_search.py) while the latency plan behind net(x) keeps the isolated picks.  Round 6 (16-tile blocks of the fused
    F(4x4,3x3) kernel): the fused kernel is the isolated pick on layer1-3 at batch 32 and the table moves layer3 to the mixed-tile
    staged form (+3 % pipelined).  Whatever the table holds: the convs of the stages it names run its algorithm in the throughput
    plan and the isolated pick in the latency plan, every other step is the same in both.  Both match the oracle; a net without
    the table falls back to the isolated picks in both modes."""
    g, b, ref = r18
    x = resnet18.make_input(32, seed=77)
    want = ref(x.copy())
    d = pa.asarray(x)
    net = pa.from_graph(g, b)
    net._load_algo_cache()
    if not net._algo_tp:
        # the shipped table is empty when the isolated picks win under the pipeline too (round 6): the mechanism is exercised
        # with a table of this test's own -- layer3 (mixed tiles alone) on the fused kernel for throughput plans only
        for sig, v in list(net._algo.items()):
            if sig[1][:2] == (32, 256) and len(sig[2]) == 4 and sig[2][2:] == (3, 3) and sig[1][2] == 14:
                net._algo_tp[sig] = 9 if v != 9 else 7
        assert net._algo_tp, "no layer3 signature in the database"
    lay = lambda plan: {a["layer"].split("@")[0].rstrip("+"): a["w_layout"] for a in plan.algos}
    tp = net.compile(d, mode="throughput")
    lat = net.compile(d, mode="latency")
    moved = {k for k in lay(tp) if lay(tp)[k] != lay(lat).get(k)}
    table = {(sig[1][1], sig[1][2]): v for sig, v in net._algo_tp.items() if sig[1][0] == 32}       # (Cin, H) at batch 32 -> algorithm
    stage = {64: "l1", 128: "l2", 256: "l3", 512: "l4"}
    named = {stage[c] for (c, _h), v in table.items() if c in stage}
    assert moved and all(k[:2] in named for k in moved), (moved, named)
    for k in moved:
        cin = {v: c for c, v in stage.items()}[k[:2]]
        assert lay(tp)[k] == [v for (c, _h), v in table.items() if c == cin][0], (k, lay(tp)[k], table)
    # ... and the stem + max-pool kernel runs strips of 14 pooled rows on half the workgroups there (7 rows in the latency plan)
    stem = lambda plan: [a["plan"] for a in plan.algos if a["kind"] == "conv_pool_q4"][0]
    assert "of 14 rows" in stem(tp) and "of 7 rows" in stem(lat), (stem(tp), stem(lat))
    tp.feed([d]); tp.launch(join=False); tp.join(); net.ctx.synchronize()
    o = tp.outputs
    assert_close((o[0] if isinstance(o, tuple) else o).get(), want, RTOL, "throughput plan")
    y = net(d)
    assert_close((y[0] if isinstance(y, tuple) else y).get(), want, RTOL, "latency plan")
    bare = pa.from_graph(g, b)
    bare._load_algo_cache()
    bare._algo_tp.clear()
    assert lay(bare.compile(d, mode="throughput")) == lay(lat)


LAYER_SHAPES = [(64, 56), (128, 28), (256, 14), (512, 7)]       # ResNet-18 layer1..4 stride-1 3x3 convs
ALGOS = [2, 8, 4, 7, 9]      # 9 = the fully fused F(4x4,3x3) kernel (conv_wf4_kernel), what layer1 runs


@pytest.mark.parametrize("chan,size", LAYER_SHAPES, ids=["layer%d" % (i + 1) for i in range(4)])
def test_every_algorithm_at_real_layer_shapes_batch32(pa, chan, size):
    from planer_amd import q4
    rng = np.random.default_rng(chan + size)
    x = rng.standard_normal((32, chan, size, size)).astype(np.float32)
    k = (rng.standard_normal((chan, chan, 3, 3)) * np.sqrt(2.0 / (9 * chan))).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, (1, chan, 1, 1)).astype(np.float32)
    sh = (rng.standard_normal((1, chan, 1, 1)) * 0.1).astype(np.float32)
    res = rng.standard_normal((32, chan, size, size)).astype(np.float32)
    conv = np.ascontiguousarray(onp.conv2d(x, k, pads=[1, 1, 1, 1]))
    want = {True: onp.relu(onp.batchnorm(conv, sc, sh) + res), False: onp.relu(onp.batchnorm(conv, sc, sh))}
    xq, rq = q4.to_q4(pa.asarray(x)), q4.to_q4(pa.asarray(res))
    dk, dsc, dsh = pa.asarray(k), pa.asarray(sc), pa.asarray(sh)
    prep = {2: q4.prepare_q4_weights, 8: q4.prepare_w1d4_q4_weights,
            4: q4.prepare_winograd_q4_weights, 7: q4.prepare_winograd4_q4_weights, 9: q4.prepare_wf4_q4_weights}
    for lay in ALGOS:
        kq = prep[lay](dk)
        for with_res in (True, False):
            yq = q4.ConvQ4(xq, kq, None, dsc, dsh, rq if with_res else None, pads=[1, 1, 1, 1], act=1, w_layout=lay)
            y = q4.from_q4(yq).get()
            e, r = rel_err(y, want[with_res]), elementwise_ratio(y, want[with_res])
            print("C%d %dx%d w_layout %d res=%d: rel err %.2e, elementwise ratio %.3f [%s]"
                  % (chan, size, size, lay, with_res, e, r, pa.hip.context().last_conv_plan()))
            assert e <= RTOL and r <= 1.0, (lay, with_res, e, r)


@pytest.mark.parametrize("lay", ALGOS)
def test_whole_net_with_one_forced_algorithm_batch32(pa, r18, lay):
    """Every eligible 3x3/s1 conv of ResNet-18 forced onto one algorithm, real fused tails, batch 32."""
    g, b, ref = r18
    x = resnet18.make_input(32, seed=7)
    net = pa.from_graph(g, b)
    net.force_algo = lay
    net.streams = "1x1"
    d = pa.asarray(x)
    y = net(d).get()
    used = []
    for a in net.compile(d).algos:
        if a["kind"] in ("conv_q4", "wino4_gemm", "wino43_gemm", "conv_q4_pair"):
            used += [a["w_layout"]] * (2 if a["kind"] == "conv_q4_pair" else 1)
    # 13 stride-1 3x3 convs; the 3 stride-2 3x3 and the 3 1x1 convs are always direct (w_layout 2)
    assert used.count(lay) == (13 if lay != 2 else 19), used
    want = ref(x.copy())
    assert_close(y, want, RTOL, "forced w_layout %d" % lay)
    assert elementwise_ratio(y, want) <= 1.0


def test_config4_batch256_on_eight_virtual_ranks(pa, r18):
    """BASELINE configs[3]: ResNet-18 batch 256 sharded 8 x 32 (net.py:94-101 per shard)."""
    from planer_amd import dist
    g, b, ref = r18
    x = resnet18.make_input(256, seed=4)
    vw = dist.VirtualWorld(8)
    nets = vw.load(g, b)                        # rank 0 uploads, ranks 1..7 get the device "broadcast"
    for n in nets:
        n.streams = "1x1"
    y = vw.forward(nets, x)
    assert y.shape == (256, 1000)
    for r in range(8):                          # oracle per shard keeps the im2col scratch small
        lo, hi = dist.shard_range(256, 8, r)
        assert (lo, hi) == (32 * r, 32 * r + 32)
        assert_close(y[lo:hi], ref(x[lo:hi].copy()), RTOL, "rank %d" % r)
    # equals ONE batch-256 pass on a single rank (BN is folded: shards are independent)
    full = nets[0](x)
    assert_close(y, full, RTOL, "sharded vs single pass")
    # a ragged global batch: 250 images -> the first 2 ranks take one extra
    yr = vw.forward(nets, x[:250])
    assert yr.shape == (250, 1000)
    assert_close(yr, full[:250], RTOL, "ragged shards")
