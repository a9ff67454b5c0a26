"""Host ndarray in -> host ndarray out (the reference's call contract, net.py:94-101) through the pinned staging rings and
copy streams of csrc/host_stage.hip: the raw C-ABI halves (pl_h2d_staged, pl_d2h_begin / pl_d2h_finish, pl_host_alloc) and the
paths `net(x_host)` / `net.submit(x_host)` take through them.  Everything is compared bit for bit: a copy is a copy."""
import ctypes

import numpy as np
import pytest

from planer_amd.irgen import resnet18

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import planer_amd
    planer_amd.hip.context()
    return planer_amd


@pytest.mark.parametrize("nbytes", [1, 4096, (128 << 10) - 4, (128 << 10), (4 << 20) + 12, 19267584, (33 << 20) + 4])
def test_staged_round_trip_of_every_size_class(pa, nbytes):
    """pl_h2d_staged -> pl_d2h_begin / pl_d2h_finish: below the staging threshold, exactly on it, one chunk plus a ragged tail,
    a ResNet batch (32 x 3 x 224 x 224 floats), several chunks; the source is overwritten right after the call returns."""
    ctx = pa.hip.context()
    rng = np.random.default_rng(nbytes)
    src = rng.integers(0, 256, nbytes, dtype=np.uint8)
    want = src.copy()
    d = pa.hip.empty((nbytes,), np.uint8, ctx)
    pa._lib.call("pl_h2d_staged", ctx.handle, None, d.ptr, src.ctypes.data, nbytes)
    src[...] = 0xEE                                             # the call has read it
    t = d.get_begin()
    assert t is not None
    got = d.get_finish(t)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(d.get(), want)                # (pl_d2h: the waiting form of the same route)


def test_tickets_finish_out_of_order_and_run_out_gracefully(pa):
    ctx = pa.hip.context()
    arrs = [pa.asarray(np.full((300000,), i, np.float32)) for i in range(8)]
    ts = [a.get_begin() for a in arrs]
    for i in (5, 0, 7, 2, 1, 6, 3, 4):
        np.testing.assert_array_equal(arrs[i].get_finish(ts[i]), np.full((300000,), i, np.float32))
    # every pinned buffer in flight: begin says so (ticket -1 -> None) instead of failing, and get() still works
    small = pa.asarray(np.arange(16, dtype=np.float32))
    held = []
    while True:
        t = small.get_begin()
        if t is None:
            break
        held.append(t)
        assert len(held) <= 256
    np.testing.assert_array_equal(small.get(), np.arange(16, dtype=np.float32))
    for t in held:
        small.get_cancel(t)
    t = small.get_begin()
    assert t is not None
    np.testing.assert_array_equal(small.get_finish(t), np.arange(16, dtype=np.float32))
    assert pa._lib.load().pl_d2h_finish(ctx.handle, 12345, None) == pa._lib.PL_EINVAL       # not a ticket in flight


def test_pinned_arrays_are_numpy_arrays_and_skip_the_staging_copy(pa):
    a = pa.hip.pinned_empty((5, 7, 11), np.float32)
    assert isinstance(a, np.ndarray) and a.shape == (5, 7, 11) and a.flags["C_CONTIGUOUS"] and a.flags["WRITEABLE"]
    a[...] = np.arange(a.size, dtype=np.float32).reshape(a.shape)
    d = pa.asarray(a)
    np.testing.assert_array_equal(d.get(), a)
    big = pa.hip.pinned_empty((6 << 20,), np.float32)
    big[...] = np.random.default_rng(0).standard_normal(big.size).astype(np.float32)
    want = big.copy()
    d = pa.hip.empty(big.shape, np.float32)
    d.set_staged(big)                                            # DMA straight out of the caller's memory, in stream order:
    d.ctx.synchronize()                                          # ours again once the consumer's stream has passed the copy
    big[...] = -1.0
    np.testing.assert_array_equal(d.get(), want)
    del a, big                                                   # (finalizers hand the memory back: nothing to assert but no crash)
    w = ctypes.c_int(-1)
    pa._lib.call("pl_copy_threads", ctypes.byref(w))
    assert 0 <= w.value <= 63


def test_submit_with_host_batches_overwritten_right_after_it_returns(pa):
    """The reference-shaped asynchronous path at full speed: pageable numpy batches into `net.submit`, the SAME two host
    buffers refilled as soon as submit returns (the staging copy has read them), results fetched later and out of step --
    every pass must equal the pass over the device-resident batch it was given (same plan), bit for bit; then batches built in pinned memory
    (read in place: refilled only after the pass that took them is done), and handles dropped without get() (their pinned
    tickets go back)."""
    g, b = resnet18.build()
    net = pa.from_graph(g, b)
    n, size, rounds = 16, 96, 30
    xs = [resnet18.make_input(n, size=size, seed=s) for s in range(5)]
    want = [net.submit(pa.asarray(x, ctx=net.ctx)).get() for x in xs]       # the same plan on device-resident batches
    assert all(isinstance(w, np.ndarray) for w in want)
    bufs = [np.empty_like(xs[0]), np.empty_like(xs[0])]
    hs = []
    for i in range(rounds):
        buf = bufs[i & 1]
        buf[...] = xs[i % 5]
        hs.append(net.submit(buf))
        buf[...] = 1e6                                            # garbage the moment the call is back
    for i in reversed(range(rounds)):
        np.testing.assert_array_equal(hs[i].get(), want[i % 5], "pageable, pass %d" % i)
    plan = net.compile(pa.asarray(xs[0], ctx=net.ctx), mode="throughput")
    if plan.streams.startswith("pipe"):                           # (the staged route is the pipeline's)
        assert all(h._tickets is None for h in hs)
    pins = [pa.hip.pinned_empty(xs[0].shape), pa.hip.pinned_empty(xs[0].shape)]
    hs = []
    for i in range(rounds):
        buf = pins[i & 1]
        if i >= 2:
            hs[i - 2].done()                                      # pinned batches are read in place by the DMA: the pass that
        buf[...] = xs[i % 5]                                      # took this buffer two submits ago has to be over first
        hs.append(net.submit(buf))
    for i in range(rounds):
        np.testing.assert_array_equal(hs[i].get(), want[i % 5], "pinned, pass %d" % i)
    for i in range(300):                                          # dropped handles: tickets must come back
        net.submit(xs[i % 5])
    np.testing.assert_array_equal(net.submit(xs[2]).get(), want[2])
    y = net.submit(xs[3]).result()                                # result() of a host submit is a host array too
    assert isinstance(y, np.ndarray)
    np.testing.assert_array_equal(y, want[3])


def test_net_call_with_a_host_batch_and_a_multi_output_net(pa):
    """`net(x_host)` orders its upload on the net's own stream (no host wait before the launch) and reads large outputs back
    through the pinned route: YOLO-v3's three heads at 128 x 128 (1.3 MB together) against the device-array call."""
    from planer_amd.irgen import yolov3
    g, b = yolov3.build()
    net = pa.from_graph(g, b)
    x = yolov3.make_input(2, size=128)
    want = [o.get() for o in net(pa.asarray(x))]
    for _ in range(3):
        xb = x.copy()
        got = net(xb)
        xb[...] = 7.0
        assert len(got) == 3 and all(isinstance(o, np.ndarray) for o in got)
        for o, w in zip(got, want):
            np.testing.assert_array_equal(o, w)
    want = net.submit(pa.asarray(x, ctx=net.ctx)).get()              # (throughput plans pick their own conv algorithms)
    hs = [net.submit(x.copy()) for _ in range(5)]
    for h in hs:
        for o, w in zip(h.get(), want):
            np.testing.assert_array_equal(o, w)


def test_layer_callables_and_nets_under_core_numpy(pa):
    """`planer.core(numpy)` is what the reference does at import (__init__.py:40) and what scripts written against it
    assume: ndarrays into layers and nets, ndarrays out.  Here that selects the ARRAY side only -- the operators still run
    on HIP: a conv, an in-place ReLU (the reference returns its mutated input, layer.py:44-46), a multi-output op and a
    whole net, against the oracle."""
    from oracle import planer_np as onp
    from planer_amd.irgen import customnet
    from tests.conftest import assert_close
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 3, 9, 11)).astype(np.float32)
    K = (rng.standard_normal((8, 3, 3, 3)) * 0.2).astype(np.float32)
    B = rng.standard_normal(8).astype(np.float32)
    try:
        assert pa.core(np, silent=True) is np
        y = pa.Conv2d(x, K, B, pads=(1, 1, 1, 1))
        assert isinstance(y, np.ndarray)
        assert_close(y, np.ascontiguousarray(onp.conv2d(x, K, B, pads=(1, 1, 1, 1))), 1e-5, "conv on host arrays")
        r = x.copy()
        out = pa.ReLU(r)
        assert out is r                                             # the caller's array, mutated
        np.testing.assert_array_equal(r, onp.relu(x.copy()))
        assert np.signbit(r[x < 0]).all()                           # negatives become -0.0 as in the reference (x * (x > 0))
        parts = pa.layer_map["split"](x, split=[1, 2], axis=1)      # a multi-output operator: a list of ndarrays
        assert all(isinstance(p, np.ndarray) for p in parts) and [p.shape[1] for p in parts] == [1, 2]
        g, b = customnet.build()
        net = pa.from_graph(g, b)
        xi = customnet.make_input(1)
        yn = net(pa.asarray(xi))                                    # asarray is numpy's under core(numpy)
        ref = onp.OracleNet()
        ref.load_json(g["input"], g["inits"], g["layers"], g["flow"])
        ref.load_weights(b)
        assert isinstance(yn, np.ndarray)
        assert_close(yn, ref(xi.copy()), 1e-5, "customnet under core(numpy)")
    finally:
        pa.core("hip", silent=True)
    d = pa.asarray(x)
    assert isinstance(d, pa.hip.DeviceArray) and isinstance(pa.Conv2d(d, pa.asarray(K), pa.asarray(B)), pa.hip.DeviceArray)
