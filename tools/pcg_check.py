"""GPU: the persistent producer/consumer conv kernel (configs p*) vs the oracle on ragged shapes, and
vs the autotuned 256-thread plans on the shapes ResNet-18 really runs (device time per launch)."""
import ctypes
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import planer_amd as pa
from planer_amd import q4, hip
from oracle import planer_np as onp

ctx = hip.context()
lib = pa._lib.load()
names = []
for c in range(lib.pl_conv2d_num_configs()):
    buf = ctypes.create_string_buffer(32)
    lib.pl_conv2d_config_name(c, buf, 32)
    names.append(buf.value.decode())
pcs = [n for n in names if n.startswith("p")]
rng = np.random.default_rng(3)


def timeit(run):
    for _ in range(3):
        run()
    best = 1e9
    for _ in range(3):
        e0 = hip.Event(ctx).record()
        for _ in range(10):
            run()
        e1 = hip.Event(ctx).record()
        best = min(best, e0.elapsed_ms(e1) / 10)
    return best


def case(xs, ks, para, tail=True, check=True, label=""):
    x = rng.standard_normal(xs).astype(np.float32)
    k = (rng.standard_normal(ks) * np.sqrt(2.0 / (ks[1] * ks[2] * ks[3]))).astype(np.float32)
    cout = ks[0]
    grp = para.get("group", 1)
    ref = None
    sc = rng.uniform(0.5, 1.5, (1, cout, 1, 1)).astype(np.float32)
    sh = (rng.standard_normal((1, cout, 1, 1)) * 0.1).astype(np.float32)
    xq = q4.to_q4(pa.asarray(x))
    kq = q4.prepare_q4_weights(pa.asarray(k), grp)
    dsc, dsh = (pa.asarray(sc), pa.asarray(sh)) if tail else (None, None)
    y0 = q4.ConvQ4(xq, kq, None, dsc, dsh, None, act=1 if tail else 0, **para)
    rq = None
    if tail:
        res = rng.standard_normal(q4.logical_shape(y0)).astype(np.float32)
        rq = q4.to_q4(pa.asarray(res))
    run = lambda: q4.ConvQ4(xq, kq, None, dsc, dsh, rq, act=1 if tail else 0, **para)
    if check:
        ref = np.ascontiguousarray(onp.conv2d(x, k, **para))
        if tail:
            ref = onp.relu(onp.batchnorm(ref, sc, sh) + res)
    ctx.set_conv_config(-1, 0)
    t_auto = timeit(run)
    plan_auto = ctx.last_conv_plan()
    fl = 2.0 * np.prod(q4.logical_shape(y0)) * ks[1] * ks[2] * ks[3]
    line = "%-34s auto %6.1f us %6.1f TF [%s]" % (label or str(xs), t_auto * 1e3, fl / t_auto / 1e9, plan_auto)
    for n in pcs:
        ctx.set_conv_config(names.index(n), 1)
        try:
            y = q4.from_q4(run()).get()
        except Exception as e:
            line += "\n    %s: %s" % (n, str(e)[:80])
            continue
        used = ctx.last_conv_plan()
        if not used.startswith(n):
            line += "\n    %s: not applicable (%s)" % (n, used.split()[0])
            continue
        err = float(np.abs(y - ref).max() / np.abs(ref).max()) if ref is not None else -1
        t = timeit(run)
        line += "\n    %-12s %6.1f us %6.1f TF  rel err %.1e" % (n, t * 1e3, fl / t / 1e9, err)
        assert ref is None or err <= 1e-4, (n, err)
    ctx.set_conv_config(-1, 0)
    print(line, flush=True)


P1 = dict(pads=[1, 1, 1, 1])
case((2, 20, 13, 11), (70, 20, 3, 3), dict(strides=[2, 2], pads=[1, 1, 1, 1]))
case((3, 32, 14, 14), (40, 32, 3, 3), P1)
case((2, 64, 7, 7), (130, 64, 1, 1), dict())
case((2, 64, 9, 9), (48, 32, 3, 3), dict(pads=[2, 2, 2, 2], dilations=[2, 2], group=2))
case((1, 36 * 16, 10, 3), (36 * 24, 16, 1, 1), dict(group=36), tail=False)
case((5, 24, 17, 19), (300, 24, 3, 3), P1)
big = os.environ.get("BIG", "1") != "0"
if big:
    case((32, 64, 56, 56), (128, 64, 3, 3), dict(strides=[2, 2], pads=[1, 1, 1, 1]), label="l20a 3x3 s2 64->128 @56")
    case((32, 128, 28, 28), (256, 128, 3, 3), dict(strides=[2, 2], pads=[1, 1, 1, 1]), label="l30a 3x3 s2 128->256 @28")
    case((32, 256, 14, 14), (512, 256, 3, 3), dict(strides=[2, 2], pads=[1, 1, 1, 1]), label="l40a 3x3 s2 256->512 @14")
    case((32, 128, 28, 28), (256, 128, 1, 1), dict(strides=[2, 2]), label="l30d 1x1 s2 128->256")
    case((1, 36 * 128, 32 * 7, 7), (36 * 128, 128, 1, 1), dict(group=36), tail=False, check=False, label="wino4 GEMM layer2")
    case((1, 36 * 256, 32 * 4, 4), (36 * 256, 256, 1, 1), dict(group=36), tail=False, check=False, label="wino4 GEMM layer3")
    case((1, 36 * 512, 32 * 2, 2), (36 * 512, 512, 1, 1), dict(group=36), tail=False, check=False, label="wino4 GEMM layer4")
    case((32, 64, 56, 56), (64, 64, 3, 3), P1, label="layer1 3x3 direct")
    case((32, 128, 28, 28), (128, 128, 3, 3), P1, label="layer2 3x3 direct")
