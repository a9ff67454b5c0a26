"""GPU: BASELINE config 2 (Conv2d 3->64 k3 s1 p1 on (8,3,224,224)) on the vector-ALU kernel and its measurement builds
(PLANER_HIP_EXPERIMENT=scv_knock=<mask>: 1 no input loads, 2 no FMAs, 4 no stores, 8 no per-tap filter reads): us per launch (HIP events
around 20 back-to-back launches behind a pre-filled queue, best of 5), GB/s of the 107.58 MB the layer moves, and -- for the
complete kernels -- bit-equality with mode 0.  PLANER_HIP_SCV_CPB=<channels per workgroup> applies to all."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import planer_amd as pa
from planer_amd import hip, _lib
ctx = hip.context()
rng = np.random.default_rng(0)
x = rng.standard_normal((8, 3, 224, 224)).astype(np.float32)
k = (rng.standard_normal((64, 3, 3, 3)) * 0.1).astype(np.float32)
b = rng.standard_normal(64).astype(np.float32)
dx, dk, db = pa.asarray(x), pa.asarray(k), pa.asarray(b)
blocker = hip.empty((256 << 20,), np.float32, ctx)
ref = None
variants = [("shipped", ""), ("no loads", "scv_knock=1"), ("no FMAs", "scv_knock=2"), ("stores only", "scv_knock=3"), ("no stores", "scv_knock=4"),
            ("FMAs + filter reads only", "scv_knock=5"), ("FMAs only", "scv_knock=13")]
for cpb in (os.environ.get("CPBS", "0").split(",")):
    if cpb != "0":
        os.environ["PLANER_HIP_SCV_CPB"] = cpb
    for name, exp in variants:
        os.environ["PLANER_HIP_EXPERIMENT"] = exp
        run = lambda: pa.Conv2d(dx, dk, db, pads=[1] * 4)
        y = run()
        if "knock" not in exp:
            got = y.get()
            if ref is None:
                ref = got
            same = bool(np.array_equal(got, ref))
        else:
            same = None
        best = 1e9
        for _ in range(5):
            for _ in range(8):
                _lib.call("pl_memset", ctx.handle, blocker.ptr, 0, blocker.nbytes)
            e0 = hip.Event(ctx).record()
            for _ in range(20):
                run()
            e1 = hip.Event(ctx).record()
            best = min(best, e0.elapsed_ms(e1) / 20)
        print("cpb %-3s %-28s %6.2f us  %5.2f TB/s of 107.58 MB  bit-equal to mode 0: %s  [%s]" % (
            cpb, name, best * 1e3, 107.58e6 / (best * 1e-3) / 1e12, same, ctx.last_conv_plan()))
