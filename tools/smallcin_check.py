"""GPU (run with PLANER_HIP_SMALLCIN=1 to select it): the small-Cin 3x3 NCHW conv kernel (BASELINE config 2 and ragged variants) vs the oracle; time and
achieved HBM GB/s of the config-2 shape (algorithmic bytes 107,584,512)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import planer_amd as pa
from planer_amd import hip
from oracle import planer_np as onp
ctx = hip.context()
rng = np.random.default_rng(0)
for (n, c, h, w, co, pad, bias) in [(2, 3, 9, 11, 20, 1, True), (1, 1, 5, 300, 70, 0, False), (3, 4, 17, 16, 64, 1, True),
                                    (2, 2, 40, 7, 130, 1, True), (8, 3, 224, 224, 64, 1, True), (8, 3, 224, 224, 64, 0, True)]:
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    k = (rng.standard_normal((co, c, 3, 3)) * 0.1).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32) if bias else None
    dx, dk, db = pa.asarray(x), pa.asarray(k), (pa.asarray(b) if bias else None)
    run = lambda: pa.Conv2d(dx, dk, db, pads=[pad] * 4)
    y = run().get()
    ref = np.ascontiguousarray(onp.conv2d(x, k, b, pads=[pad] * 4))
    err = float(np.abs(y - ref).max() / np.abs(ref).max())
    for _ in range(3): run()
    best = 1e9
    for _ in range(3):
        e0 = hip.Event(ctx).record()
        for _ in range(10): run()
        e1 = hip.Event(ctx).record()
        best = min(best, e0.elapsed_ms(e1) / 10)
    nbytes = x.nbytes + k.nbytes + (b.nbytes if bias else 0) + ref.nbytes
    print("N%d C%d %dx%d->%d pad %d: rel err %.2e  %.1f us  %.0f GB/s (%.2f of 8 TB/s)  [%s]"
          % (n, c, h, w, co, pad, err, best * 1e3, nbytes / best / 1e6, nbytes / best / 1e6 / 8000, ctx.last_conv_plan()), flush=True)
    assert err <= 1e-4
