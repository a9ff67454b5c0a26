"""GPU: fused 1-D Winograd F(4,3) conv (w_layout 8) -- parity vs the oracle on ragged and real ResNet
shapes, and device time per launch.  PLANER_HIP_W1D4_PC=0/1 picks the 256-thread kernel or the
persistent producer/consumer kernel (read once per process)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import planer_amd as pa
from planer_amd import q4, hip
from oracle import planer_np as onp

ctx = hip.context()
rng = np.random.default_rng(5)
shapes = [(2, 16, 9, 11, 24), (3, 32, 13, 17, 70), (1, 48, 5, 30, 64), (32, 64, 56, 56, 64), (32, 128, 28, 28, 128),
          (32, 256, 14, 14, 256), (32, 512, 7, 7, 512), (128, 64, 56, 56, 64)]
lay = int(os.environ.get("LAY", "8"))
for n, c, h, w, co in shapes:
    if lay in (4, 7) and co % 4:
        continue
    if lay in (4, 7) and co % 4:
        continue
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    k = (rng.standard_normal((co, c, 3, 3)) * np.sqrt(2.0 / (9 * c))).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, (1, co, 1, 1)).astype(np.float32)
    sh = (rng.standard_normal((1, co, 1, 1)) * 0.1).astype(np.float32)
    res = rng.standard_normal((n, co, h, w)).astype(np.float32)
    xq, rq = q4.to_q4(pa.asarray(x)), q4.to_q4(pa.asarray(res))
    prep = {8: q4.prepare_w1d4_q4_weights, 5: q4.prepare_w1d_q4_weights, 2: q4.prepare_q4_weights,
            7: q4.prepare_winograd4_q4_weights}[lay]
    kq, dsc, dsh = prep(pa.asarray(k)), pa.asarray(sc), pa.asarray(sh)
    run = lambda: q4.ConvQ4(xq, kq, None, dsc, dsh, rq, pads=[1, 1, 1, 1], act=1, w_layout=lay)
    y = q4.from_q4(run()).get()
    err = -1.0
    if n <= 32:
        ref = onp.relu(onp.batchnorm(np.ascontiguousarray(onp.conv2d(x, k, pads=[1, 1, 1, 1])), sc, sh) + res)
        err = float(np.abs(y - ref).max() / np.abs(ref).max())
    for _ in range(3):
        run()
    best = 1e9
    for _ in range(3):
        e0 = hip.Event(ctx).record()
        for _ in range(10):
            run()
        e1 = hip.Event(ctx).record()
        best = min(best, e0.elapsed_ms(e1) / 10)
    fl = 2.0 * n * co * h * w * c * 9
    print("N%d C%d %dx%d->%d  rel err %.2e  %.1f us  %.1f TFLOP/s algorithmic  [%s]"
          % (n, c, h, w, co, err, best * 1e3, fl / best / 1e9, ctx.last_conv_plan()), flush=True)
    assert err <= 1e-4
