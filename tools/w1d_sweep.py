import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, planer_amd
from planer_amd import q4
rng = np.random.default_rng(0)
def timeit(f, reps=10):
    for _ in range(3): f()
    e0 = planer_amd.hip.Event().record()
    for _ in range(reps): f()
    e1 = planer_amd.hip.Event().record()
    return e0.elapsed_ms(e1) / reps * 1e3
for (n, hw, cout) in [(32, 56, 64), (128, 56, 64), (32, 28, 128), (128, 28, 128)]:
    pts = []
    for cin in (16, 32, 64, 128, 256):
        x = q4.to_q4(planer_amd.asarray(rng.standard_normal((n, cin, hw, hw)).astype(np.float32)))
        k = planer_amd.asarray((rng.standard_normal((cout, cin, 3, 3)) * 0.05).astype(np.float32))
        u = q4.prepare_w1d_q4_weights(k)
        pts.append((3 * cin // 16, timeit(lambda: q4.ConvQ4(x, u, pads=[1] * 4, w_layout=5))))
    (c0, t0), (c1, t1) = pts[1], pts[-1]
    slope = (t1 - t0) / (c1 - c0)
    tiles = (cout // 64) * (n * hw * ((hw + 1) // 2) + 63) // 64
    ideal = tiles / 1024.0 * 32 * 64 / 2100.0     # us per chunk if all 1024 SIMDs ran MFMAs back to back at 2.1 GHz
    print("N%d %dx%d Cout %d (%d tiles): " % (n, hw, hw, cout, tiles) + " ".join("%dch %.1fus" % p for p in pts) +
          " | slope %.2f us/chunk (MFMA-bound %.2f) intercept %.1f us" % (slope, ideal, t0 - slope * c0))
