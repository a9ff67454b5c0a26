import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["PLANER_HIP_SMALLCIN_VALU"] = "1"
import planer_amd as pa
from oracle import planer_np as onp
rng = np.random.default_rng(78)
for (n, c, h, w, co, pad) in [(1, 1, 6, 302, 8, 0), (1, 1, 6, 14, 70, 0), (1, 1, 6, 62, 8, 0), (1, 1, 6, 30, 8, 0), (1, 2, 6, 302, 8, 0), (1, 1, 6, 304, 8, 1)]:
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    k = (rng.standard_normal((co, c, 3, 3)) * 0.1).astype(np.float32)
    y = pa.Conv2d(pa.asarray(x), pa.asarray(k), None, pads=[pad] * 4).get()
    ref = np.ascontiguousarray(onp.conv2d(x, k, None, pads=[pad] * 4))
    d = np.abs(y - ref)
    bad = np.argwhere(d > 1e-3)
    print((n, c, h, w, co, pad), pa.hip.context().last_conv_plan(), "max err", d.max(), "nbad", len(bad))
    for b in bad[:6]:
        nn, ch, r, cc = b
        # which single-tap omission / substitution explains it?
        terms = [(k[ch, 0, i, j] * x[0, 0, r + i - pad, cc + j - pad]) if 0 <= r + i - pad < h and 0 <= cc + j - pad < w else 0.0 for i in range(3) for j in range(3)]
        print("   at", tuple(b), "got %.5f ref %.5f diff %.5f" % (y[tuple(b)], ref[tuple(b)], y[tuple(b)] - ref[tuple(b)]), "terms", np.round(terms, 4))
