import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import planer_amd as pa
from planer_amd import hip
ctx = hip.context()
rng = np.random.default_rng(0)
x = rng.standard_normal((8, 3, 224, 224)).astype(np.float32)
k = (rng.standard_normal((64, 3, 3, 3)) * 0.1).astype(np.float32)
b = rng.standard_normal(64).astype(np.float32)
dx, dk, db = pa.asarray(x), pa.asarray(k), pa.asarray(b)
run = lambda: pa.Conv2d(dx, dk, db, pads=[1] * 4)
for _ in range(3): run()
best = 1e9
for _ in range(4):
    e0 = hip.Event(ctx).record()
    for _ in range(10): run()
    e1 = hip.Event(ctx).record()
    best = min(best, e0.elapsed_ms(e1) / 10)
print("%s: %.1f us [%s]" % (os.environ.get("TAG"), best * 1e3, ctx.last_conv_plan()))
