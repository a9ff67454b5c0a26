import sys, numpy as np
sys.path.insert(0, "/root/repo")
import planer_amd as pa
from planer_amd import hip, _lib
ctx = hip.context()
for mb in (103, 256, 1024):
    n = mb * 1000 * 1000
    a = hip.empty((n // 4,), np.float32, ctx)
    b = hip.empty((n // 4,), np.float32, ctx)
    for name, fn in (("memset", lambda: _lib.call("pl_memset", ctx.handle, a.ptr, 0, n)),
                     ("d2d copy", lambda: _lib.call("pl_d2d", ctx.handle, a.ptr, b.ptr, n)),
                     ("relu (r+w)", lambda: _lib.call("pl_relu_f32", ctx.handle, b.ptr, a.ptr, n // 4))):
        for _ in range(3): fn()
        e0 = hip.Event(ctx).record()
        for _ in range(10): fn()
        e1 = hip.Event(ctx).record()
        ms = e0.elapsed_ms(e1) / 10
        print("%4d MB %-10s %.1f us  %.2f TB/s written" % (mb, name, ms * 1e3, n / ms / 1e9))
