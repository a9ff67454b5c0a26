#!/usr/bin/env python
"""Host -> device routes for one ResNet batch (32 x 3 x 224 x 224 floats = 19.3 MB), GB/s each: the runtime's own pageable
hipMemcpyAsync (pl_h2d), the staged route (copy threads -> pinned ring -> DMA on the consumer's stream; PLANER_HIP_COPY_STREAMS=1:
on a copy stream) per worker count / chunk size, and DMA straight out of pinned memory.  Run on the GPU box.

    python tools/h2d_probe.py [threads ...]
"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def child():
    import numpy as np
    import planer_amd
    from planer_amd import hip
    ctx = hip.context()
    n = 32 * 3 * 224 * 224
    x = np.random.default_rng(0).standard_normal(n).astype(np.float32)
    d = hip.empty((n,), np.float32, ctx)
    which = os.environ["H2D_PROBE"]
    if which == "pinned":
        p = hip.pinned_empty((n,), np.float32)
        p[...] = x
        x = p
    if which == "memcpy":                     # host memcpy alone through the copy pool (pinned destination)
        p = hip.pinned_empty((n,), np.float32)
        for _ in range(3):
            p[...] = x
        t0 = time.perf_counter()
        for _ in range(20):
            np.copyto(p, x)
        dt = (time.perf_counter() - t0) / 20
        print("%-34s %7.1f GB/s  %.3f ms (numpy copyto, one thread)" % (which, x.nbytes / dt / 1e9, dt * 1e3))
        return
    tag = "%s threads=%s chunk=%s" % (which, os.environ.get("PLANER_HIP_COPY_THREADS", "dflt"), os.environ.get("PLANER_HIP_COPY_CHUNK_KB", "4096"))

    def timed(label, fn, reps=30):
        for _ in range(5):
            fn()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%-34s %7.1f GB/s  %.3f ms per batch (%s)" % (tag, x.nbytes / dt / 1e9, dt * 1e3, label))
    if which in ("plain", "pinned"):
        timed("pl_h2d: the runtime's own route, host and stream wait", lambda: d.set(x))
    if which != "plain":
        def one():
            d.set_staged(x)
            ctx.synchronize()
        timed("pl_h2d_staged + sync each", one)
        timed("pl_h2d_staged, host not waiting", lambda: d.set_staged(x))


if __name__ == "__main__":
    if os.environ.get("H2D_PROBE"):
        child()
        sys.exit(0)
    threads = sys.argv[1:] or ["0", "3", "7", "15", "31"]
    runs = [("plain", {}), ("memcpy", {}), ("pinned", {})]
    runs += [("staged", {"PLANER_HIP_COPY_THREADS": t}) for t in threads]
    runs += [("staged", {"PLANER_HIP_COPY_THREADS": "7", "PLANER_HIP_COPY_CHUNK_KB": c}) for c in ("1024", "2048", "8192", "32768")]
    for which, env in runs:
        e = dict(os.environ, H2D_PROBE=which, **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True)
        sys.stdout.write(r.stdout if r.returncode == 0 else "%s FAILED: %s\n" % (which, r.stderr[-400:]))
