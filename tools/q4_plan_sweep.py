"""GPU: one Q4 direct conv shape, every tile configuration x split-K (forced plans), timed as a captured chain of 20
launches, next to what the autotuner picks.  usage: q4_plan_sweep.py N Cin H W Cout k stride"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import planer_amd
from planer_amd import q4, _lib

shapes = [tuple(int(v) for v in sys.argv[1:8])] if len(sys.argv) >= 8 else [
    (1, 256, 26, 26, 512, 3, 2), (1, 512, 13, 13, 1024, 3, 1), (1, 1024, 13, 13, 512, 1, 1), (1, 256, 26, 26, 512, 3, 1),
    (1, 512, 26, 26, 256, 1, 1), (1, 128, 52, 52, 256, 3, 1)]
ctx = planer_amd.hip.context()
lib = _lib.load()
names = []
for c in range(lib.pl_conv2d_num_configs()):
    buf = ctypes.create_string_buffer(32); lib.pl_conv2d_config_name(c, buf, 32); names.append(buf.value.decode())


def chain_us(f, n=20):
    for _ in range(2): f()
    ctx.synchronize()
    _lib.call("pl_capture_begin", ctx.handle)
    keep = [f() for _ in range(n)]
    g = _lib.c_void_p(); _lib.call("pl_capture_end", ctx.handle, _lib.byref(g))
    best = 1e9
    for _ in range(4):
        ctx.synchronize(); t0 = time.perf_counter()
        _lib.call("pl_graph_launch", g); ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    _lib.call("pl_graph_destroy", g)
    return best


for n, cin, h, w, cout, k, st in shapes:
    rng = np.random.default_rng(1)
    x = q4.to_q4(planer_amd.asarray(rng.standard_normal((n, cin, h, w)).astype(np.float32)))
    K = q4.prepare_q4_weights(planer_amd.asarray((rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32)))
    sc = planer_amd.asarray(np.ones((1, cout, 1, 1), np.float32)); sh = planer_amd.asarray(np.zeros((1, cout, 1, 1), np.float32))
    f = lambda: q4.ConvQ4(x, K, None, sc, sh, None, pads=[k // 2] * 4, strides=[st, st], act=2, alpha=0.1, w_layout=2)
    ctx.set_conv_config(-1, 0)
    auto = chain_us(f)
    auto_plan = ctx.last_conv_plan()
    rows = []
    for c, nm in enumerate(names):
        if not nm.startswith("q"): continue
        for s in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            try:
                ctx.set_conv_plan(c, 0, s)
                f(); plan = ctx.last_conv_plan()
                if s > 1 and "split=1 " in plan: break
                rows.append((chain_us(f), nm, s, plan))
            except Exception as e:
                break
    ctx.set_conv_config(-1, 0)
    rows.sort()
    print("N%d C%d %dx%d -> %d k%d s%d: autotuned %.2f us [%s]" % (n, cin, h, w, cout, k, st, auto, auto_plan))
    for t, nm, s, plan in rows[:5]:
        print("    %-12s split %2d: %.2f us  [%s]" % (nm, s, t, plan))
