"""GPU: F(4x4,3x3) conv (w_layout 7) time per shape, for PLANER_HIP_WINO_ROWS unset / 0 / 1 (one process each),
as a captured chain of 20 launches.  Shapes: ResNet-18 layer2-4 at batch 32, YOLO-v3 52^2 / 26^2 / 13^2 at batch 1."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(32, 128, 28), (32, 256, 14), (32, 512, 7), (1, 128, 52), (1, 256, 26), (1, 512, 13), (1, 64, 104)]
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    import numpy as np, time
    import planer_amd
    from planer_amd import q4, _lib
    ctx = planer_amd.hip.context()
    out = {}
    for n, c, s in SHAPES:
        co = c if n == 32 else 2 * c
        rng = np.random.default_rng(1)
        x = q4.to_q4(planer_amd.asarray(rng.standard_normal((n, c, s, s)).astype(np.float32)))
        k = q4.prepare_winograd4_q4_weights(planer_amd.asarray((rng.standard_normal((co, c, 3, 3)) * 0.05).astype(np.float32)))
        sc = planer_amd.asarray(np.ones((1, co, 1, 1), np.float32)); sh = planer_amd.asarray(np.zeros((1, co, 1, 1), np.float32))
        f = lambda: q4.ConvQ4(x, k, None, sc, sh, None, pads=[1, 1, 1, 1], act=1, w_layout=7)
        for _ in range(3): y = f()
        ctx.synchronize()
        _lib.call("pl_capture_begin", ctx.handle)
        ys = [f() for _ in range(20)]
        g = _lib.c_void_p(); _lib.call("pl_capture_end", ctx.handle, _lib.byref(g))
        best = 1e9
        for _ in range(5):
            ctx.synchronize(); t0 = time.perf_counter()
            _lib.call("pl_graph_launch", g); ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) / 20 * 1e6)
        out["N%d C%d->%d %dx%d" % (n, c, co, s, s)] = round(best, 2)
    print(json.dumps(out))
else:
    res = {}
    for mode in ("auto", "0", "1"):
        env = dict(os.environ)
        env.pop("PLANER_HIP_WINO_ROWS", None)
        if mode != "auto": env["PLANER_HIP_WINO_ROWS"] = mode
        r = subprocess.run([sys.executable, __file__, "worker"], env=env, capture_output=True, text=True)
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else r.stderr[-400:]
    for k in res["0"]:
        print("%-26s rows=0 %7.2f us   rows=1 %7.2f us   auto %7.2f us" % (k, res["0"][k], res["1"][k], res["auto"][k]))
