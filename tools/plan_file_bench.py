"""GPU: what a host WITHOUT the Python plan compiler gets from a plan file (pl_plan_build / pl_plan_run): ResNet-18 batch 32,
R plans built from one file on R contexts (streams), used round robin with the inputs resident in HBM -- the bench's step.
Export happens first (needs planer_amd); the timed part below uses ctypes and the C ABI only.
    python tools/plan_file_bench.py [R ...]        (default 1 3 7)"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
B = int(os.environ.get("BATCH", "32"))
MODE = os.environ.get("MODE", "throughput")          # which conv algorithm table the exported pass takes (latency / throughput)
path = "/tmp/resnet18_b%d_%s.plplan" % (B, MODE)
x = np.random.default_rng(1).standard_normal((B, 3, 224, 224)).astype(np.float32)
if not os.path.exists(path):
    import planer_amd
    from planer_amd.export import export_plan
    from planer_amd.irgen import resnet18
    g, b = resnet18.build()
    net = planer_amd.from_graph(g, b)
    t0 = time.perf_counter()
    export_plan(net, x, path=path, mode=MODE)
    print("exported %s: %.1f MB in %.2f s" % (path, os.path.getsize(path) / 1e6, time.perf_counter() - t0))
    want = net(x)
else:
    want = None
blob = open(path, "rb").read()
lib = ctypes.CDLL(os.path.join(ROOT, "planer_amd", "libplaner_hip.so"))
P, I, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
lib.pl_last_error.restype = ctypes.c_char_p
for name, sig in {"pl_ctx_create": [I, ctypes.POINTER(P)], "pl_plan_build": [P, P, Z, ctypes.POINTER(P)], "pl_plan_run": [P, ctypes.POINTER(P), ctypes.POINTER(P)],
                  "pl_plan_tensor": [P, I, I, ctypes.POINTER(P), ctypes.POINTER(Z), ctypes.POINTER(I), ctypes.POINTER(I), ctypes.POINTER(I)],
                  "pl_sync": [P], "pl_alloc": [P, Z, ctypes.POINTER(P)], "pl_h2d": [P, P, P, Z], "pl_d2h": [P, P, P, Z], "pl_stream_wait": [P, P]}.items():
    getattr(lib, name).argtypes = sig


def ok(rc):
    assert rc == 0, lib.pl_last_error().decode()


buf = ctypes.create_string_buffer(blob, len(blob))
main = P()
ok(lib.pl_ctx_create(0, ctypes.byref(main)))
xd = [P(), P()]
for d in xd:                                            # two resident batches on the host's own context
    ok(lib.pl_alloc(main, x.nbytes, ctypes.byref(d)))
    ok(lib.pl_h2d(main, d, x.ctypes.data, x.nbytes))
for R in [int(a) for a in sys.argv[1:]] or [1, 3, 7]:
    ctxs, plans = [], []
    t0 = time.perf_counter()
    for r in range(R):
        c, p = P(), P()
        ok(lib.pl_ctx_create(0, ctypes.byref(c)))
        ok(lib.pl_plan_build(c, buf, len(blob), ctypes.byref(p)))
        ctxs.append(c); plans.append(p)
    build_s = time.perf_counter() - t0

    def loop(k):
        for i in range(k):
            ins = (P * 1)(xd[i & 1].value)
            ok(lib.pl_plan_run(plans[i % R], ins, None))
        for c in ctxs:
            ok(lib.pl_sync(c))
    loop(3 * R)
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        loop(150)
        best = max(best, B * 150 / (time.perf_counter() - t0))
    y = np.empty((B, 1000), np.float32)
    ptr, nb = P(), Z()
    ok(lib.pl_plan_tensor(plans[0], 1, 0, ctypes.byref(ptr), ctypes.byref(nb), None, None, None))
    ok(lib.pl_d2h(ctxs[0], y.ctypes.data, ptr, y.nbytes))
    same = None if want is None else bool(np.array_equal(y, want))
    print("plan file, %d plan(s) on %d stream(s): %.0f img/s (build %.2f s); output equals the Python host's: %s" % (R, R, best, build_s, same))
