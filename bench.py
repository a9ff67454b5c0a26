#!/usr/bin/env python
"""Headline benchmark: images/sec of the ResNet-18 fp32 forward pass on
MI355X, next to the numpy-CPU baseline, with the roofline of the dominant
kernel and a parity check of exactly what was timed.

    python bench.py --gpus N --steps K --warmup W

N>1 runs one rank per GPU: `python -m planer_amd.launch --nproc N bench.py --gpus N ...`
(the product's own torch-free spawner) or, as the driver does, `python -m
torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` -- either is
only a process spawner: ranks exchange the RCCL id through a file and the
weight blob through ONE RCCL broadcast over xGMI; the forward pass itself has
no collective (batch shards are independent), so scaling is weak: 32 images
per GPU.  A step = one captured forward pass over one resident batch.
Rank 0 prints one JSON line.

What the line's numbers mean:
  value               images/sec of the timed, pipelined run (inputs resident in HBM)
  parity_rel_err      max|logits - oracle| / max|oracle| of the plan that was just timed
  config.algos        which kernel family + launch plan every conv layer ran
  roofline.frac       EXECUTED MFMA FLOPs of the dominant kernel / its device time / 157.3 TFLOP/s
                      (= matrix-core utilisation; Winograd kernels execute 1.5-4x fewer FLOPs
                      than the direct algorithm, padding included)
  roofline.effective_frac  the same with ALGORITHMIC (direct-conv) FLOPs
  roofline_hbm        the HBM-bound single-kernel layers: algorithmic bytes / time / 8 TB/s
  cpu_baseline        the numpy oracle on the host cores of this box (all cores at batch 32 and
                      batch 1, and OPENBLAS_NUM_THREADS=1 per-core), with library versions
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0
PER_GPU_BATCH = 32                 # BASELINE.json configs[2] / configs[3]: 32 images per GPU
PROF_REPEAT = 10                   # launches of a step between its two stream markers in the per-layer passes
PROFILE_TAG = "r06"


def cdiv(a, b):
    return -(-a // b)


def conv_table(g, shapes):
    """Every conv / dense layer of the graph: algorithmic FLOPs (2*MAC) and geometry, by layer name."""
    kinds = {n: (k, p) for n, k, p in g["layers"]}
    out = {}
    for src, names, dst in g["flow"]:
        kind, para = kinds[names[0]]
        if kind == "conv":
            n, cout, ho, wo = shapes[dst]
            _, cin_g, kh, kw = shapes[src[1]]
            _, cin, h, w = shapes[src[0]]
            out[names[0]] = {"flops": 2.0 * n * cout * ho * wo * cin_g * kh * kw, "k": (kh, kw), "n": n, "cin": cin,
                             "cout": cout, "h": h, "w": w, "ho": ho, "wo": wo, "group": int(para.get("group", 1)),
                             "cls": "conv%dx%d" % (kh, kw)}
        elif kind == "dense":
            n, cout = shapes[dst]
            out[names[0]] = {"flops": 2.0 * n * cout * shapes[src[0]][1], "k": (1, 1), "n": n, "cin": shapes[src[0]][1],
                             "cout": cout, "h": 1, "w": 1, "ho": 1, "wo": 1, "group": 1, "cls": "dense"}
    return out


def executed_flops(rec):
    """MFMA FLOPs the chosen kernel really issues (tile, K-chunk and Winograd-tile padding included): the library
    reports the GEMM it executed -- (groups, rows, columns, K) rounded up to whole tiles and chunks
    (pl_conv2d_last_extents) -- for every conv / dense step of the plan."""
    ext = rec.get("extents") if rec else None
    if not ext or min(ext) <= 0:
        return None
    g, rows, cols, k = ext
    return 2.0 * g * rows * cols * k


def split_step(name):
    """Plan step name -> (layer of the user's graph, stage): "l20b_conv+@chain" -> ("l20b_conv", "chain"); a pair of
    sibling convs in one launch, "l20a_conv+&l20d_conv+", keeps both names: "l20a_conv&l20d_conv"."""
    base, _, stage = name.partition("@")
    return "&".join(part.rstrip("+") for part in base.split("&")), stage


def conv_of(convs, base):
    """Geometry / FLOPs of a plan step's conv(s): a paired step sums its two convs (class of the first)."""
    if "&" not in base:
        return convs.get(base)
    parts = [convs.get(b) for b in base.split("&")]
    if any(c is None for c in parts):
        return None
    return dict(parts[0], flops=sum(c["flops"] for c in parts))


def transform_bytes(prog, convs):
    """Algorithmic HBM bytes of the Winograd F(4x4,3x3) transform steps of a fused program (each tensor read or
    written once): x + V for an input transform, M (+ residual) + y for an output transform, M (+ residual)
    (+ y when something else reads it) + V for a chained one.  V / M hold 36 values per 4x4 tile and channel."""
    out = {}
    for src, names, dst in prog.flow:
        name = names[0]
        obj = prog.objs[name]
        base, stage = split_step(name)
        c = conv_of(convs, base)
        if c is None or stage not in ("in", "out", "chain"):
            continue
        tiles = c["n"] * cdiv(c["h"], 4) * cdiv(c["w"], 4)
        act_in, act_out = 4.0 * c["n"] * c["cin"] * c["h"] * c["w"], 4.0 * c["n"] * c["cout"] * c["ho"] * c["wo"]
        v_in, m_out = 4.0 * 36 * c["cin"] * tiles, 4.0 * 36 * c["cout"] * tiles
        if obj.name.startswith("wino43"):            # mixed tiles: 121 frequency groups x (H / 7) (W / 7) tiles per image
            cols = c["n"] * (c["h"] // 7) * (c["w"] // 7)
            v_in, m_out = 4.0 * 121 * c["cin"] * cols, 4.0 * 121 * c["cout"] * cols
        if stage == "in":
            out[name] = act_in + v_in
        else:
            res = act_out if (len(src) > 4 and src[4] != "None") else 0.0
            if stage == "out":
                out[name] = m_out + res + act_out
            else:
                out[name] = m_out + res + (act_out if obj.para().get("keep_y", True) else 0.0) + m_out
    return out


def sample_sclk(ctx, step, sync, samples=5):
    """Shader clock (MHz) of this GPU while `step` runs back to back (untimed): the level sysfs marks active in
    /sys/bus/pci/devices/<bus id>/pp_dpm_sclk, sampled between bursts, the median.  None when sysfs is not there.
    (No management CLI is forked: a fork with captured graphs in flight crashes rocprofv3's interception.)"""
    try:
        path = "/sys/bus/pci/devices/%s/pp_dpm_sclk" % ctx.pci_bus_id()
        seen = []
        for _ in range(samples):
            for _ in range(10):                        # the settle loop's burst: ~7 ms of queued work
                step()
            with open(path) as f:                      # read while the queue is still full
                m = re.search(r":\s*(\d+)\s*Mhz\s*\*", f.read(), re.I)
            sync()
            if m:
                seen.append(int(m.group(1)))
        return sorted(seen)[len(seen) // 2] if seen else None
    except Exception:
        return None


def oracle_net(g, b):
    from oracle import planer_np as onp
    net = onp.OracleNet()
    net.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    net.load_weights(b)
    return net


def _median_rate(net, x, iters):
    net(x.copy())
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        net(x.copy())
        ts.append(time.perf_counter() - t0)
    return x.shape[0] / float(np.median(ts)), x.shape[0] / min(ts)


def cpu_worker(batch, iters):
    """`bench.py --cpu-worker B I`: the oracle alone in a fresh process (so OPENBLAS_NUM_THREADS set
    by the parent takes effect); prints images/sec."""
    from planer_amd.irgen import resnet18
    g, b = resnet18.build()
    x = np.random.default_rng(1).standard_normal((batch, 3, 224, 224)).astype(np.float32)
    med, best = _median_rate(oracle_net(g, b), x, iters)
    print(json.dumps({"images_per_sec": med, "best": best}))


def cpu_baseline(g, b, x, iters):
    """The numpy restatement of the reference (oracle/planer_np.py: im2col + OpenBLAS sgemm) timed
    on the host cores of this box on a bounded sample.  Returns the
    report and the oracle logits of `x` (for the parity check)."""
    net = oracle_net(g, b)
    logits = net(x.copy())
    med, best = _median_rate(net, x, iters)
    n1, _ = _median_rate(net, x[:1], 3)
    threads, blas = len(os.sched_getaffinity(0)), "unknown"
    try:
        import threadpoolctl
        info = [i for i in threadpoolctl.threadpool_info() if i.get("user_api") == "blas"]
        if info:
            threads = int(info[0]["num_threads"])
            blas = "%s %s (%s)" % (info[0].get("internal_api"), info[0].get("version"), info[0].get("threading_layer"))
    except Exception:
        pass
    cpu = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu = [ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")][0]
    except Exception:
        pass
    per_core = None
    try:            # one BLAS thread, fresh process: the per-core figure BASELINE.md asks for
        env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", "4", "2"], env=env,
                           capture_output=True, text=True, timeout=300)
        per_core = json.loads(r.stdout.strip().splitlines()[-1])["images_per_sec"]
    except Exception:
        pass
    # the BLAS thread count that serves this path best on this box (the all-cores default above is the reference's own
    # behaviour: its 9-worker im2col pool x every OpenBLAS thread oversubscribes a many-core host): 1 warm-up + 2 forwards
    # of the same batch per setting, in-process through threadpoolctl
    sweep, best_threads = {}, None
    try:
        import threadpoolctl
        for t in (1, 8, 16, 32, 64):
            if t > (os.cpu_count() or 1):
                continue
            with threadpoolctl.threadpool_limits(limits=t, user_api="blas"):
                sweep[str(t)] = round(_median_rate(net, x, 2)[0], 2)
        if sweep:
            k = max(sweep, key=lambda q: sweep[q])
            best_threads = {"threads": int(k), "images_per_sec": sweep[k], "sweep": sweep,
                            "note": "OpenBLAS threads limited in-process (threadpoolctl), batch %d, median of 2 forwards "
                                    "after 1 warm-up per setting" % x.shape[0]}
    except Exception:
        pass
    rep = {"value": med, "unit": "images/sec", "cores": threads, "kind": "port", "best_threads": best_threads,
           "sample": "%d forwards of ResNet-18 batch %d after 1 warm-up, median (best %.1f img/s); batch 1: 3 forwards; "
                     "per_core: 2 forwards of batch 4 with OPENBLAS_NUM_THREADS=1 in a fresh process"
                     % (iters, x.shape[0], best),
           "batch1_images_per_sec": round(n1, 2), "per_core_images_per_sec": None if per_core is None else round(per_core, 3),
           "logical_cpus": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0)), "blas": blas,
           "numpy": np.__version__, "cpu_model": cpu}
    return rep, logits


def _algorithmic_bytes(g, shapes):
    """Unfused HBM bytes of one forward: every flow step reads its activation inputs and weights once and writes its
    outputs once (4 bytes per element)."""
    total = 0.0
    kinds = {n: k for n, k, _ in g["layers"]}
    for src, names, dst in g["flow"]:
        if all(kinds.get(n) in ("return", "flatten", "identity", "reshape", "squeeze", "unsqueeze") for n in names):
            continue                               # views: nothing moves
        for k in (src if isinstance(src, list) else [src]) + (dst if isinstance(dst, list) else [dst]):
            shp = shapes.get(k)
            if shp is not None and len(shp):
                total += 4.0 * float(np.prod(shp))
    return total


def secondary_workloads(planer_amd, ctx, budget_s=3.0):
    """BASELINE configs 1, 2 and 5 on the same box, after the headline: one image (batch) at a time through `net(x_dev)`
    on the latency plan (one captured graph).  Per workload: `ms_per_step` = device time of one replay (HIP events around K
    back-to-back replays, K bounded by `budget_s` / 3 per workload), `latency_ms` = host wall time of net(x) + sync (median),
    the roofline fraction that bounds it, and parity of that plan's output against the oracle."""
    from planer_amd.irgen import customnet, yolov3
    from planer_amd.irgen.builder import GraphBuilder

    def conv2_build():
        rng = np.random.default_rng(0)
        gb = GraphBuilder(["x"])
        gb.init("K", (rng.standard_normal((64, 3, 3, 3)) * 0.1).astype(np.float32))
        gb.init("B", rng.standard_normal(64).astype(np.float32))
        gb.op("conv", ["x", "K", "B"], "y", name="conv", group=1, strides=[1, 1], dilations=[1, 1], pads=[1, 1, 1, 1])
        return gb.finish(["y"])
    out = {}
    jobs = (("config2", conv2_build, (8, 3, 224, 224), 2, "hbm"),
            ("yolov3_b1", yolov3.build, (1, 3, 416, 416), 1, "mfma"),
            ("customnet_b1", customnet.build, (1, 3, 64, 64), 1, "hbm"))
    for key, build, shape, check, bound in jobs:
        try:
            g, blob = build()
            xh = np.random.default_rng(7).standard_normal(shape).astype(np.float32)
            net = planer_amd.Net(ctx)
            net.load_json(g["input"], g["inits"], g["layers"], g["flow"])
            net.load_weights(blob)
            net.streams = "1x1"
            x = planer_amd.asarray(xh, ctx=ctx)
            t0 = time.perf_counter()
            plan = net.compile(x)
            ctx.synchronize()
            compile_s = time.perf_counter() - t0
            y = net(x)
            got = [t.get() for t in (y if isinstance(y, tuple) else (y,))]
            ref = oracle_net(g, blob)(xh[:check].copy())
            ref = list(ref) if isinstance(ref, tuple) else [ref]
            if shape[0] == 1 and got[0].ndim == np.asarray(ref[0]).ndim - 1:      # net.py:101 drops a batch of one
                got = [a[None] for a in got]
            parity = max(float(np.abs(a[:check].astype(np.float64) - np.asarray(r)).max() / max(np.abs(r).max(), 1e-30))
                         for a, r in zip(got, ref))
            # device time per replay: K graph launches between two markers, enqueued BEHIND a few ms of memsets -- one
            # hipGraphLaunch costs the host 10-40 us, more than a small graph runs, and a GPU that waits for the host
            # would bill the wait to the workload
            e0, e1 = planer_amd.hip.Event(ctx), planer_amd.hip.Event(ctx)
            for _ in range(5):
                plan.launch()
            ctx.synchronize()
            e0.record(); plan.launch(); e1.record()
            one = max(e0.elapsed_ms(e1), 1e-3)
            blocker = planer_amd.hip.empty((256 << 20,), np.float32, ctx)          # 1 GiB
            plan.max_in_flight = 0                 # no host-side ring waits inside the timed burst
            k = int(max(8, min(48, 2.0 / one)))
            samples = []
            t_end = time.perf_counter() + budget_s / 3 / 2
            while len(samples) < 3 or (time.perf_counter() < t_end and len(samples) < 30):
                for _ in range(16):
                    planer_amd._lib.call("pl_memset", ctx.handle, blocker.ptr, 0, blocker.nbytes)
                e0.record()
                for _ in range(k):
                    plan.launch()
                e1.record()
                samples.append(e0.elapsed_ms(e1) / k)
            ms = float(np.median(samples))
            plan.max_in_flight = 8
            del blocker
            lat = []
            t_end = time.perf_counter() + budget_s / 3 / 2
            while len(lat) < 10 or (time.perf_counter() < t_end and len(lat) < 500):
                t0 = time.perf_counter()
                net(x)
                ctx.synchronize()
                lat.append(time.perf_counter() - t0)
            shapes = {kk: a.shape for kk, a in zip(net.inits, net.weights)}
            shapes[g["input"][0]] = x.shape
            net._interpret(net._program, [x.copy()], shapes=shapes)
            convs = conv_table(g, shapes)
            alg = sum(c["flops"] for c in convs.values())
            exe = sum(executed_flops(a) or 0.0 for a in plan.algos)
            row = {"workload": {"config2": "BASELINE configs[1]: Conv2d 3->64 k3 s1 p1 on (8,3,224,224)",
                                "yolov3_b1": "BASELINE configs[4]: YOLO-v3 @416, batch 1",
                                "customnet_b1": "BASELINE configs[0]: README CustomNet on (1,3,64,64)"}[key],
                   "ms_per_step": round(ms, 5), "images_per_sec": round(shape[0] / (ms * 1e-3), 1),
                   "latency_ms": round(float(np.median(lat)) * 1e3, 4), "replays_timed": k * len(samples),
                   "parity_rel_err": parity, "parity_checked_images": check, "compile_s": round(compile_s, 2),
                   "tune_source": net.tune_source()}
            if bound == "hbm":
                nbytes = _algorithmic_bytes(g, shapes)
                row["roofline"] = {"bound": "hbm", "bytes": nbytes, "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1),
                                   "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                   "note": "algorithmic bytes: every tensor of the unfused flow read once and written once"}
            else:
                row["roofline"] = {"bound": "mfma", "executed_flops": exe, "algorithmic_flops": alg,
                                   "achieved": round(exe / (ms * 1e-3) / 1e12, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                                   "unit": "TFLOP/s", "frac": round(exe / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                   "effective_frac": round(alg / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
            if not parity <= 1e-4:
                row["parity_failure"] = True
            out[key] = row
            del plan, net, x
        except Exception as e:                 # noqa: BLE001 -- a secondary line must never take the headline down
            out[key] = {"error": repr(e)[:300]}
    return out


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(int(sys.argv[2]), int(sys.argv[3]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--detail", action="store_true", help="also print the per-layer table to stderr")
    ap.add_argument("--workload", default="resnet18", choices=["resnet18", "yolov3", "conv2"],
                    help="resnet18 = the headline (BASELINE configs[2]); yolov3 = config 5 at batch 1; "
                         "conv2 = config 2's single Conv2d 3->64 on (8,3,224,224)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive net(x_host) leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads (BASELINE configs 1, 2, 5) after the headline")
    ap.add_argument("--no-sclk", action="store_true", help="skip the rocm-smi shader-clock samples (profiling runs)")
    ap.add_argument("--per-layer-csv", help="write the per-layer table (HIP events) to this file")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of K steps each, back to back; `value` is the median repeat (config.repeat_values)")
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="run the step loop untimed for this long before the W warm-up steps, so the timed region sees "
                         "steady-state clocks (DVFS ramps over ~100 ms; the K timed steps last ~15-35 ms)")
    args = ap.parse_args()

    import planer_amd
    from planer_amd import dist
    from planer_amd.irgen import resnet18, yolov3
    from planer_amd.irgen.builder import GraphBuilder

    rank, world, _ = dist.env_world()
    if world != args.gpus:
        if args.gpus != 1 and world == 1:
            sys.exit("bench.py --gpus %d must be launched with one process per GPU "
                     "(python -m planer_amd.launch --nproc %d bench.py --gpus %d ...; python -m torch.distributed.run "
                     "--nproc-per-node %d works as well)" % (args.gpus, args.gpus, args.gpus, args.gpus))
    ctx = planer_amd.hip.context()
    # RCCL or nothing: the same-node file transport is only used when asked for by name, so a
    # fallback run can never be mistaken for the RCCL path
    comm = dist.init(ctx, fallback=False)

    # ---- model: graph on every rank, weights from rank 0 by RCCL broadcast -------
    if args.workload == "yolov3":
        build, in_shape, args.batch = yolov3.build, (3, 416, 416), (args.batch if args.batch != PER_GPU_BATCH else 1)
    elif args.workload == "conv2":
        def build():
            rng = np.random.default_rng(0)
            gb = GraphBuilder(["x"])
            gb.init("K", (rng.standard_normal((64, 3, 3, 3)) * 0.1).astype(np.float32))
            gb.init("B", rng.standard_normal(64).astype(np.float32))
            gb.op("conv", ["x", "K", "B"], "y", name="conv", group=1, strides=[1, 1], dilations=[1, 1], pads=[1, 1, 1, 1])
            return gb.finish(["y"])
        in_shape, args.batch = (3, 224, 224), (args.batch if args.batch != PER_GPU_BATCH else 8)
    else:
        build, in_shape = resnet18.build, (3, 224, 224)
    # weights come from rank 0 by ONE RCCL broadcast; only the (explicit) file transport
    # regenerates the seeded blob on every rank
    g, blob = build() if (rank == 0 or not comm.device_transport) else (build()[0], None)
    net = planer_amd.Net(ctx)
    net.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    comm.load_weights(net, blob)                # rank 0 uploads; ONE ncclBroadcast; timed between barriers (comm.bcast_ms)
    ctx.synchronize()
    bcast_ms = comm.bcast_ms or 0.0
    rccl_ranks = comm.transport_ranks()
    if world > 1 and comm.device_transport and rccl_ranks != world:
        sys.exit("RCCL counts %d ranks in the communicator, the launcher started %d" % (rccl_ranks, world))

    # ---- data: this rank's shard of the global synthetic batch, resident in HBM ---
    n = args.batch
    global_batch = n * world
    lo, hi = dist.shard_range(global_batch, world, rank)
    xs_host = [np.random.default_rng(1 + 1000 * i + rank).standard_normal((hi - lo,) + in_shape).astype(np.float32)
               for i in range(2)]
    xs = [planer_amd.asarray(a, ctx=ctx) for a in xs_host]
    # fuse + pick algorithms + tune + warm the pool + capture the hipGraph(s)
    t_compile = time.perf_counter()
    plan = net.compile(xs[0], mode="throughput")
    ctx.synchronize()
    compile_s = time.perf_counter() - t_compile
    stream_probe = getattr(plan, "stream_probe", None)
    tune_misses = {"launch_plans": ctx.tune_stats()[1] + sum(c.tune_stats()[1] for c in net._side),
                   "conv_algorithms": net.algo_misses, "stream_plans": net.stream_misses}
    tune_src, wino_chains = net.tune_source(), net.wino_chains      # where the TIMED plan's kernel choices came from
    ctx.save_tune_cache()                       # no-op unless PLANER_HIP_TUNE_CACHE is set
    net.save_algo_cache()
    state = {"i": 0}

    def step():
        plan.feed([xs[state["i"] & 1]])        # rotate two distinct resident batches
        plan.launch(join=False)
        state["i"] += 1

    def sync():
        plan.join()                            # side streams -> main stream
        ctx.synchronize()

    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:      # untimed: clocks and caches reach steady state
        for _ in range(10):
            step()
        sync()
    # `repeats` timed regions of exactly K steps each (barrier + device sync on both sides, MAX over ranks);
    # the line reports the MEDIAN repeat, with the spread next to it (median and best)
    spans, own = dist.timed_repeats(comm, step, sync, args.steps, args.warmup, args.repeats)
    # A disturbed invocation (seen once in ~20 on the gpurun boxes: a first region at half rate, recovering over the next four while the
    # shader clock comes back) is measured longer, not reported: when the regions spread by more than 5 %, as many regions again are
    # timed and the median is taken over all of them (every rank sees the same MAX-over-ranks spans, so all ranks decide alike)
    extended = False
    if args.repeats > 1 and max(spans) > 1.05 * min(spans):
        s2, o2 = dist.timed_repeats(comm, step, sync, args.steps, 0, args.repeats)
        spans, own, extended = spans + s2, own + o2, True
    order = sorted(range(len(spans)), key=lambda i: spans[i])
    mid = order[len(order) // 2]
    elapsed = spans[mid]
    ms_per_step = elapsed / args.steps * 1e3
    value = global_batch * args.steps / elapsed
    repeat_values = {"repeats": len(spans), "min": round(global_batch * args.steps / max(spans), 1),
                     "median": round(value, 1), "max": round(global_batch * args.steps / min(spans), 1),
                     "all": [round(global_batch * args.steps / t, 1) for t in spans], "extended": extended}
    # every rank's own rate over the median repeat (no barrier wait inside): min / max over ranks
    my_rate = (hi - lo) * args.steps / own[mid]
    rank_rates = {"min": round(comm.min_over_ranks(my_rate), 1), "max": round(comm.max_over_ranks(my_rate), 1)}
    # clock state (untimed): the shader clock sysfs reports while the same loop keeps running
    sclk_mhz = sample_sclk(ctx, step, sync) if rank == 0 and not args.no_sclk else None

    # ---- parity of what was just timed: every replica of the plan on batch 0 vs the oracle ----
    sync()
    got = []
    for _ in range(len(getattr(plan, "replicas", [0]))):
        plan.feed([xs[0]])
        plan.launch(join=False)
        sync()
        o = plan.outputs
        got.append([t.get() for t in (o if isinstance(o, tuple) else (o,))])
    for other in got[1:]:
        for a, c in zip(got[0], other):
            assert np.array_equal(a, c), "replicas of one plan disagree"
    assert all(np.isfinite(t).all() for t in got[0])

    if rank != 0:
        return

    # rank 0 alone from here (the other ranks are done: no collective follows): the CPU baseline and the
    # full-batch parity check run at every world size, on rank 0's shard
    cpu_rep, want = None, None
    if args.workload == "resnet18" and not args.no_cpu_baseline:
        cpu_rep, want = cpu_baseline(g, blob, xs_host[0], args.cpu_iters)
        want, checked = [want], xs_host[0].shape[0]
    else:                                      # bounded: the first images of the batch only
        checked = min(2, xs_host[0].shape[0])
        w = oracle_net(g, blob)(xs_host[0][:checked].copy())
        want = list(w) if isinstance(w, tuple) else [w]
    parity = 0.0
    for a, r in zip(got[0], want):
        r = np.asarray(r)
        parity = max(parity, float(np.abs(a[:r.shape[0]].astype(np.float64) - r).max() / max(np.abs(r).max(), 1e-30)))
    if not parity <= 1e-4:
        sys.exit("PARITY FAILURE: the timed plan differs from the oracle by %.3e of max|ref| (> 1e-4)" % parity)

    # ---- per-layer device time: HIP events around every layer of the same fused program,
    #      launched eagerly on the main stream (no trial launches: algorithms and launch plans are
    #      cached by now) ----------------------------------------------------------------------
    shapes = {k: a.shape for k, a in zip(net.inits, net.weights)}
    shapes[g["input"][0]] = xs[0].shape
    net._interpret(net._program, [xs[0].copy()], shapes=shapes)
    convs = conv_table(g, shapes)
    with net.picking("throughput"):           # the TIMED plan's program: throughput plans take the pipeline-judged algorithm picks
        prog, _ = net._fuse(shapes)
    tbytes = transform_bytes(prog, convs)
    per_layer = {}
    prof_steps = min(max(args.steps, 5), 20)
    # Each profiled pass is enqueued BEHIND ~2 ms of memsets: the Python interpreter needs about as long to launch a step
    # as the GPU needs to run it, and a GPU that waits for the host would bill the wait to the layer.  With the queue
    # pre-filled the events bracket back-to-back kernels (what the captured graph replays).
    # Every step is launched PROF_REPEAT times between its two markers: a marker alone costs ~5 us of stream time (the
    # `flatten` / `return` steps, which launch nothing, show it), more than the small kernels it is meant to time.
    blocker = planer_amd.hip.empty((256 << 20,), np.float32, ctx)          # 1 GiB
    prof_steps = max(3, prof_steps // 4)
    for it in range(prof_steps + 2):
        xin = xs[it & 1].copy()
        for _ in range(8):
            planer_amd._lib.call("pl_memset", ctx.handle, blocker.ptr, 0, blocker.nbytes)
        net._interpret(prog, [xin], profile=True, repeat=PROF_REPEAT)
        if it >= 2:
            for name, kind, ms in net.last_events:
                per_layer.setdefault((name, kind), []).append(ms)
    # ... and the same program with every step launched ONCE between its markers, in forward order: a kernel then finds its
    # operands where a real forward leaves them (the fused Winograd kernel streams 2.4 MB of filter fragments per layer2 conv:
    # 43.7 us when the previous launch has just pulled them through L2, 52-54 us inside a forward and in the rocprofv3 trace).
    # The marker's own stream time is read off the steps that launch nothing (flatten / return) and taken off every step.
    once = {}
    for it in range(prof_steps + 2):
        xin = xs[it & 1].copy()
        for _ in range(8):
            planer_amd._lib.call("pl_memset", ctx.handle, blocker.ptr, 0, blocker.nbytes)
        net._interpret(prog, [xin], profile=True, repeat=1)
        if it >= 2:
            for name, kind, ms in net.last_events:
                once.setdefault((name, kind), []).append(ms)
    null_ms = [float(np.median(v)) for (name, kind), v in once.items() if kind in ("flatten", "return", "identity")]
    marker_ms = float(np.median(null_ms)) if null_ms else 0.0
    in_forward_ms = {name: max(float(np.median(v)) - marker_ms, 0.0) for (name, kind), v in once.items()}
    del blocker
    algos = {split_step(a["layer"])[0]: a for a in plan.algos}
    rows, classes, families = [], {}, {}
    for (name, kind), v in per_layer.items():
        ms = float(np.median(v))                 # median: one pool-growth hiccup must not skew a layer
        base, stage = split_step(name)
        c, rec = conv_of(convs, base), algos.get(base)
        mfma_step = c is not None and stage in ("", "gemm")          # the step of a conv that runs its GEMM(s)
        alg = c["flops"] if mfma_step else 0.0
        exe = executed_flops(rec) if mfma_step else None
        cls = c["cls"] if c else kind
        fam = (rec["algo"] if (rec and c) else kind)
        rows.append({"layer": name, "class": cls, "kernel": fam, "stage": stage, "kind": kind,
                     "plan": rec["plan"] if (rec and mfma_step) else "", "ms": ms,
                     "algorithmic_flops": alg, "executed_flops": exe, "hbm_bytes": tbytes.get(name)})
        for key, table in ((cls, classes), (fam, families)):
            t = table.setdefault(key, {"ms": 0.0, "flops": 0.0, "executed": 0.0, "launches": 0, "steps": 0})
            t["ms"] += ms
            t["flops"] += alg
            t["executed"] += exe or 0.0
            t["launches"] += 1 if (mfma_step or not c) else 0            # convs (not their transform steps)
            t["steps"] += 1
        if args.detail:
            print("%-18s %-10s %8.3f ms  %7.2f TFLOP/s algorithmic  %7.2f executed  %s"
                  % (name, cls, ms, alg / ms / 1e9 if ms else 0, (exe or 0) / ms / 1e9 if ms else 0, fam), file=sys.stderr)
    if args.per_layer_csv:
        with open(args.per_layer_csv, "w") as f:
            f.write("layer,class,kernel,plan,us_hip_events,algorithmic_flops,executed_flops,algorithmic_tflops,executed_tflops,hbm_bytes\n")
            for r in rows:
                f.write("%s,%s,\"%s\",\"%s\",%.2f,%.0f,%.0f,%.2f,%.2f,%.0f\n"
                        % (r["layer"], r["class"], r["kernel"], r["plan"], r["ms"] * 1e3, r["algorithmic_flops"],
                           r["executed_flops"] or 0, r["algorithmic_flops"] / r["ms"] / 1e9,
                           (r["executed_flops"] or 0) / r["ms"] / 1e9, r["hbm_bytes"] or 0))

    e2e = None
    if not args.no_e2e:                        # PCIe-inclusive: host batch in, host logits out
        net.streams = "1x1"                    # one full-batch graph: the same kernels as the timed plan
        net(xs_host[0])
        t0 = time.perf_counter()
        for i in range(10):
            net(xs_host[i & 1])
        e2e = n * 10 / (time.perf_counter() - t0)
    # the reference-shaped calls (net.py:94-101) on device-resident batches: net(x) one call at a time (latency plan: every
    # call joins before it returns its outputs) and the asynchronous form net.submit(x) (replicas of the throughput plan in
    # rotation -- the same graphs as the timed loop above, plus a private copy of the outputs per call)
    call_rates = {}
    if rank == 0 and not args.no_e2e:
        net.streams = "auto"
        # host ndarray in -> host ndarray out (the reference's call contract, net.py:94-101) through net.submit: pageable numpy
        # batches staged into the pinned ring by the library's copy threads, DMA on the copy stream, logits back through pinned
        # tickets; a window of twelve passes in flight (every handle owns private copies of its outputs, so the window may be
        # deeper than the pipeline: with six the host finds itself waiting for the oldest pass -- a pass with its DMA in front
        # takes longer than seven steps of the device-resident loop)
        import collections
        window, pend = 12, collections.deque()

        def host_loop(k, batches):
            got = None
            for i in range(k):
                pend.append(net.submit(batches[i & 1]))
                if len(pend) > window:
                    got = pend.popleft().get()
            while pend:
                got = pend.popleft().get()
            return got
        host_loop(20, xs_host)
        t0 = time.perf_counter()
        got = host_loop(100, xs_host)
        call_rates["net_submit_host"] = round(n * 100 / (time.perf_counter() - t0), 1)
        ref_host = net.submit(xs[1]).get()                   # (100 passes: the last one read batch 1; xs[1] is its device copy)
        call_rates["net_submit_host_max_abs_diff_vs_device_submit"] = float(np.abs(got - ref_host).max())
        px = [planer_amd.hip.pinned_empty(xs_host[0].shape), planer_amd.hip.pinned_empty(xs_host[0].shape)]
        px[0][...] = xs_host[0]
        px[1][...] = xs_host[1]
        host_loop(20, px)
        t0 = time.perf_counter()
        host_loop(100, px)
        call_rates["net_submit_pinned_host"] = round(n * 100 / (time.perf_counter() - t0), 1)
        del px
        for label, fn in (("net_call", lambda a: net(a)), ("net_submit", lambda a: net.submit(a))):
            for i in range(20):
                last = fn(xs[i & 1])
            ctx.synchronize()
            t0 = time.perf_counter()
            for i in range(100):
                last = fn(xs[i & 1])
            if label == "net_submit":
                last.done()
            sync()
            call_rates[label] = round(n * 100 / (time.perf_counter() - t0), 1)
            del last
    algo_list = [{"layer": a["layer"], "algo": a["algo"], "plan": a["plan"]} for a in plan.algos]

    if args.workload != "resnet18":
        tot = sum(c["flops"] for c in convs.values())
        print(json.dumps({"metric": "images/sec %s fp32 forward" % args.workload, "value": round(value, 1),
                          "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "dtype": "f32",
                          "parity_rel_err": parity, "parity_checked_images": checked,
                          "config": {"workload": args.workload, "per_gpu_batch": n, "fused_steps": plan.fused_steps,
                                     "streams": plan.streams, "repeat_values": repeat_values, "tune_source": tune_src,
                                     "wino_chains": wino_chains, "algos": algo_list},
                          "conv_tflops_whole_step": round(tot / (ms_per_step * 1e-3) / 1e12, 2),
                          "by_class_ms": {k: round(v["ms"], 4) for k, v in sorted(classes.items())},
                          "pcie_inclusive_images_per_sec": e2e}))
        return

    def tf(flops, ms):
        return flops / (ms * 1e-3) / 1e12 if ms else 0.0
    conv_fams = {k: v for k, v in families.items() if v["flops"] > 0 and k != "igemm-nchw"}
    dom_name = max(conv_fams, key=lambda k: conv_fams[k]["ms"])
    dom = conv_fams[dom_name]
    # the dominant family's steps as a forward runs them (one launch each, marker taken off): what `achieved` / `frac` quote
    dom_fwd_ms = sum(in_forward_ms.get(r["layer"], r["ms"]) for r in rows if r["kernel"] == dom_name)
    c3 = classes["conv3x3"]
    total_alg = sum(c["flops"] for c in convs.values())
    total_exe = sum(r["executed_flops"] or 0.0 for r in rows)
    # HBM traffic of the dominant family comes from the committed rocprofv3 PMC passes of this same command
    # (tools/profile_bench.sh: FETCH_SIZE doubled per the MI355X guide, WRITE_SIZE as reported, matched kernel by
    # kernel to the plan's steps) -- it cannot be counted from inside the process, so it describes the profiled
    # run of this build with the shipped tuning database, i.e. the same kernels as this run when tune_source is "shipped"
    traffic, traffic_src = None, None
    # ... and the same table gives the family's utilisation from the rocprofv3 kernel trace: executed FLOPs of its convs / the
    # summed average durations of ALL its kernels (roofline.frac_rocprof; the HIP-event figure above is the in-process estimate
    # of the same quantity and must agree with it)
    frac_rocprof, rocprof_us, gemm_us_rocprof = None, None, 0.0
    tpath = os.path.join(ROOT, "profiles", PROFILE_TAG + "_per_layer.csv")
    # (only for a run whose kernels ARE the profiled ones: every launch plan, algorithm and stream plan from the shipped database)
    if os.path.exists(tpath) and tune_src == "shipped":
        try:
            import csv
            fam_of = {split_step(r["layer"])[0]: r["kernel"] for r in rows if r["class"].startswith("conv")}
            tot, nk, us_sum, exe_by_layer = 0.0, 0, 0.0, {}
            for r in csv.DictReader(open(tpath)):
                base = split_step(r["layer"])[0]
                if fam_of.get(base) != dom_name:
                    continue
                us_sum += float(r["us_rocprof_avg"])
                if split_step(r["layer"])[1] == "gemm":
                    gemm_us_rocprof += float(r["us_rocprof_avg"])
                if r.get("layer_executed_flops") not in (None, "", "None"):
                    exe_by_layer[base] = float(r["layer_executed_flops"])
                if r.get("hbm_read_bytes") not in (None, ""):
                    tot += float(r["hbm_read_bytes"]) + float(r["hbm_write_bytes"])
                    nk += 1
            if us_sum > 0 and exe_by_layer:
                rocprof_us = us_sum
                frac_rocprof = round(sum(exe_by_layer.values()) / (us_sum * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
            if nk and dom["launches"]:
                traffic = round(tot / dom["launches"])
                traffic_src = ("profiles/%s_per_layer.csv: HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE) of ALL %d kernels of the family "
                               "in one forward / its %d convs" % (PROFILE_TAG, nk, dom["launches"]))
        except Exception:
            pass
    roofline = {
        "bound": "mfma",
        "kernel": dom_name,
        "definition": "dominant = the conv kernel family with the largest summed device time in one forward (HIP events, "
                      "single stream); its launch duration = every step of the fused program launched ONCE between two stream markers in forward "
                      "order (%d passes, median; the marker's own %.1f us, read off the steps that launch nothing, taken off) -- operands where a "
                      "real forward leaves them; *_back_to_back = every step launched 10 times between its markers (warm caches); achieved/frac "
                      "count the MFMA FLOPs the kernel EXECUTES (tile, K-chunk and Winograd-tile padding included), effective_* count the direct "
                      "algorithm's FLOPs" % (prof_steps, marker_ms * 1e3),
        "launches_per_forward": dom["launches"], "kernel_steps_per_forward": dom["steps"],
        "unit_of_a_launch": "one convolution of the family = all of its kernels (Winograd: transforms + 36 grouped GEMMs)",
        "avg_launch_ms": round(dom_fwd_ms / dom["launches"], 5),
        "executed_flops_per_launch": dom["executed"] / dom["launches"],
        "algorithmic_flops_per_launch": dom["flops"] / dom["launches"],
        "achieved": round(tf(dom["executed"], dom_fwd_ms), 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": round(tf(dom["executed"], dom_fwd_ms) / PEAK_FP32_MFMA_TFLOPS, 4),
        "avg_launch_ms_back_to_back": round(dom["ms"] / dom["launches"], 5),
        "frac_back_to_back": round(tf(dom["executed"], dom["ms"]) / PEAK_FP32_MFMA_TFLOPS, 4),
        "marker_us": round(marker_ms * 1e3, 2),
        "frac_rocprof": frac_rocprof,
        # the family's GEMM steps alone (20-30 us, compute-bound kernels: neither the tool's per-dispatch overhead on short kernels nor
        # warm caches under the ten-fold repetition move them): HIP events against the committed trace
        "gemm_steps": None if not gemm_us_rocprof else {
            "us_hip_events": round(sum(r["ms"] for r in rows if r["kernel"] == dom_name and r["stage"] == "gemm") * 1e3, 2),
            "us_rocprof": round(gemm_us_rocprof, 2)},
        "frac_rocprof_source": None if frac_rocprof is None else
        "profiles/%s_per_layer.csv: executed FLOPs of the family's convs / %.1f us = the summed average durations of all its "
        "kernels in the one-stream rocprofv3 kernel trace of this build (shipped tuning database)" % (PROFILE_TAG, rocprof_us),
        "effective_achieved": round(tf(dom["flops"], dom_fwd_ms), 2),
        "effective_frac": round(tf(dom["flops"], dom_fwd_ms) / PEAK_FP32_MFMA_TFLOPS, 4),
        "traffic": traffic, "traffic_source": traffic_src,
        "conv3x3": {"ms": round(c3["ms"], 4), "launches": c3["launches"],
                    "mfma_util": round(tf(c3["executed"], c3["ms"]) / PEAK_FP32_MFMA_TFLOPS, 4),
                    "effective_frac": round(tf(c3["flops"], c3["ms"]) / PEAK_FP32_MFMA_TFLOPS, 4)},
        "whole_forward_timed_run": {
            "mfma_util": round(value / world * (total_exe / n) / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4),
            "effective_frac": round(value / world * ((total_alg) / n) / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4),
            "note": "executed / algorithmic conv+dense FLOPs of one forward x images/sec of the timed pipelined run / peak"},
        "by_kernel": {k: {"ms": round(v["ms"], 4), "launches": v["launches"],
                          "mfma_util": round(tf(v["executed"], v["ms"]) / PEAK_FP32_MFMA_TFLOPS, 4),
                          "effective_frac": round(tf(v["flops"], v["ms"]) / PEAK_FP32_MFMA_TFLOPS, 4)}
                      for k, v in sorted(conv_fams.items())},
        "by_class_ms": {k: round(v["ms"], 4) for k, v in sorted(classes.items())}}
    # HBM-bound single-kernel layers: algorithmic bytes = input read once + output written once
    hbm = []

    def hbm_row(name, kind, nbytes, ms):
        return {"layer": name, "kernel": kind, "bytes": nbytes, "ms": round(ms, 5),
                "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
    for (name, kind), v in per_layer.items():
        if kind in ("maxpool_q4", "maxpool", "gap_q4", "gap"):
            src = [f for f in g["flow"] if f[1][0] == name][0]
            nbytes = 4.0 * (np.prod(shapes[src[0] if isinstance(src[0], str) else src[0][0]]) + np.prod(shapes[src[2]]))
            hbm.append(hbm_row(name, kind, nbytes, float(np.mean(v))))
    # the Winograd transform kernels (each its own plan step now): tensors read / written once
    tr = [r for r in rows if r["hbm_bytes"]]
    for r in tr:
        hbm.append(hbm_row(r["layer"], r["kind"], r["hbm_bytes"], r["ms"]))
    if tr:
        hbm.append(dict(hbm_row("all Winograd transform steps of one forward", "wino4_* / wino43_* in + out + chain",
                                sum(r["hbm_bytes"] for r in tr), sum(r["ms"] for r in tr)), steps=len(tr)))

    out = {"metric": "images/sec ResNet-18 fp32 forward", "value": round(value, 1), "unit": "images/sec",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic: standard-normal (N,3,224,224) fp32, seeded He-normal weights (planer_amd.irgen.resnet18)",
           "parity_rel_err": parity, "parity_checked_images": checked,
           "config": {"workload": "ResNet-18 planer IR (70 layers), forward, batch %d per GPU, 224x224, fp32, "
                                  "channel-quad activations, fused conv epilogues, hipGraph replay" % n,
                      "global_batch": global_batch, "per_gpu_batch": n, "parallelism": "batch-shard x%d" % world,
                      "weight_bcast_ms": round(bcast_ms, 3), "rccl_ranks": rccl_ranks,
                      "rank_images_per_sec": rank_rates, "repeat_values": repeat_values,
                      "tune_source": tune_src, "wino_chains": wino_chains, "conv_pairs": net.conv_pairs,
                      "compile_s": round(compile_s, 2), "tune_misses": tune_misses,
                      "plan_steps": [[names[0], prog.objs[names[0]].name] for _, names, _ in prog.flow],
                      "weight_exchange": ("single process" if world == 1 else "one ncclBroadcast of the uint8 blob (RCCL)"
                                          if comm.device_transport else "local upload per rank -- " + getattr(comm, "why", "")),
                      "fused_steps": plan.fused_steps,
                      "streams": plan.streams, "stream_probe": stream_probe,
                      # PCIe-inclusive throughput = host ndarray in -> host ndarray out with passes in flight (net.submit(x_host)
                      # ... .get(), the same pipeline `value` times; = net_submit_host_images_per_sec).  Until round 5 this key held
                      # the one-call-at-a-time figure, which is latency (upload + one-stream forward + download in series) and
                      # now lives under net_call_host_images_per_sec.
                      "pcie_inclusive_images_per_sec": call_rates.get("net_submit_host", None if e2e is None else round(e2e, 1)),
                      "pcie_inclusive_form": "net.submit(x_host).get(), pageable numpy batches, 12 passes in flight" if "net_submit_host" in call_rates
                                             else "net(x_host) one call at a time",
                      "net_call_host_images_per_sec": None if e2e is None else round(e2e, 1),
                      "net_call_images_per_sec": call_rates.get("net_call"),
                      "net_submit_images_per_sec": call_rates.get("net_submit"),
                      # host arrays in, host arrays out, twelve passes in flight: pageable numpy batches / batches built in
                      # hip.pinned_empty memory; and the largest difference between a host-array pass and the pass over the device copy of its batch
                      "net_submit_host_images_per_sec": call_rates.get("net_submit_host"),
                      "net_submit_pinned_host_images_per_sec": call_rates.get("net_submit_pinned_host"),
                      "net_submit_host_max_abs_diff_vs_device_submit": call_rates.get("net_submit_host_max_abs_diff_vs_device_submit"),
                      "device": ctx.arch, "cu_count": ctx.cu_count,
                      "tune_cache": os.environ.get("PLANER_HIP_TUNE_CACHE"), "settle_ms": args.settle_ms,
                      "sclk_mhz_under_load": sclk_mhz,
                      "timed_region_ms": round(elapsed * 1e3, 3),
                      "algos": algo_list},
           "roofline": roofline, "roofline_hbm": hbm,
           "per_layer": [{"layer": r["layer"], "kernel": r["kernel"].split(" ")[0], "us": round(r["ms"] * 1e3, 2),
                          "algorithmic_flops": r["algorithmic_flops"], "executed_flops": r["executed_flops"],
                          "hbm_bytes": r["hbm_bytes"]} for r in rows]}
    if cpu_rep is not None:
        out["cpu_baseline"] = cpu_rep
        out["gpu_over_cpu"] = round(value / cpu_rep["value"], 1)
    if not args.no_extra:
        # the other BASELINE configurations on this box, one forward at a time (bounded: ~1 s of timing each)
        del plan
        out["extra"] = secondary_workloads(planer_amd, ctx)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
