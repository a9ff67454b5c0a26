#!/usr/bin/env python
"""Headline benchmark: images/sec of the ResNet-18 fp32 forward pass on
MI355X, next to the numpy-CPU baseline, with the roofline of the dominant
kernel (the 3x3 implicit-GEMM convolutions).

    python bench.py --gpus N --steps K --warmup W

N>1 is launched by the driver as `python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU); torchrun is
only the process spawner -- ranks exchange the RCCL id through /tmp and the
weight blob through ONE RCCL broadcast over xGMI; the forward pass itself has
no collective (batch shards are independent), so scaling is weak: 32 images
per GPU.  A step = one captured forward pass over one resident batch.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0
PER_GPU_BATCH = 32                 # BASELINE.json configs[2] / configs[3]: 32 images per GPU


def conv_flops(g, shapes):
    """Algorithmic FLOPs (2*MAC) of every conv layer of the graph, by layer name."""
    kinds = {n: (k, p) for n, k, p in g["layers"]}
    out = {}
    for src, names, dst in g["flow"]:
        kind, para = kinds[names[0]]
        if kind == "conv":
            n, cout, ho, wo = shapes[dst]
            _, cin_g, kh, kw = shapes[src[1]]
            out[names[0]] = (2.0 * n * cout * ho * wo * cin_g * kh * kw, (kh, kw))
    return out


def cpu_baseline(g, b, x, iters):
    """The numpy restatement of the reference (oracle/planer_np.py: im2col +
    OpenBLAS sgemm, all host cores) timed on a bounded sample: `iters`
    forwards of the same batch-32 workload after one warm-up."""
    from oracle import planer_np as onp
    net = onp.OracleNet()
    net.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    net.load_weights(b)
    net(x.copy())
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        net(x.copy())
        ts.append(time.perf_counter() - t0)
    threads = len(os.sched_getaffinity(0))
    try:
        import threadpoolctl
        info = [i for i in threadpoolctl.threadpool_info() if i.get("user_api") == "blas"]
        if info:
            threads = int(info[0]["num_threads"])
    except Exception:
        pass
    med = float(np.median(ts))
    return {"value": x.shape[0] / med, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "%d forwards of ResNet-18 batch %d after 1 warm-up, median; best %.1f img/s; "
                      "host has %d logical CPUs" % (iters, x.shape[0], x.shape[0] / min(ts), os.cpu_count())}


def main():
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
    ap.add_argument("--e2e", action="store_true", help="also time net(x_host): H2D + forward + D2H (PCIe-inclusive)")
    args = ap.parse_args()

    import planer_amd
    from planer_amd import dist
    from planer_amd.irgen import resnet18, yolov3
    from planer_amd.irgen.builder import GraphBuilder

    rank, world, _ = dist.env_world()
    if world != args.gpus:
        if args.gpus != 1 and world == 1:
            sys.exit("bench.py --gpus %d must be launched with one process per GPU "
                     "(python -m torch.distributed.run --nproc-per-node %d bench.py ...)" % (args.gpus, args.gpus))
    ctx = planer_amd.hip.context()
    comm = dist.init(ctx, fallback=True)      # RCCL; same-node file fallback if it cannot come up

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
    # weights come from rank 0 by ONE RCCL broadcast; only the file fallback regenerates the
    # (seeded) blob on every rank
    g, blob = build() if (rank == 0 or not comm.device_transport) else (build()[0], None)
    net = planer_amd.Net(ctx)
    net.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    t0 = time.perf_counter()
    comm.load_weights(net, blob)
    ctx.synchronize()
    bcast_ms = (time.perf_counter() - t0) * 1e3

    # ---- data: this rank's shard of the global synthetic batch, resident in HBM ---
    n = args.batch
    global_batch = n * world
    lo, hi = dist.shard_range(global_batch, world, rank)
    xs_host = [np.random.default_rng(1 + 1000 * i + rank).standard_normal((hi - lo,) + in_shape).astype(np.float32)
               for i in range(2)]
    xs = [planer_amd.asarray(a, ctx=ctx) for a in xs_host]
    # fuse + tune + warm the pool + capture the hipGraph(s); "throughput": sub-batch streams free-run
    plan = net.compile(xs[0], mode="throughput")
    ctx.save_tune_cache()                       # no-op unless PLANER_HIP_TUNE_CACHE is set
    state = {"i": 0}

    def step():
        if not os.environ.get("PLANER_BENCH_NOFEED"):
            plan.feed([xs[state["i"] & 1]])        # rotate two distinct resident batches
        plan.launch(join=False)
        state["i"] += 1

    def sync():
        plan.join()                            # side streams -> main stream
        ctx.synchronize()

    elapsed = dist.timed_steps(comm, step, sync, args.steps, args.warmup)
    ms_per_step = elapsed / args.steps * 1e3
    value = global_batch * args.steps / elapsed

    # ---- correctness guard on what was just timed (cheap: logits of 2 images) ----
    sync()
    logits = plan.outputs[0].get() if isinstance(plan.outputs, tuple) else plan.outputs.get()
    assert np.isfinite(logits).all()

    if rank != 0:
        return

    # ---- roofline of the dominant kernel: HIP events around every layer of the
    #      same fused program, launched eagerly on the same stream, K steps ----------
    shapes = {k: a.shape for k, a in zip(net.inits, net.weights)}
    shapes[g["input"][0]] = xs[0].shape
    net._interpret(net._program, [xs[0].copy()], shapes=shapes)
    flops = conv_flops(g, shapes)
    prog, _ = net._fuse(shapes)
    per_layer = {}
    prof_steps = min(args.steps, 20)
    for it in range(prof_steps + 2):
        net._interpret(prog, [xs[it & 1].copy()], profile=True)
        if it >= 2:
            for name, kind, ms in net.last_events:
                per_layer.setdefault((name, kind), []).append(ms)
    classes = {}
    for (name, kind), v in per_layer.items():
        ms = float(np.mean(v))
        base = name[:-1] if name.endswith("+") else name
        if base in flops:
            f, (kh, kw) = flops[base]
            cls = "conv%dx%d" % (kh, kw)
        else:
            f, cls = 0.0, kind
        c = classes.setdefault(cls, {"ms": 0.0, "flops": 0.0, "launches": 0})
        c["ms"] += ms
        c["flops"] += f
        c["launches"] += 1
        if args.detail:
            print("%-14s %-10s %8.3f ms %8.2f TFLOP/s" % (name, cls, ms, f / ms / 1e9 if ms else 0), file=sys.stderr)
    e2e = None
    if args.e2e:
        net(xs_host[0])
        t0 = time.perf_counter()
        for i in range(10):
            net(xs_host[i & 1])
        e2e = n * 10 / (time.perf_counter() - t0)
    if args.workload != "resnet18":
        tot = sum(f for f, _ in flops.values())
        print(json.dumps({"metric": "images/sec %s fp32 forward" % args.workload, "value": round(value, 1),
                          "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "dtype": "f32",
                          "config": {"workload": args.workload, "per_gpu_batch": n, "fused_steps": plan.fused_steps,
                                     "streams": plan.streams},
                          "conv_tflops_whole_step": round(tot / (ms_per_step * 1e-3) / 1e12, 2),
                          "by_class_ms": {k: round(v["ms"], 4) for k, v in sorted(classes.items())},
                          "pcie_inclusive_images_per_sec": e2e}))
        return
    c3 = classes["conv3x3"]
    achieved = c3["flops"] / (c3["ms"] * 1e-3) / 1e12
    total_flops = sum(f for f, _ in flops.values()) + 2.0 * n * 512 * 1000   # + the 512->1000 dense layer
    # HBM traffic cannot be counted from inside this process: it comes from the committed rocprofv3
    # PMC digest of this same command (tools/profile_bench.sh -> profiles/r01_hbm_traffic.json),
    # FETCH_SIZE doubled as the MI355X guide prescribes for gfx950, averaged per conv-family launch.
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    if os.path.exists(tpath):
        fam = [r for r in json.load(open(tpath))
               if any(t in r["kernel"] for t in ("conv_q4_kernel", "conv_w1d_kernel", "conv_w1d4_kernel", "conv_tap_kernel", "conv_igemm_kernel",
                                                 "reduce_tiles", "wino_"))]
        launches = sum(r["launches"] for r in fam)
        if launches:
            traffic = round(sum((r["read_mb_per_launch_corrected"] * 1e6 + r["write_kb_per_launch"] * 1e3)
                                * r["launches"] for r in fam) / launches)
            traffic_src = "profiles/r01_hbm_traffic.json: HBM bytes per conv-family kernel launch (PMC run of this command)"
    roofline = {"bound": "mfma", "kernel": "the 16 conv3x3 layers of one forward on channel-quad tensors, each on the fastest of "
                                          "conv_q4_kernel (direct implicit GEMM, stride-2 layers), conv_w1d4_kernel (fused 1-D "
                                          "Winograd F(4,3), layer1) and the 2-D Winograd pipelines F(2x2,3x3) / F(4x4,3x3) (transforms + one grouped "
                                          "1x1 conv_q4_kernel, layer2-4), incl. split-K tile-reduce and transform launches; "
                                          "achieved = algorithmic FLOPs / time",
                "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(c3["ms"] / c3["launches"], 4),
                "flops_per_launch": c3["flops"] / c3["launches"],
                "whole_forward_mfma_frac": round(value / world * (total_flops / n) / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4),
                "whole_forward_note": "ALGORITHMIC FLOPs of the timed (pipelined) run / peak; the Winograd paths execute "
                                      "1.5x / 2x (fused 1-D F(2,3) / F(4,3)), 2.25x (F(2x2,3x3)) or 4x (F(4x4,3x3)) fewer multiplies than that",
                "by_class_ms": {k: round(v["ms"], 4) for k, v in sorted(classes.items())}}

    out = {"metric": "images/sec ResNet-18 fp32 forward", "value": round(value, 1), "unit": "images/sec",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic: standard-normal (N,3,224,224) fp32, seeded He-normal weights (planer_amd.irgen.resnet18)",
           "config": {"workload": "ResNet-18 planer IR (70 layers), forward, batch %d per GPU, 224x224, fp32, "
                                  "channel-quad activations, fused conv epilogues, hipGraph replay" % n,
                      "global_batch": global_batch, "per_gpu_batch": n, "parallelism": "batch-shard x%d" % world,
                      "weight_bcast_ms": round(bcast_ms, 2),
                      "weight_exchange": ("single process" if world == 1 else "one ncclBroadcast of the uint8 blob (RCCL)"
                                          if comm.device_transport else "local upload per rank -- " + getattr(comm, "why", "")),
                      "fused_steps": plan.fused_steps,
                      "streams": plan.streams,
                      "pcie_inclusive_images_per_sec": None if e2e is None else round(e2e, 1),
                      "device": ctx.arch, "cu_count": ctx.cu_count},
           "roofline": roofline}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(g, blob, xs_host[0], args.cpu_iters)
        out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
