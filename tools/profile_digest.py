"""Digest of tools/profile_bench.sh's rocprofv3 outputs (runs on the GPU box, or here on merged files).

    python tools/profile_digest.py <dir> <tag> [--install]

Writes into <dir>:
  <tag>_hbm_traffic.json / .md   HBM bytes per launch per kernel (FETCH_SIZE x2 per the MI355X guide's gfx950
                                 correction, WRITE_SIZE as reported)
  <tag>_mfma_util.md / .json     per kernel: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = matrix-pipe
                                 utilisation; MFMA instructions and 'MOPS' per launch; VALU instructions
  <tag>_per_layer.csv            one forward of the one-stream run: layer, kernel, us (rocprof), algorithmic FLOPs,
                                 executed FLOPs, algorithmic bytes -- layers from the bench line's config.algos
--install copies the judged files into profiles/.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

SIMDS = 256 * 4
XCDS = 8            # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs' GRBMs (checked against timestamps)


def short(k):
    for t in ("void ", "(anonymous namespace)::"):
        k = k.replace(t, "")
    k = k.replace("HIP_vector_type<float, 4u>", "f4")
    m = re.match(r"([\w:]+(?:<.*?>)?)\(", k)
    return (m.group(1) if m else k.split("(")[0]).replace("QuadCfg", "Q")


def counters(out, tag, idx):
    f = glob.glob(os.path.join(out, "%s_pmc%d_counter_collection.csv" % (tag, idx)))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])) if f else []:
        k = short(r["Kernel_Name"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return {k: {c: v / len(disp[k]) for c, v in cs.items()} for k, cs in per.items()}, {k: len(v) for k, v in disp.items()}


def main():
    out, tag = sys.argv[1], sys.argv[2]
    rd, nrd = counters(out, tag, 1)
    wr, _ = counters(out, tag, 2)
    sq, nsq = counters(out, tag, 3)
    # ---- HBM traffic -------------------------------------------------------------------------------
    rows = []
    for k in sorted(rd, key=lambda k: -rd[k].get("FETCH_SIZE", 0) * nrd[k]):
        rows.append({"kernel": k, "launches": nrd[k], "fetch_kb_per_launch": round(rd[k].get("FETCH_SIZE", 0), 1),
                     "read_mb_per_launch_corrected": round(2 * rd[k].get("FETCH_SIZE", 0) / 1e3, 3),
                     "write_kb_per_launch": round(wr.get(k, {}).get("WRITE_SIZE", 0), 1)})
    json.dump(rows, open(os.path.join(out, tag + "_hbm_traffic.json"), "w"), indent=1)
    with open(os.path.join(out, tag + "_hbm_traffic.md"), "w") as f:
        f.write("HBM traffic per launch, one-stream run of `bench.py` (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
                "passes; FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B, MI355X guide).\n\n")
        f.write("| kernel | launches | FETCH_SIZE KB/launch | corrected read MB/launch | WRITE_SIZE KB/launch |\n|---|---|---|---|---|\n")
        for r in rows:
            f.write("| `%s` | %d | %.0f | %.2f | %.0f |\n" % (r["kernel"], r["launches"], r["fetch_kb_per_launch"],
                                                             r["read_mb_per_launch_corrected"], r["write_kb_per_launch"]))
    # ---- MFMA utilisation ----------------------------------------------------------------------------
    util = []
    for k, c in sorted(sq.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) * nsq[kv[0]]):
        busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        util.append({"kernel": k, "launches": nsq[k], "mfma_busy_cycles": busy, "gui_active_cycles": gui,
                     "mfma_util": round(busy / (gui / XCDS * SIMDS), 4) if gui else None,
                     "mfma_insts": c.get("SQ_INSTS_MFMA", 0.0), "mfma_mops_f32": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0),
                     "valu_insts": c.get("SQ_INSTS_VALU", 0.0), "busy_cu_cycles": c.get("SQ_BUSY_CU_CYCLES", 0.0)})
    json.dump(util, open(os.path.join(out, tag + "_mfma_util.json"), "w"), indent=1)
    with open(os.path.join(out, tag + "_mfma_util.md"), "w") as f:
        f.write("Matrix-pipe utilisation per kernel, one-stream run of `bench.py` (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES "
                "SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU "
                "GRBM_GUI_ACTIVE; per-launch averages).  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): "
                "the fraction of SIMD-cycles the matrix pipe was busy while the kernel ran (v_mfma_f32_32x32x2_f32 = 64 busy cycles).\n\n")
        f.write("| kernel | launches | MFMA busy cycles | GUI active cycles | mfma_util | MFMA insts (SQ_INSTS_MFMA) | VALU insts |\n|---|---|---|---|---|---|---|\n")
        for u in util:
            if u["mfma_busy_cycles"] or "conv" in u["kernel"] or "wino" in u["kernel"]:
                f.write("| `%s` | %d | %.3g | %.3g | %s | %.3g | %.3g |\n" % (u["kernel"], u["launches"], u["mfma_busy_cycles"],
                        u["gui_active_cycles"], u["mfma_util"], u["mfma_insts"], u["valu_insts"]))
    # ---- per-layer table from the one-stream kernel trace -------------------------------------------
    trace = glob.glob(os.path.join(out, tag + "_bench_1stream_kernel_trace.csv"))
    line = os.path.join(out, tag + "_bench_1stream_line.json")
    if trace and os.path.exists(line):
        try:
            per_layer(out, tag, trace[0], json.loads(open(line).read().strip().splitlines()[-1]))
        except Exception as e:                                    # noqa: BLE001
            print("per-layer table failed:", e)
    print(open(os.path.join(out, tag + "_mfma_util.md")).read())
    print(open(os.path.join(out, tag + "_hbm_traffic.md")).read())
    if "--install" in sys.argv:
        install(out, tag)


def expected_kernels(a):
    """Kernel-name fragments one layer launches, in order, from its config.algos entry."""
    plan, algo = a["plan"], a["algo"]
    if plan.startswith("dense32x32"):
        return ["dense_small_kernel"]
    if plan.startswith("pair["):
        return ["conv_q4_pair_kernel"]
    plan = re.sub(r"^wino\d+\[(.*)\]$", r"\1", plan) if algo == "direct" else plan
    if plan.startswith("as128"):
        return ["wino4_gemm_as_kernel"]      # filter-stationary GEMM stage of a 128-channel F(4x4,3x3) conv
    if plan.startswith("stem+maxpool(nchw)"):
        return ["conv_stem_pool_kernel"]     # reads the NCHW batch itself; runs when the plan is fed, in front of the graph
    if plan.startswith("stem+maxpool"):
        return ["nchw_to_rowpack_kernel", "conv_stem_pool_kernel"]      # the re-layout runs when the plan is fed
    if plan.startswith("smallcin3x3valu"):
        return ["conv_smallcin_valu_kernel"]
    gemm = ["conv_ks_kernel" if "[k" in plan or plan.startswith("k") else "conv_q4_kernel" if "[q" in plan or plan.startswith("q") else "conv_igemm_kernel"]
    if re.search(r"split=([2-9]|\d\d)", plan):
        gemm.append("reduce_tiles")
    if algo.startswith("wf4"):
        return ["conv_wf4_kernel"]
    if algo.startswith("rowpack"):
        return ["nchw_to_rowpack_kernel"] + gemm
    if algo.startswith("w1d4"):
        return ["conv_w1d4_kernel"]
    if algo.startswith("wino43"):
        return ["wino43_"] + gemm + ["wino43_"]
    if algo.startswith("wino4x4"):
        return ["wino4_input_"] + gemm + ["wino4_output_"]          # ..._q4_kernel or the row-split ..._rows_q4_kernel
    if algo.startswith("wino2x2"):
        return ["wino_input_q4_kernel"] + gemm + ["wino_output_q4_kernel"]
    return gemm


def step_kernels(step, kind, algos):
    """Kernel-name fragments a plan step launches, in order (config.plan_steps of the bench line); None = unknown kind."""
    a = algos.get(step.split("@")[0])
    if kind in ("conv_q4", "conv_fused", "conv", "dense", "matmul", "conv_q4_pair", "conv_pool_q4"):
        return expected_kernels(a) if a else None
    if kind in ("wino4_in", "wino4_out", "wino4_chain"):
        return ["wino4_"]                    # wino4_chain_kernel<..> (LDS) or wino4_input_ / wino4_output_ (register kernels)
    if kind in ("wino43_in", "wino43_out", "wino43_chain"):
        return ["wino43_"]                   # wino43_lds_kernel<..> or the register kernels wino43_input_ / wino43_output_
    if kind in ("wino4_gemm", "wino43_gemm"):
        return expected_kernels(dict(a, algo="direct")) if a else None
    return {"maxpool_q4": ["pool"], "gap_q4": ["gap_q4_kernel"], "to_q4": ["nchw_to_q4"], "from_q4": ["q4_to_nchw"],
            "flatten": [], "return": [], "identity": []}.get(kind)


def matched_forwards(names, seq):
    """Start indices of every run of `names` that launches exactly the plan's kernel sequence."""
    first = seq[0][1]
    ok = []
    for s in (i for i, n in enumerate(names) if first in n):
        if s + len(seq) <= len(names) and all(frag in names[s + i] for i, (_, frag) in enumerate(seq)):
            ok.append(s)
    return ok


def pmc_dispatches(out, tag, idx):
    """[(kernel, {counter: value})] in dispatch order for one --pmc pass."""
    f = glob.glob(os.path.join(out, "%s_pmc%d_counter_collection.csv" % (tag, idx)))
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])) if f else []:
        if "rocclr" in r["Kernel_Name"]:
            continue
        d = disp.setdefault(int(r["Dispatch_Id"]), [short(r["Kernel_Name"]), {}])
        d[1][r["Counter_Name"]] = d[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [disp[k] for k in sorted(disp)]


def per_layer(out, tag, trace, bench):
    algos = {a["layer"]: a for a in bench["config"]["algos"]}
    seq = []                                  # (plan step, kernel fragment)
    for step, kind in bench["config"]["plan_steps"]:
        frags = step_kernels(step, kind, algos)
        if frags is None:
            print("per-layer table: no kernel list for step %s (%s)" % (step, kind))
            return
        seq += [(step, f) for f in frags]
    rows = [r for r in csv.DictReader(open(trace)) if "rocclr" not in r["Kernel_Name"]]
    names = [short(r["Kernel_Name"]) for r in rows]
    acc = collections.OrderedDict()
    starts = matched_forwards(names, seq)
    for s in starts:
        for i, (layer, frag) in enumerate(seq):
            r = rows[s + i]
            acc.setdefault((i, layer, names[s + i]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    # HBM bytes of the same kernels from the FETCH_SIZE / WRITE_SIZE passes (their own runs of the same command)
    hbm = {}
    for idx, counter, scale in ((1, "FETCH_SIZE", 2.0 * 1e3), (2, "WRITE_SIZE", 1e3)):     # KB; FETCH doubled (gfx950, MI355X guide)
        d = pmc_dispatches(out, tag, idx)
        dn = [k for k, _ in d]
        for s in matched_forwards(dn, seq) if d else []:
            for i, (layer, frag) in enumerate(seq):
                hbm.setdefault((i, counter), []).append(d[s + i][1].get(counter, 0.0) * scale)
    per = {r["layer"]: r for r in bench.get("per_layer", [])}
    nfw = len(starts)
    with open(os.path.join(out, tag + "_per_layer.csv"), "w") as f:
        f.write("layer,kernel,us_rocprof_avg,us_rocprof_min,forwards_matched,layer_algorithmic_flops,layer_executed_flops,"
                "layer_us_hip_events,hbm_read_bytes,hbm_write_bytes,algorithmic_hbm_bytes\n")
        for (i, layer, kern), v in acc.items():
            if len(v) * 10 < nfw:
                continue                      # a handful of eager passes launch a layer's kernels in another order
            p = per.get(layer, {})
            rd, wr = hbm.get((i, "FETCH_SIZE")), hbm.get((i, "WRITE_SIZE"))
            f.write("%s,\"%s\",%.2f,%.2f,%d,%s,%s,%s,%s,%s,%s\n" % (
                layer, kern, sum(v) / len(v), min(v), len(v), p.get("algorithmic_flops", ""), p.get("executed_flops", ""),
                p.get("us", ""), "%.0f" % (sum(rd) / len(rd)) if rd else "", "%.0f" % (sum(wr) / len(wr)) if wr else "",
                "%.0f" % p["hbm_bytes"] if p.get("hbm_bytes") else ""))
    print("per-layer table: %d forwards of %d kernels matched" % (nfw, len(seq)))


def install(out, tag):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(root, "profiles")
    for name in ("_bench_kernel_stats.csv", "_bench_1stream_kernel_stats.csv", "_hbm_traffic.json", "_hbm_traffic.md",
                 "_mfma_util.json", "_mfma_util.md", "_per_layer.csv", "_bench_line.json", "_bench_line_under_rocprof.json",
                 "_bench_1stream_line.json", "_tune_cache.txt", "_tune_cache.txt.algo.json", "_other_workloads.jsonl"):
        src = os.path.join(out, tag + name)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(dst, tag + name))
            print("installed", tag + name)


if __name__ == "__main__":
    main()
