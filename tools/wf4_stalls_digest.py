#!/usr/bin/env python
"""Digest of tools/wf4_stalls.sh: per instantiation of conv_wf4_kernel, the mean of every SQ counter per dispatch, the mean
duration in the same traced pass, and a cycle budget in wave-cycle shares (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~
WAVE_CYCLES per the MI355X guide's PMC section; SQ_* wave counters count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles).

    python tools/wf4_stalls_digest.py gpurun_out/wf4_stalls > digest.md
"""
import collections
import csv
import glob
import re
import sys

out = sys.argv[1]
means = collections.defaultdict(dict)        # instantiation -> counter -> mean per dispatch
durs = collections.defaultdict(list)         # instantiation -> [us] (all passes)
meta = {}


def inst(name):
    m = re.search(r"conv_wf4_kernel<([^>]*)>", name)
    return m.group(1).replace(" ", "") if m else None


for f in sorted(glob.glob(out + "/p*_counter_collection.csv")):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = inst(r["Kernel_Name"])
        if k is None:
            continue
        per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[k] = (r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("LDS_Block_Size"), r.get("VGPR_Count"),
                   r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size", r.get("Private_Segment_Size")))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for (k, _), cs in per.items():
        for c, v in cs.items():
            agg[k][c].append(v)
    for k, cs in agg.items():
        for c, vs in cs.items():
            vs = sorted(vs)[len(vs) // 10: len(vs) - len(vs) // 10] or vs      # trim the first-launch outliers
            means[k][c] = sum(vs) / len(vs)
for f in sorted(glob.glob(out + "/p*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        k = inst(r["Kernel_Name"])
        if k:
            durs[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)

print("# conv_wf4_kernel: stall counters per launch (tools/wf4_stalls.sh, MI355X)\n")
print("Counters are sums over the chip per dispatch, means over the dispatches of four traced passes (first and last tenth")
print("trimmed).  Wave counters count quad-cycles (x 4 = shader cycles).\n")
try:
    print("Untraced times of the same command:\n\n```\n" + open(out + "/untraced.log").read().strip() + "\n```\n")
except OSError:
    pass
for k in sorted(means):
    m = means[k]
    d = sorted(durs[k])
    dur = d[len(d) // 2] if d else float("nan")
    grid, wg, lds, vgpr, agpr, sgpr, scratch = meta[k]
    nwg = int(grid) // int(wg) if grid and wg else 0
    print("## conv_wf4_kernel<%s>\n" % k)
    print("grid %s / workgroup %s = **%d workgroups**, LDS %s B, VGPR %s, AGPR %s, SGPR %s, scratch %s; median duration under the tool **%.2f us** (%d dispatches)\n"
          % (grid, wg, nwg, lds, vgpr, agpr, sgpr, scratch, dur, len(d)))
    print("| counter | per launch |")
    print("|---|---|")
    for c in sorted(m):
        print("| %s | %.4g |" % (c, m[c]))
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        print("\nCycle budget (share of SQ_WAVE_CYCLES = resident wave time; x duration = us of the launch):\n")
        print("| term | share | us |")
        print("|---|---|---|")
        rows = [("wave parked: s_waitcnt / barrier (SQ_WAIT_ANY)", "SQ_WAIT_ANY"),
                ("issue stall: pipe / dependency (SQ_WAIT_INST_ANY)", "SQ_WAIT_INST_ANY"),
                ("   of which LDS issue stall (SQ_WAIT_INST_LDS)", "SQ_WAIT_INST_LDS"),
                ("issuing (SQ_ACTIVE_INST_ANY)", "SQ_ACTIVE_INST_ANY"),
                ("   VALU incl. MFMA issue (SQ_ACTIVE_INST_VALU)", "SQ_ACTIVE_INST_VALU"),
                ("   LDS (SQ_ACTIVE_INST_LDS)", "SQ_ACTIVE_INST_LDS"),
                ("   VMEM (SQ_ACTIVE_INST_VMEM)", "SQ_ACTIVE_INST_VMEM"),
                ("   scalar (SQ_ACTIVE_INST_SCA)", "SQ_ACTIVE_INST_SCA"),
                ("   misc: barrier / nop / setprio (SQ_ACTIVE_INST_MISC)", "SQ_ACTIVE_INST_MISC")]
        for label, c in rows:
            if c in m:
                print("| %s | %.3f | %.1f |" % (label, m[c] / wc, m[c] / wc * dur))
        waves = m.get("SQ_WAVES", 0)
        if waves:
            print("\nwaves %.0f; wave-cycles per wave %.0f quad-cycles = %.1f us at 2.4 GHz (the launch: %.1f us)"
                  % (waves, wc / waves, wc / waves * 4 / 2400.0, dur))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
        busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        print("\nmatrix-pipe busy: %.4f of all 1024 SIMDs over the launch (%.4f of the SIMDs of the %d CUs that hold a workgroup)"
              % (busy, busy * 256.0 / max(min(nwg, 256), 1), min(nwg, 256)))
    if "SQ_INSTS_MFMA" in m:
        print("MFMA instructions %.4g -> %.3f GFLOP executed (2048 per v_mfma_f32_16x16x4_f32)" % (m["SQ_INSTS_MFMA"], m["SQ_INSTS_MFMA"] * 2048 / 1e9))
    if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m:
        print("LDS bank-conflict cycles / LDS active cycles: %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1)))
    print()
