"""Digest a rocprofv3 --kernel-trace csv of tools/latency_bench.py: the LAST complete forward's kernels in launch
order with duration and the gap to the previous kernel's end.  usage: trace_forward.py <kernel_trace.csv> <n_kernels>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if not n:                                           # period = distance between the last two launches of the first-kernel name
    names = [r["Kernel_Name"] for r in rows]
    last = len(names) - 1
    first = names[last]
    prev = max(i for i in range(last) if names[i] == first)
    n = last - prev
    # find a true period: smallest p >= n such that names[-p:] == names[-2p:-p]
    p = max(n, 8)                                   # (several result copies in a row are not a period)
    while names[-p:] != names[-2 * p:-p]:
        p += 1
    n = p
seg = rows[-n:]
prev_end = int(rows[-n - 1]["End_Timestamp"])
tot_k = tot_g = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
    print("%-60s grid %7s wg %4s  %7.2f us  gap %6.2f" % (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), (e - s) / 1e3, (s - prev_end) / 1e3))
    tot_k += e - s; tot_g += s - prev_end; prev_end = e
print("kernels %d: busy %.1f us, gaps %.1f us, span %.1f us" % (n, tot_k / 1e3, tot_g / 1e3, (tot_k + tot_g) / 1e3))
