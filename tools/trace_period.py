"""Digest a rocprofv3 --kernel-trace csv of a loop of identical forward passes (tools/latency_bench.py): finds the period of the
kernel-name sequence, then prints -- per kernel name, in order of first appearance inside a period -- launches per forward, mean
duration and mean gap before it, over the complete periods of the trace's second half; and busy / gap / span per forward.
usage: trace_period.py <kernel_trace.csv> [--list]   (--list: every launch of one period in order)"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:70] for r in rows]
n = len(names)
tail = names[n // 2:]
best = None
for p in range(20, min(len(tail) // 3, 2000)):
    ok = sum(tail[i] == tail[i + p] for i in range(len(tail) - p))
    if ok >= 0.98 * (len(tail) - p):
        best = p
        break
if best is None:
    sys.exit("no period found")
p = best
# align: a period starts where the gap before the kernel is largest on average (host turnaround between forwards)
s0 = n // 2
gaps = [0.0] * p
cnt = [0] * p
for i in range(s0 + 1, n):
    g = int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])
    gaps[(i - s0) % p] += g
    cnt[(i - s0) % p] += 1
phase = max(range(p), key=lambda k: gaps[k] / max(cnt[k], 1))
start = s0 + phase
periods = [(a, a + p) for a in range(start, n - p + 1, p) if names[a:a + p] == names[start:start + p]]
order, agg = [], {}
busy = gapt = 0.0
for a, b in periods:
    for i in range(a, b):
        d = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
        g = (int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3 if i > a else 0.0
        k = names[i]
        if k not in agg:
            agg[k] = [0, 0.0, 0.0]
            order.append(k)
        agg[k][0] += 1; agg[k][1] += d; agg[k][2] += g
        busy += d; gapt += g
np_ = len(periods)
print("period %d kernels, %d complete periods" % (p, np_))
for k in order:
    c, d, g = agg[k]
    print("%-70s x%5.1f  %7.2f us each  gap %5.2f  total %7.1f us" % (k, c / np_, d / c, g / c, d / np_))
print("per forward: busy %.1f us, gaps %.1f us, span %.1f us" % (busy / np_, gapt / np_, (busy + gapt) / np_))
if "--list" in sys.argv:
    a, b = periods[-1]
    for i in range(a, b):
        d = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
        g = (int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3
        print("%3d %-70s grid %8s  %7.2f us  gap %5.2f" % (i - a, names[i], rows[i].get("Grid_Size_X", rows[i].get("Grid_Size", "?")), d, g))
