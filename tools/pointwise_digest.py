#!/usr/bin/env python
"""rocprofv3 kernel trace of tools/pointwise_gbs.py -> markdown table of achieved GB/s per stand-alone HBM-bound layer kernel.

    python tools/pointwise_digest.py <kernel_trace.csv> <manifest.json> [hip_event_table.md]

The tool under trace separates its cases by three consecutive fill-buffer dispatches; every other dispatch of a segment belongs
to that case: device time of the segment / launches = time per layer call (a Concatenate is one copy kernel per input)."""
import csv
import json
import sys

PEAK = 8000.0


def main():
    trace, manifest = sys.argv[1], json.load(open(sys.argv[2]))
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    segs, cur, fills = [], [], 0
    for r in rows:
        name = r["Kernel_Name"]
        if "fillBuffer" in name or "FillBuffer" in name:
            fills += 1
            if fills == 3:
                segs.append(cur)
                cur = []
            continue
        fills = 0
        cur.append((name, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    segs.append(cur)
    segs = segs[1:]                    # before the first separator: uploads of the operands (no kernels) / warm-up
    if len(segs) != len(manifest):
        sys.exit("trace has %d segments, manifest %d cases" % (len(segs), len(manifest)))
    print("| workload | layer (shape) | kernel(s) | algorithmic MB | us per call (rocprofv3) | GB/s | of 8 TB/s |")
    print("|---|---|---|---|---|---|---|")
    for m, seg in zip(manifest, segs):
        ns = sum(d for _, d in seg)
        us = ns / 1e3 / m["launches"]
        names = sorted({n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0] for n, _ in seg})
        per_call = len(seg) / m["launches"]
        gbs = m["bytes"] / us / 1e3
        print("| %s | %s | %s%s | %.2f | %.2f | %.0f | %.3f |" % (m["net"], m["case"], ", ".join(names),
                                                                 " x%g" % per_call if per_call != 1 else "", m["bytes"] / 1e6, us, gbs, gbs / PEAK))


if __name__ == "__main__":
    main()
