#!/usr/bin/env python
"""Which tile configuration of conv_q4_kernel should the layer3 / layer4 Winograd GEMMs of ResNet-18 run with when the
three-stream plan is what is timed?  The shipped launch plans were picked on isolated launches; here every candidate is
layered over the shipped database through a user tune cache (PLANER_HIP_TUNE_CACHE) and judged by the pipeline's rate
(tools/throughput_probe.py, best of four 150-step runs).  Run on the GPU box:

    python tools/gemm_cfg_timed.py [--keys l3,l4,l3+l4] [--cfgs q128x64x32,q64x128x32,...]
"""
import argparse
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..")
DB = os.path.join(ROOT, "planer_amd", "tuned", "gfx950_cu256.plans")
KEYS = {"l3": "2 1 9216 128 4 9216 1 1 1 1 1 1 0 0 36 0 0 0", "l4": "2 1 18432 64 2 18432 1 1 1 1 1 1 0 0 36 0 0 0"}


def rate(lines, tag):
    with tempfile.NamedTemporaryFile("w", suffix=".plans", delete=False) as f:
        f.write("".join(l + "\n" for l in lines))
    env = dict(os.environ, PLANER_HIP_TUNE_CACHE=f.name, TAG=tag)
    out = subprocess.run([sys.executable, os.path.join(HERE, "throughput_probe.py")], env=env, capture_output=True, text=True)
    os.unlink(f.name)
    last = [l for l in out.stdout.splitlines() if "img/s" in l]
    return last[-1] if last else "FAILED: " + out.stderr[-300:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keys", default="l3,l4,l3+l4")
    ap.add_argument("--cfgs", default="q64x64x32,q128x64x32,q64x128x32,q128x128x16,q128x128x32,q128x32x32,q256x64x16,q64x256x16,q64x64x16")
    args = ap.parse_args()
    have = [l.strip() for l in open(DB) if l.strip()]
    for k in KEYS.values():
        assert any(l.startswith(k + " ") for l in have), "key not in the shipped database: " + k
    print(rate([], "shipped"), flush=True)
    for ks in args.keys.split(","):
        for cfg in args.cfgs.split(","):
            lines = ["%s %s 0 1 0" % (KEYS[k], cfg) for k in ks.split("+")]
            print(rate(lines, "%s %s" % (ks, cfg)), flush=True)


if __name__ == "__main__":
    main()
