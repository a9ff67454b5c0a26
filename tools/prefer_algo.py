"""Writes a user algorithm cache (<cache>.algo.json, to be layered over the shipped database with PLANER_HIP_TUNE_CACHE=<cache>) in
which every stored pick of w_layout FROM for a matching input shape is replaced by TO -- for A/B runs of one algorithm against
another under bench.py.    python tools/prefer_algo.py <cache> <from> <to> [H ...]      (H: only maps of these heights)"""
import ast, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cache, frm, to = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
hs = [int(a) for a in sys.argv[4:]]
d = json.load(open(os.path.join(root, "planer_amd", "tuned", "gfx950_cu256.algo.json")))
out = {"device": d["device"], "algo": {}, "streams": {}}
for k, v in d["algo"].items():
    sig = ast.literal_eval(k)
    if v == frm and (not hs or sig[1][2] in hs):
        out["algo"][k] = to
json.dump(out, open(cache + ".algo.json", "w"), indent=1)
print(len(out["algo"]), "picks", frm, "->", to)
