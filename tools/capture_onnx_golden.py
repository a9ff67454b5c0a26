"""Build container only: run the REFERENCE's importer (io.read_onnx, /root/reference/planer/io.py:53-287)
on the stand-in protobuf objects of tests/onnx_standin.py and commit what it returns as
tests/golden/onnx_ir.json.  `onnx` is not installed, so a fake `onnx` module whose `load()` hands back the
stand-in model (and whose numpy_helper.to_array unwraps the stand-in tensors) is put in sys.modules first --
the reference code itself runs unmodified."""
import contextlib
import hashlib
import io
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HOME", "/tmp/planer_home")
os.makedirs(os.environ["HOME"], exist_ok=True)

from tests import onnx_standin as st  # noqa: E402

fake = types.ModuleType("onnx")
fake.numpy_helper = types.ModuleType("onnx.numpy_helper")
fake.numpy_helper.to_array = st.to_array
fake.load = lambda path: st.MODELS[os.path.basename(path)]()
sys.modules["onnx"], sys.modules["onnx.numpy_helper"] = fake, fake.numpy_helper

sys.path.insert(0, "/root/reference")
with contextlib.redirect_stdout(io.StringIO()):
    import planer  # noqa: E402
    from planer import io as ref_io  # noqa: E402

out = {}
for name in st.MODELS:
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        g, w = ref_io.read_onnx(name)
    if g == "lost":
        out[name] = {"lost": w.op_type, "printed": buf.getvalue()}
        continue
    out[name] = {"graph": json.loads(json.dumps(g)), "blob_len": int(w.size), "blob_sha256": hashlib.sha256(w.tobytes()).hexdigest()}
    if w.size < 4096:
        out[name]["blob"] = w.tolist()
with open(os.path.join(ROOT, "tests", "golden", "onnx_ir.json"), "w") as f:
    json.dump(out, f, indent=0)
print({k: (v.get("blob_len"), len(v.get("graph", {}).get("layers", []))) for k, v in out.items()})
