"""Assemble planer json/flow IR + its uint8 weight blob.

The on-disk/IR format is the reference's (io.py:8-34 reader, io.py:284-299
writer): a dict with keys `input`, `inits` ([name, shape, dtype]), `layers`
([name, kind, params]) and `flow` ([src | [srcs], [layer names], dst |
[dsts]]), plus a 1-D uint8 array holding the raw bytes of every init in
`inits` order (net.py:83-88).  `onnx` is not installed in this image, so the
model graphs the benchmarks need are emitted here directly, laid out the way
read_onnx would have produced them (BatchNorm pre-folded to a per-channel
affine with eps 1e-5, io.py:76-91; a trailing `return` layer, io.py:284-285).
"""
import hashlib
import json
import os
import zipfile

import numpy as np


class GraphBuilder:
    def __init__(self, inputs):
        self.inputs = list(inputs)
        self.inits, self.layers, self.flow, self._arrays = [], [], [], []
        self._names = set()

    # -- tensors ----------------------------------------------------------
    def init(self, name, array):
        """Register a weight; 0-d values are stored as 1 element (io.py:62-63)."""
        a = np.ascontiguousarray(array)
        if a.ndim == 0:
            a = a.reshape(1)
        assert name not in self._names, name
        self._names.add(name)
        self.inits.append([name, list(a.shape), str(a.dtype)])
        self._arrays.append(a)
        return name

    # -- ops --------------------------------------------------------------
    def op(self, kind, src, dst, name=None, **params):
        """Add one layer and the flow step that runs it."""
        name = name or "%s_%d" % (kind, len(self.layers))
        self.layers.append([name, kind, params])
        self.flow.append([src, [name], dst])
        return dst

    def finish(self, outputs):
        self.layers.append(["return", "return", {}])
        self.flow.append([list(outputs), ["return"], "plrst"])
        graph = {"input": self.inputs, "inits": self.inits,
                 "layers": self.layers, "flow": self.flow}
        blob = (np.concatenate([a.reshape(-1).view(np.uint8)
                                for a in self._arrays])
                if self._arrays else np.zeros(0, np.uint8))
        return graph, blob


def blob_sha256(blob):
    return hashlib.sha256(np.ascontiguousarray(blob).tobytes()).hexdigest()


def save_model(path, graph, blob, pla=False):
    """Write `<path>.json` + `<path>.npy`, or the `.pla` zip of both
    (same members as io.onnx2pla writes, io.py:289-299)."""
    base = os.path.split(path)[1]
    if pla:
        with zipfile.ZipFile(path + ".pla", "w") as z:
            z.writestr(base + ".json", json.dumps(graph))
            from io import BytesIO
            buf = BytesIO()
            np.save(buf, blob)
            z.writestr(base + ".npy", buf.getvalue())
    else:
        with open(path + ".json", "w") as f:
            json.dump(graph, f)
        np.save(path + ".npy", blob)
