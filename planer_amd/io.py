"""Model files -> `Net` on the GPU (the reference's io.read_net, io.py:8-34).

Formats are the reference's own: a `.pla` zip holding `<base>.json` +
`<base>.npy`, or the two files side by side.  The json carries `input`,
`inits`, `layers`, `flow`; the npy is a 1-D uint8 array with every init's raw
bytes back to back (io.py:286, net.py:83-88).  A lone `.onnx` file goes through
planer_amd.onnx_import.read_onnx (io.py:24-27), which needs the `onnx` package.
"""
import json
import os
import zipfile
from io import BytesIO

import numpy

from .net import Net


def _load_pair(path):
    if os.path.exists(path + ".pla"):
        with zipfile.ZipFile(path + ".pla") as z:
            base = os.path.split(path)[1]
            graph = json.loads(z.read(base + ".json"))
            blob = numpy.load(BytesIO(z.read(base + ".npy")))
        return graph, blob
    if os.path.exists(path + ".json"):
        with open(path + ".json") as f:
            graph = json.load(f)
        return graph, numpy.load(path + ".npy")
    return None, None


def read_net(path, debug=False, ctx=None, comm=None):
    """io.read_net.  Missing model: prints and returns None like the
    reference (io.py:30-31).

    With `comm` (planer_amd.dist.Communicator) only rank 0 needs the weight
    file contents: it uploads the blob and one RCCL broadcast over xGMI fills
    every other rank's device copy.
    """
    path = path.replace(".onnx", "")
    graph, blob = _load_pair(path)
    if graph is None and os.path.exists(path + ".onnx"):           # io.py:24-27
        from .onnx_import import read_onnx
        graph, blob = read_onnx(path + ".onnx")
        if graph == "lost":
            return blob                                             # the node nobody knows (io.py:26)
    if graph is None:
        return print("model %s not found!" % path)
    net = Net(ctx)
    net.load_json(graph["input"], graph["inits"], graph["layers"], graph["flow"], debug)
    if comm is None or comm.world == 1:
        net.load_weights(blob)
    else:
        comm.load_weights(net, blob if comm.rank == 0 else None)
    return net


def from_graph(graph, blob, ctx=None):
    """Build a Net from an in-memory (graph dict, uint8 blob) pair."""
    net = Net(ctx)
    net.load_json(graph["input"], graph["inits"], graph["layers"], graph["flow"])
    if blob is not None:
        net.load_weights(blob)
    return net
