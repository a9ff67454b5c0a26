"""Model files -> `Net` on the GPU (the reference's io.read_net, io.py:8-34).

Formats are the reference's own: a `.pla` zip holding `<base>.json` +
`<base>.npy`, or the two files side by side.  The json carries `input`,
`inits`, `layers`, `flow`; the npy is a 1-D uint8 array with every init's raw
bytes back to back (io.py:286, net.py:83-88).  `.onnx` import needs the `onnx`
package (io.py:53-54), which this image does not have; graphs are produced by
planer_amd.irgen or by the reference's own onnx2pla elsewhere.
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
    if graph is None:
        if os.path.exists(path + ".onnx"):
            raise NotImplementedError("reading .onnx needs the `onnx` package (io.py:53-54); "
                                      "convert with onnx2pla first")
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
