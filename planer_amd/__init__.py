"""planer_amd -- planer's per-layer forward pass, native on MI355X (gfx950).

Drop-in for the hot path of Image-Py/planer (reference __init__.py, layer.py,
net.py, io.read_net): same `Net` / layer callables / json-flow IR /
`read_net()` / `core()` surface, but every operator is a hand-written HIP
kernel reached through a C ABI (include/planer_hip.h) via ctypes.  numpy +
ctypes only; no PyTorch, no cupy.

Importing this package does not touch the GPU and does not need the shared
library; the first device operation loads it and raises loudly if it is
missing -- there is no CPU fallback inside this package.
"""
from . import hip
from . import util  # noqa: F401  (tile / resize: tiled large-image inference, util.py:253-348)
from .hip import DeviceArray
from .io import from_graph, read_net
from .onnx_import import onnx2pla, read_onnx  # noqa: F401  (io.py:53-299; need the `onnx` package at call time)
from .layer import *  # noqa: F401,F403  (Conv2d, Dense, ..., layer_map, wrap)
from .layer import layer_map, prepare_conv_weights, prepare_winograd_weights, wrap
from .net import Net

# compatible with onnxruntime, as in the reference (__init__.py:7)
InferenceSession = read_net

backend = hip


def core(obj="hip", silent=False):
    """Backend switch with the reference's signature (__init__.py:22-38).

    The reference rebinds the `np` of its modules to any numpy-like module.
    This package has exactly one backend -- the HIP one -- so `core` accepts
    'hip' / `planer_amd.hip` (and returns it, as the reference returns the
    backend).  numpy is refused on purpose: the numpy path is the reference
    itself, not something this package re-implements or falls back to.
    """
    name = obj if isinstance(obj, str) else getattr(obj, "__name__", "")
    if name not in ("hip", "planer_amd.hip"):
        raise ValueError("planer_amd has a single backend, 'hip'; got %r. "
                         "Use the reference planer package for numpy/cupy." % (name,))
    if not silent:
        print("\nuser switch engine:", hip.__name__)
    return hip


def asnumpy(arr, **key):
    """__init__.py:42"""
    return hip.asnumpy(arr, **key)


def asarray(arr, **key):
    """__init__.py:44"""
    return hip.asarray(arr, **key)
