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
    """Backend switch with the reference's signature (__init__.py:22-38): returns the array module.

    The reference rebinds the `np` of its modules to any numpy-like module -- the module decides what KIND OF ARRAY the
    package hands out and takes; the arithmetic follows from that.  Here the arithmetic is always the HIP library (this
    package holds no CPU implementation of any operator and never falls back to one), so `core` chooses the array side only:

      core('hip') / core(planer_amd.hip)   device arrays: `asarray` uploads, layers and nets return DeviceArrays
      core(numpy)                          host arrays, the reference's import-time default (__init__.py:40): `asarray` /
                                           `asnumpy` are numpy's, `net(x)` and every layer callable take ndarrays and return
                                           ndarrays -- computed on the GPU (uploaded and fetched per call, net.py:96-100)

    A script written against the reference that calls `planer.core(numpy)` (or never calls core) therefore runs unchanged.
    Anything else (cupy, a numpy clone) is refused: use the reference package for those backends.
    """
    global backend
    name = obj if isinstance(obj, str) else getattr(obj, "__name__", "")
    if name in ("hip", "planer_amd.hip"):
        backend = hip
    elif name == "numpy":
        import numpy
        backend = numpy
    else:
        raise ValueError("planer_amd computes on HIP only; core() takes 'hip' (device arrays) or numpy (host arrays in and "
                         "out, still computed on the GPU), got %r.  For cupy or another numpy-like backend use the reference "
                         "planer package." % (name,))
    if not silent:
        print("\nuser switch engine:", backend.__name__)
    return backend


def asnumpy(arr, **key):
    """__init__.py:42"""
    return hip.asnumpy(arr, **key)


def asarray(arr, **key):
    """__init__.py:44: the array type of the current backend -- a DeviceArray under core('hip'), an ndarray under core(numpy)"""
    if backend is hip:
        return hip.asarray(arr, **key)
    import numpy
    return arr.get() if isinstance(arr, DeviceArray) else numpy.asarray(arr, **{k: v for k, v in key.items() if k != "ctx"})
