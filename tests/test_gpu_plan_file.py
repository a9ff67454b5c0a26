"""pl_plan_build / pl_plan_run / pl_plan_destroy on a real MI355X: a forward pass replayed from a plan file by a host that uses
NOTHING but ctypes and include/planer_hip.h's entry points (SURVEY 8(b) export list; the loop it replaces: net.py:37-72).

The file is written once by planer_amd.export.export_plan (Python, the plan compiler); the run side below never imports
planer_amd.net / layer / q4 -- it dlopens the library, builds the plan, writes the input into the plan's buffer, runs, reads the
output.  ResNet-18 batch 2 must reproduce the reference's logits (tests/golden/resnet18_b2.npz, 1e-4) and the Python host's
captured plan bit for bit (same kernels, same launch plans, same order)."""
import ctypes
import os

import numpy as np
import pytest

from tests.conftest import ROOT, RTOL, assert_close, load_golden

pytestmark = pytest.mark.gpu

LIB = os.path.join(ROOT, "planer_amd", "libplaner_hip.so")


def _bind():
    lib = ctypes.CDLL(LIB)
    P, I, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    lib.pl_last_error.restype = ctypes.c_char_p
    lib.pl_ctx_create.argtypes = [I, ctypes.POINTER(P)]
    lib.pl_ctx_destroy.argtypes = [P]
    lib.pl_plan_build.argtypes = [P, P, Z, ctypes.POINTER(P)]
    lib.pl_plan_info.argtypes = [P, ctypes.POINTER(I), ctypes.POINTER(I), ctypes.POINTER(Z), ctypes.POINTER(Z), ctypes.POINTER(I)]
    lib.pl_plan_tensor.argtypes = [P, I, I, ctypes.POINTER(P), ctypes.POINTER(Z), ctypes.POINTER(I), ctypes.POINTER(I), ctypes.POINTER(I)]
    lib.pl_plan_run.argtypes = [P, ctypes.POINTER(P), ctypes.POINTER(P)]
    lib.pl_plan_destroy.argtypes = [P]
    lib.pl_h2d.argtypes = [P, P, P, Z]
    lib.pl_d2h.argtypes = [P, P, P, Z]
    lib.pl_sync.argtypes = [P]
    lib.pl_alloc.argtypes = [P, Z, ctypes.POINTER(P)]
    lib.pl_free.argtypes = [P, P]
    return lib


def _ok(lib, rc):
    assert rc == 0, lib.pl_last_error().decode()


def _run_plan(lib, blob, inputs, through_pointers=False):
    """ctypes only: -> list of output arrays"""
    ctx = ctypes.c_void_p()
    _ok(lib, lib.pl_ctx_create(0, ctypes.byref(ctx)))
    plan = ctypes.c_void_p()
    buf = ctypes.create_string_buffer(blob, len(blob))
    _ok(lib, lib.pl_plan_build(ctx, buf, len(blob), ctypes.byref(plan)))
    n_in, n_out, ncalls = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    arena, consts = ctypes.c_size_t(), ctypes.c_size_t()
    _ok(lib, lib.pl_plan_info(plan, ctypes.byref(n_in), ctypes.byref(n_out), ctypes.byref(arena), ctypes.byref(consts), ctypes.byref(ncalls)))
    assert n_in.value == len(inputs) and ncalls.value > 0
    outs = []
    for rep in range(3):                                    # replays must be stable
        in_ptrs = (ctypes.c_void_p * n_in.value)()
        held = []
        for i, x in enumerate(inputs):
            ptr, nbytes = ctypes.c_void_p(), ctypes.c_size_t()
            dims = (ctypes.c_int * 8)()
            nd, dt = ctypes.c_int(), ctypes.c_int()
            _ok(lib, lib.pl_plan_tensor(plan, 0, i, ctypes.byref(ptr), ctypes.byref(nbytes), ctypes.byref(dt), ctypes.byref(nd), dims))
            assert nbytes.value == x.nbytes and tuple(dims[:nd.value]) == x.shape and dt.value == 0
            if through_pointers:                            # the caller's own device buffer, copied in by pl_plan_run
                mine = ctypes.c_void_p()
                _ok(lib, lib.pl_alloc(ctx, x.nbytes, ctypes.byref(mine)))
                _ok(lib, lib.pl_h2d(ctx, mine, x.ctypes.data, x.nbytes))
                in_ptrs[i] = mine.value
                held.append(mine)
            else:
                _ok(lib, lib.pl_h2d(ctx, ptr, x.ctypes.data, x.nbytes))
        _ok(lib, lib.pl_plan_run(plan, in_ptrs if through_pointers else None, None))
        _ok(lib, lib.pl_sync(ctx))
        got = []
        for i in range(n_out.value):
            ptr, nbytes = ctypes.c_void_p(), ctypes.c_size_t()
            dims = (ctypes.c_int * 8)()
            nd, dt = ctypes.c_int(), ctypes.c_int()
            _ok(lib, lib.pl_plan_tensor(plan, 1, i, ctypes.byref(ptr), ctypes.byref(nbytes), ctypes.byref(dt), ctypes.byref(nd), dims))
            y = np.empty(tuple(dims[:nd.value]), np.float32)
            assert y.nbytes == nbytes.value
            _ok(lib, lib.pl_d2h(ctx, y.ctypes.data, ptr, y.nbytes))
            got.append(y)
        for m in held:
            lib.pl_free(ctx, m)
        outs.append(got)
    for other in outs[1:]:
        for a, c in zip(outs[0], other):
            np.testing.assert_array_equal(a, c)
    _ok(lib, lib.pl_plan_destroy(plan))
    _ok(lib, lib.pl_ctx_destroy(ctx))
    return outs[0]


def test_resnet18_batch2_from_a_plan_file_with_ctypes_only(tmp_path):
    import planer_amd
    from planer_amd.export import export_plan
    from planer_amd.irgen import resnet18
    g, b = resnet18.build()
    x = resnet18.make_input(2)
    net = planer_amd.from_graph(g, b)
    want = net(x)
    path = tmp_path / "resnet18_b2.plplan"
    blob = export_plan(net, x, path=str(path))
    assert blob[:8] == b"PLPLAN1\0" and path.stat().st_size == len(blob)
    lib = _bind()
    for through in (False, True):
        got, = _run_plan(lib, open(path, "rb").read(), [x], through_pointers=through)
        assert got.shape == (2, 1000)
        assert_close(got, load_golden("resnet18_b2.npz")["logits"], RTOL, "plan file vs the reference's logits")
        np.testing.assert_array_equal(got, want)            # the Python host's plan: same kernels in the same order
    # a corrupted file is refused, not executed
    bad = bytearray(blob)
    bad[3] ^= 0xFF
    ctx, plan = ctypes.c_void_p(), ctypes.c_void_p()
    _ok(lib, lib.pl_ctx_create(0, ctypes.byref(ctx)))
    buf = ctypes.create_string_buffer(bytes(bad), len(bad))
    assert lib.pl_plan_build(ctx, buf, len(bad), ctypes.byref(plan)) != 0 and b"magic" in lib.pl_last_error()
    assert lib.pl_plan_build(ctx, buf, 40, ctypes.byref(plan)) != 0
    lib.pl_ctx_destroy(ctx)


def test_yolov3_and_customnet_plan_files(tmp_path):
    """Three outputs (YOLO-v3 at 160 px: upsample / concat routes, late residuals, fused 1x1 -> Winograd kernels) and the README's
    CustomNet (pointwise layers, NCHW program)."""
    import planer_amd
    from planer_amd.export import export_plan
    from planer_amd.irgen import customnet, yolov3
    lib = _bind()
    for mod, x in ((yolov3, yolov3.make_input(1, size=160)), (customnet, customnet.make_input(2))):
        g, b = mod.build()
        net = planer_amd.from_graph(g, b)
        want = net(x)
        want = list(want) if isinstance(want, tuple) else [want]
        got = _run_plan(lib, export_plan(net, x), [x])
        assert len(got) == len(want)
        for a, w in zip(got, want):
            assert_close(a, w, 1e-5, mod.__name__)


def test_hostile_plan_files_are_refused_not_read_past_the_end(tmp_path):
    """Round-5 advisor finding: lengths and offsets come from the file, and `p + n > end` / `offset + bytes > arena` /
    `(len + 7) / 8 * 8` wrap for values near 2^64.  A tensor record whose offset wraps, a call whose name length or whose host
    blob length is 2^64 - 1 (rounded up: 0), a header that promises more constants than the file holds: each must come back as
    PL_EINVAL with a message, never a crash or a read past the buffer."""
    import struct
    import planer_amd
    from planer_amd.export import export_plan
    from planer_amd.irgen import customnet
    lib = _bind()
    g, b = customnet.build()
    x = customnet.make_input(1)
    good = export_plan(planer_amd.from_graph(g, b), x, path=str(tmp_path / "c.plplan"))
    ctx = ctypes.c_void_p()
    _ok(lib, lib.pl_ctx_create(0, ctypes.byref(ctx)))

    def refused(blob, what):
        plan = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(blob), len(blob))
        rc = lib.pl_plan_build(ctx, buf, len(blob), ctypes.byref(plan))
        assert rc == 1 and what in lib.pl_last_error(), (rc, lib.pl_last_error())      # PL_EINVAL

    n_in, n_out = struct.unpack_from("<II", good, 24)
    bad = bytearray(good)
    struct.pack_into("<Q", bad, 40, 2 ** 64 - 8)                      # first tensor: offset + bytes wraps to a small number
    refused(bad, b"bad tensor record")
    first_call = 40 + (n_in + n_out) * 56
    bad = bytearray(good)
    struct.pack_into("<I", bad, first_call, 0xFFFFFFFF)               # a name as long as the address space
    refused(bad, b"bad call record")
    bad = bytearray(good)
    struct.pack_into("<Q", bad, 8, len(good) * 4)                     # more constants than the file has bytes
    refused(bad, b"constants")
    # a call whose host-blob argument claims 2^64 - 1 bytes (rounds up to 0 under the 8-byte padding)
    name = b"pl_memset"
    rec = struct.pack("<I", len(name)) + name + b"\0" * (-len(name) % 4) + struct.pack("<I", 4)
    rec += struct.pack("<IIQ", 6, 0, 0)                               # ctx
    rec += struct.pack("<IIQ", 5, 0, 2 ** 64 - 1)                     # host blob, hostile length
    tiny = b"PLPLAN1\0" + struct.pack("<QQIIII", 0, 64, 0, 0, 1, 0) + rec
    refused(tiny, b"bad argument")
    lib.pl_ctx_destroy(ctx)
