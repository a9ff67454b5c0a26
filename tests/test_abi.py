"""C-ABI checks that need no GPU: the library loads, exports every symbol
include/planer_hip.h declares, the ctypes table matches the header, and the
product refuses to run without a device (no CPU fallback)."""
import os
import re
import subprocess

import pytest

from planer_amd import _lib
from tests.conftest import ROOT

HEADER = os.path.join(ROOT, "include", "planer_hip.h")


def header_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pl_[a-z0-9_]+)\s*\(", text)))


def _built():
    return os.path.exists(_lib.LIB_PATH)


def test_header_and_ctypes_table_agree():
    assert header_functions() == sorted(_lib.SIGNATURES)


@pytest.mark.skipif(not _built(), reason="libplaner_hip.so not built")
def test_library_exports_every_declared_symbol():
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (pl_[a-z0-9_]+)$", out, flags=re.M))
    missing = [f for f in header_functions() if f not in exported]
    assert not missing, missing
    lib = _lib.load()            # declares argtypes for every entry: AttributeError if absent
    assert lib.pl_version() >= 100
    assert lib.pl_conv2d_num_configs() >= 4


@pytest.mark.skipif(not _built(), reason="libplaner_hip.so not built")
def test_no_device_means_loud_failure():
    import ctypes
    lib = _lib.load()
    n = ctypes.c_int(-1)
    rc = lib.pl_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible here")
    import planer_amd
    from planer_amd.irgen import customnet
    g, b = customnet.build()
    with pytest.raises(_lib.HipBackendError):
        planer_amd.from_graph(g, b)


def test_core_chooses_the_array_side_only(capsys):
    """core() returns the array module like the reference's (__init__.py:22-38): 'hip' = device arrays, numpy = host arrays
    in and out (the reference's import-time default, __init__.py:40) -- computed on HIP either way; anything else is refused."""
    import types
    import numpy
    import planer_amd
    try:
        assert planer_amd.core("hip", silent=True) is planer_amd.hip
        assert planer_amd.core(planer_amd.hip, silent=True) is planer_amd.hip
        assert planer_amd.core(numpy) is numpy and planer_amd.backend is numpy
        assert "user switch engine: numpy" in capsys.readouterr().out          # the reference's message (__init__.py:37)
        a = planer_amd.asarray([[1.0, 2.0]])                                   # no device involved: numpy's asarray
        assert isinstance(a, numpy.ndarray) and planer_amd.asnumpy(a) is a
        for bad in (types.ModuleType("cupy"), "numexpr", types.ModuleType("jax.numpy")):
            with pytest.raises(ValueError, match="reference"):
                planer_amd.core(bad)
    finally:
        planer_amd.core("hip", silent=True)


def test_product_never_imports_oracle_or_torch():
    pkg = os.path.join(ROOT, "planer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|torch)\b", src, flags=re.M), f


# the kinds of the reference's operator table (layer.py:262-281), names only
REFERENCE_KINDS = """add averagepool batchnorm cast clip concat const constantofshape conv convtranspose dense div equal erf
exp expand flatten gap gather greater greaterorequal hardsigmoid identity instancenormalization leakyrelu log logsoftmax
lstm matmul maxpool mul nonzero pad pow range reciprocal reducemax reducemean reducemin reducesum relu reshape resize
return scatternd shape sigmoid slice softmax split sqrt squeeze sub tanh tile topk transpose unsqueeze upsample
where""".split()


def test_operator_table_covers_every_reference_kind():
    from planer_amd import layer
    assert len(REFERENCE_KINDS) == 60
    assert [k for k in REFERENCE_KINDS if k not in layer.layer_map] == []
    assert layer.NOT_ON_DEVICE == []
    assert not any(f.__name__.startswith("missing_") for f in layer.layer_map.values())
    # the importer's op table only emits kinds the operator table has
    from planer_amd.onnx_import import OP_TABLE
    assert [kind for kind, _ in OP_TABLE.values() if kind not in layer.layer_map] == []


def test_plan_dispatch_table_is_current():
    """csrc/plan_dispatch.inc (the thunks pl_plan_build resolves call names with) is generated from the ctypes signature table:
    an ABI change without re-running tools/gen_plan_dispatch.py would leave a plan file calling through a stale prototype."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_plan_dispatch.py"), "--check"])
    assert r.returncode == 0, "run tools/gen_plan_dispatch.py and rebuild"


def test_stem_pool_shape_predicates_need_no_device():
    """The plan compiler asks the library which stem shapes the one-kernel stem + max-pool paths take (host functions, no GPU):
    3 channels, 7x7 / stride 2 / pad 3, Cout % 4 == 0 at any width; the form that reads the NCHW batch itself needs W % 4 == 0."""
    from planer_amd import q4
    para = dict(group=1, strides=[2, 2], dilations=[1, 1], pads=[3, 3, 3, 3])
    for w, packed, nchw in ((224, True, True), (160, True, True), (1000, True, True), (37, True, False), (231, True, False), (8, True, True)):
        assert q4.stem_pool_eligible((2, 3, 64, w), (64, 3, 7, 7), **para) is packed, w
        assert q4.stem_pool_nchw_eligible((2, 3, 64, w), (64, 3, 7, 7), **para) is nchw, w
    assert not q4.stem_pool_nchw_eligible((2, 3, 64, 64), (62, 3, 7, 7), **para)                     # Cout % 4
    assert not q4.stem_pool_nchw_eligible((2, 3, 64, 64), (64, 3, 7, 7), **dict(para, strides=[1, 1]))
    assert not q4.stem_pool_nchw_eligible((2, 4, 64, 64), (64, 4, 7, 7), **para)                     # 3 input channels only
    assert not q4.stem_pool_nchw_eligible((2, 3, 64, 64), (64, 3, 3, 3), **dict(para, pads=[1, 1, 1, 1]))
    n = __import__("ctypes").c_size_t()
    from planer_amd import _lib
    _lib.call("pl_conv2d_stem_nchw_filter_elems", 64, __import__("ctypes").byref(n))
    assert n.value == 48 * 64 * 4                                                                    # [48 k-quads][Cout][4]
