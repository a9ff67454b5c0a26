"""The fully fused Winograd F(4x4,3x3) kernel (csrc/conv_wf4_kernel.h, ConvQ4 w_layout 9) on a real MI355X against the
oracle (util.conv_for, util.py:17-44, + layer.BatchNorm / Add / ReLU / LeakyReLU) to 1e-4 of max|ref|, over every block
geometry (56 / 28 / 14 / 7 / odd maps, several images per block), channel counts that do not fill the 64-channel
block or the 4-channel quad chunk loop, and every fused tail."""
import numpy as np
import pytest

from tests.conftest import RTOL, assert_close
from tests.test_gpu_wino_chain import TAILS, _act, _operands, _oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import planer_amd
    planer_amd.hip.context()
    return planer_amd


SHAPES = [(2, 64, 56, 56, 64), (3, 128, 28, 28, 128), (5, 64, 14, 14, 256), (9, 32, 7, 7, 512), (1, 4, 1, 1, 4),
          (2, 8, 13, 13, 12), (3, 16, 26, 26, 32), (1, 16, 5, 9, 16), (2, 12, 4, 4, 72), (1, 32, 52, 52, 16),
          (5, 20, 3, 17, 24), (1, 8, 70, 66, 8), (33, 4, 2, 2, 4),
          # 28-pixel maps at a batch where fewer tile rows x more images per block need fewer blocks (ResNet-18's layer2 in
          # the shipped throughput plans: 1 tile row x 4 images x 8 columns): several images AND several row blocks per block grid
          (32, 8, 28, 28, 72), (64, 4, 27, 26, 8)]


def _patch_cells(h, w, tiles=32):
    """16-byte cells of one block's input patch (wf4_launch's geometry: `tiles` = NB images x BR x BC tiles)."""
    th, tw = -(-h // 4), -(-w // 4)
    bc = 1
    while bc < tw and bc < 16:
        bc *= 2
    br = 1
    while br < th and br * bc < tiles:
        br *= 2
    return (tiles // (br * bc)) * (4 * br + 2) * 4 * (bc + 1)


@pytest.mark.parametrize("half", ["1", "0"], ids=["16-tile blocks", "32-tile blocks"])
@pytest.mark.parametrize("shape", SHAPES, ids=["x".join(map(str, s)) for s in SHAPES])
def test_fused_f4x4_conv_vs_oracle(pa, shape, half, monkeypatch):
    from planer_amd import q4
    monkeypatch.setenv("PLANER_HIP_EXPERIMENT", "wf4_half=" + half)
    n, cin, h, w, cout = shape
    rng = np.random.default_rng(11 + sum(shape))
    for tail in TAILS:
        host, dev = _operands(pa, rng, n, cin, h, w, cout, tail)
        u = q4.prepare_wf4_q4_weights(dev["k"])
        kw = dict(pads=(1, 1, 1, 1), act=_act(tail), alpha=0.1, w_layout=9)
        if _patch_cells(h, w, 16 if half == "1" else 32) > (512 if half == "1" else 1024):
            # one-tile maps put 32 images into a block: more patch cells than the kernel's LDS buffers hold -> refused
            # (the plan compiler then keeps another algorithm), never computed wrongly
            with pytest.raises(NotImplementedError):
                q4.ConvQ4(dev["xq"], u, dev["b"], dev["scale"], dev["shift"], dev["resq"], **kw)
            continue
        yq = q4.ConvQ4(dev["xq"], u, dev["b"], dev["scale"], dev["shift"], dev["resq"], **kw)
        assert q4.logical_shape(yq) == (n, cout, h, w)
        assert_close(q4.from_q4(yq).get(), _oracle(host, tail), 3e-5, "%s %s [%s]" % (shape, tail, pa.hip.context().last_conv_plan()))


def test_28_pixel_maps_take_the_one_row_four_image_block(pa, monkeypatch):
    """Round-5 advisor finding: the block shape the shipped throughput database runs on layer2 (NB images x BR = 1 tile row x BC = 8
    tile columns, 7 row blocks: `4x1x8` with 32-tile blocks, `2x1x8` with the 16-tile blocks of round 6) must be what these shapes
    really launch -- the parity above then covers it; both block sizes are run."""
    from planer_amd import q4
    rng = np.random.default_rng(28)
    for n, cin, h, w, cout, half in ((32, 8, 28, 28, 72, "1"), (64, 4, 27, 26, 8, "1"), (32, 8, 28, 28, 72, "0")):
        monkeypatch.setenv("PLANER_HIP_EXPERIMENT", "wf4_half=" + half)
        x = q4.to_q4(pa.asarray(rng.standard_normal((n, cin, h, w)).astype(np.float32)))
        k = pa.asarray((rng.standard_normal((cout, cin, 3, 3)) * 0.1).astype(np.float32))
        q4.ConvQ4(x, q4.prepare_wf4_q4_weights(k), pads=(1, 1, 1, 1), w_layout=9)
        plan = pa.hip.context().last_conv_plan()
        nb = 2 if "16tiles" in plan else 4              # (round 6: blocks of 16 tiles on four waves by default; wf4_half=0: 32 on eight)
        assert "(%dx1x8)" % nb in plan, plan
        assert "blocks=%d" % ((n // nb) * 7 * ((cout + 63) // 64)) in plan, plan


PACKED = [(8, 64, 56, 56, 64), (16, 16, 56, 56, 24), (4, 8, 48, 48, 8), (8, 8, 54, 55, 12), (16, 4, 53, 56, 8), (5, 8, 20, 20, 8)]


@pytest.mark.parametrize("shape", PACKED, ids=["x".join(map(str, s)) for s in PACKED])
def test_packed_blocks_equal_plain_blocks_bit_for_bit(pa, shape, monkeypatch):
    """Round 5: where a block has spare slot columns (14 tile columns in a 16-column block) and the batch is a multiple of G = tw / sc + 1,
    the spare slots of G - 1 images' blocks carry the G-th image's tiles (ResNet-18 layer1 at batch 32: 196 instead of 224
    workgroups).  Same arithmetic per tile: bit-identical to the plain blocks (PLANER_HIP_EXPERIMENT=wf4_pack=0) and within
    the conv tolerance of the oracle, with every tail."""
    from planer_amd import q4
    n, cin, h, w, cout = shape
    rng = np.random.default_rng(3 + sum(shape))
    for tail in TAILS:
        host, dev = _operands(pa, rng, n, cin, h, w, cout, tail)
        u = q4.prepare_wf4_q4_weights(dev["k"])
        kw = dict(pads=(1, 1, 1, 1), act=_act(tail), alpha=0.1, w_layout=9)
        monkeypatch.delenv("PLANER_HIP_EXPERIMENT", raising=False)
        packed = q4.ConvQ4(dev["xq"], u, dev["b"], dev["scale"], dev["shift"], dev["resq"], **kw)
        plan = pa.hip.context().last_conv_plan()
        monkeypatch.setenv("PLANER_HIP_EXPERIMENT", "wf4_pack=0")
        plain = q4.ConvQ4(dev["xq"], u, dev["b"], dev["scale"], dev["shift"], dev["resq"], **kw)
        assert "packed" not in pa.hip.context().last_conv_plan()
        if shape != PACKED[-1]:
            assert "packed" in plan, plan                       # (the last shape has no spare columns: 5 tile columns in 8 -> 3, 5 % 3)
        np.testing.assert_array_equal(packed.get(), plain.get(), err_msg="%s %s [%s]" % (shape, tail, plan))
        assert_close(q4.from_q4(packed).get(), _oracle(host, tail), 3e-5, "%s %s [%s]" % (shape, tail, plan))


def test_fused_f4x4_is_linear_and_shard_independent(pa):
    """Size-independent properties at a real layer size (batch 32, 64 channels, 56x56): conv(a x1 + x2) = a conv(x1) + conv(x2),
    and row i of the batch equals the same image run alone."""
    from planer_amd import q4
    rng = np.random.default_rng(5)
    k = pa.asarray((rng.standard_normal((64, 64, 3, 3)) * (2.0 / 576) ** 0.5).astype(np.float32))
    u = q4.prepare_wf4_q4_weights(k)
    x1 = rng.standard_normal((32, 64, 56, 56)).astype(np.float32)
    x2 = rng.standard_normal((32, 64, 56, 56)).astype(np.float32)

    def conv(x):
        return q4.from_q4(q4.ConvQ4(q4.to_q4(pa.asarray(x)), u, pads=(1, 1, 1, 1), w_layout=9)).get()
    y1, y2, y3 = conv(x1), conv(x2), conv(2.0 * x1 + x2)
    assert_close(y3, 2.0 * y1 + y2, 2e-5, "linearity")
    np.testing.assert_array_equal(conv(x1[7:8]), y1[7:8])
    # and against the staged pipeline (same algorithm, other kernels)
    y7 = q4.from_q4(q4.ConvQ4(q4.to_q4(pa.asarray(x1)), q4.prepare_winograd4_q4_weights(k), pads=(1, 1, 1, 1), w_layout=7)).get()
    assert_close(y1, y7, 2e-5, "fused vs staged F(4x4,3x3)")


def test_fused_f4x4_through_the_plan_compiler(pa):
    """ResNet-18 with every eligible conv forced onto the fused kernel: logits against the oracle."""
    import planer_amd
    from oracle import planer_np as onp
    from planer_amd.irgen import resnet18
    g, b = resnet18.build()
    x = resnet18.make_input(4)
    net = planer_amd.from_graph(g, b)
    net.force_algo, net.streams = 9, "1x1"
    y = net(planer_amd.asarray(x.copy())).get()
    used = [a["w_layout"] for a in net.compile(planer_amd.asarray(x)).algos]
    assert used.count(9) == 13, used
    ref = onp.OracleNet()
    ref.load_json(g["input"], g["inits"], g["layers"], g["flow"])
    ref.load_weights(b)
    assert_close(y, ref(x.copy()), RTOL, "resnet18 with the fused F(4x4,3x3) kernel")
