"""GPU: what would mixed Winograd tiles -- F(4,3) x F(3,3) on maps that are not multiples of 4 (14 = 4+4+3+3, 7 = 4+3) -- buy the
layer3 / layer4 GEMMs of ResNet-18 at batch 32?  Today 14x14 pads to 16x16 and 7x7 to 8x8: 36 frequencies x (N x 16 | N x 4)
tile columns, 31 % of them padding.  Mixed tiles give four tile classes (6x6, 6x5, 5x6, 5x5 frequencies = 121 GEMM groups in
all) with N x 4 (layer3) or N x 1 (layer4) columns each and no padding tile: 0.84 of the multiplies, 3.4x the filter bytes.
Only the GEMM is timed here -- it is 2/3 of the conv and the part the padding costs -- as the grouped 1x1 convolution the
Winograd pipeline runs (pl_conv2d_q4_f32, autotuned launch plan, random operands): today's shape against the mixed-tile shape.
A transform for the mixed tiles moves 0.84x the bytes of today's at best, so the conv gains at most what this prints + 16 % of
its transform time."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import planer_amd as pa
from planer_amd import hip, q4, _lib
ctx = hip.context()
rng = np.random.default_rng(0)
blocker = hip.empty((256 << 20,), np.float32, ctx)


def time_grouped(groups, c, cols, label):
    """36 (or 121) GEMMs of (c x c) . (c x cols) as one grouped 1x1 conv on a (1, groups*c, cols, 1) channel-quad tensor"""
    x = q4.to_q4(pa.asarray(rng.standard_normal((1, groups * c, cols, 1)).astype(np.float32)))
    k = q4.prepare_q4_weights(pa.asarray((rng.standard_normal((groups * c, c, 1, 1)) * 0.05).astype(np.float32)), groups)
    run = lambda: q4.ConvQ4(x, k, group=groups, w_layout=2)
    for _ in range(3):
        run()
    best = 1e9
    for _ in range(5):
        for _ in range(4):
            _lib.call("pl_memset", ctx.handle, blocker.ptr, 0, blocker.nbytes)
        e0 = hip.Event(ctx).record()
        for _ in range(10):
            run()
        e1 = hip.Event(ctx).record()
        best = min(best, e0.elapsed_ms(e1) / 10)
    flops = 2.0 * groups * c * c * cols
    print("%-46s %3d groups x %4d columns: %6.2f us  %5.1f TFLOP/s executed  [%s]" % (label, groups, cols, best * 1e3, flops / best / 1e9, ctx.last_conv_plan()))
    return best * 1e3


n = 32
for layer, c, tiles_now, tiles_mixed in (("layer3 (256 ch, 14x14)", 256, 16, 4), ("layer4 (512 ch, 7x7)", 512, 4, 1)):
    a = time_grouped(36, c, n * tiles_now, layer + " today: F(4x4) padded")
    b = time_grouped(121, c, n * tiles_mixed, layer + " mixed F(4,3) x F(3,3)")
    print("   -> %.2f us per conv (%.0f %%); filter bytes %.1f -> %.1f MB" % (a - b, 100 * (a - b) / a, 36 * c * c * 4 / 1e6, 121 * c * c * 4 / 1e6))
