"""The decoder chain's bf16 data gradient with the pad / upsample adjoint in the GEMM epilogue (conv3x3_bf16_kernel
EPI = 1 / 2 + csrc/dgrad_band.hip) against the form it replaces (padded-domain GEMM into an fp32 workspace +
dgrad_fold8_kernel; ST_NO_FUSED_FOLD=1), through st_conv3x3_dgrad_bf16_g8_chain on the same operands.

Every fp32 value is produced by the same MFMA sequence and summed in the same order, so the G8 and the pixel-major
bf16 results must be IDENTICAL; the bias partial sums are summed in another fixed order (fp32 rounding).
Reference semantics: the adjoint of SymmetricPadding2D((1,1,1,1)) (src/models/generator.py:19-39) and of
UpsamplingNearest2d(2) (:92, :105, :112) on the decoder's backward, pinned to the reference's autograd by
tests/test_gpu_backward.py / test_gpu_decoder_chain_g8.py through the per-layer path."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(dy, wpt, cin, up, gate):
    from style_transfer_amd import ops
    out = []
    for fused in (False, True):
        if fused:
            os.environ.pop("ST_NO_FUSED_FOLD", None)
        else:
            os.environ["ST_NO_FUSED_FOLD"] = "1"
        try:
            out.append(ops.conv3x3_dgrad_g8_chain(dy, wpt, cin, up, gate, want_f32=False, want_g8=True, want_g16=True,
                                                  want_bias=True))
        finally:
            os.environ.pop("ST_NO_FUSED_FOLD", None)
    torch.cuda.synchronize()
    return out


# (n, k = the layer's output channels, c = its input channels, h, w of the conv input, upsample, dy form, gate form)
CASES = [
    (2, 3, 64, 32, 64, False, "f32", "f32"),       # the top layer: fp32 dy of 3 channels, fp32 gate
    (2, 3, 64, 64, 96, False, "f32", "g8"),        # the top layer with its input handed over in G8 (TOP_G8)
    (1, 3, 64, 40, 32, False, "f32", "g8"),        # ... ragged tile rows
    (2, 64, 64, 16, 64, True, "g8", "g8"),         # 64 -> 64 under the upsample
    (1, 64, 128, 8, 32, False, "g8", "g8"),        # 128 -> 64
    (2, 128, 128, 16, 32, True, "g8", "g8"),       # 128 -> 128 under the upsample, one tile column
    (1, 256, 256, 4, 32, False, "g8", "g8"),       # smallest source grid (4 rows)
    (1, 256, 256, 8, 32, True, "g8", None),        # no gate
    (3, 128, 256, 20, 96, False, "g8", "g8"),      # ragged tile rows (20 = 16 + 4), three tile columns, odd batch
    (1, 64, 64, 40, 160, True, "g8", "g8"),        # ragged rows under the upsample (40 = 32 + 8), five tile columns
    (1, 64, 64, 64, 64, False, "g8", "f32"),       # two full tile rows of the 64-channel tile
]


@pytest.mark.parametrize("n,k,c,h,w,up,dyf,gatef", CASES)
def test_fused_fold_equals_workspace_fold(n, k, c, h, w, up, dyf, gatef):
    from style_transfer_amd import ops
    g = torch.Generator().manual_seed(h * 1000 + w + k)
    wt = torch.randn(k, c, 3, 3, generator=g) * (2.0 / (9 * k)) ** 0.5      # the forward layer's OIHW weight
    wpt = ops.pack_conv3x3_weights(wt.cuda(), transpose_flip=True, compute="bf16")
    dy = torch.randn(n, k, h, w, generator=g)
    dy = dy * (torch.rand(n, k, h, w, generator=g) > 0.3)                   # a gated gradient: many exact zeros
    dy_dev = ops.to_g8(dy.cuda()) if dyf == "g8" else dy.cuda().contiguous()
    s = 2 if up else 1
    act = torch.relu(torch.randn(n, c, h // s, w // s, generator=g))
    gate = None if gatef is None else (ops.to_g8(act.cuda()) if gatef == "g8" else act.cuda().contiguous())
    (_, g8a, g16a, ba), (_, g8b, g16b, bb) = _both(dy_dev, wpt, c, up, gate)
    assert float(g16a.float().abs().max()) > 0
    # interior AND border pixels, both hand-over forms: identical values
    assert torch.equal(g16b.float(), g16a.float()), (g16b.float() - g16a.float()).abs().max()
    assert torch.equal(ops.from_g8(g8b), ops.from_g8(g8a))
    assert torch.equal(ops.from_g8(g8b), g16b.float())
    # the two forms' bias partial sums add up to the same totals (another fixed order)
    sa, sb = ba.sum(0), bb.sum(0)
    torch.testing.assert_close(sb, sa, rtol=2e-5, atol=2e-5 * float(sa.abs().max()) + 1e-6)
    # and equal the sum of the (rounded-before? no: fp32) gated gradient: check against the bf16 result loosely
    ref = g16a.float().sum(dim=(0, 2, 3))
    torch.testing.assert_close(sb, ref, rtol=2e-2, atol=2e-2 * float(ref.abs().max()) + 1e-3)


@pytest.mark.parametrize("n,k,c,h,w,up", [(2, 256, 512, 8, 32, False), (1, 256, 512, 64, 64, False), (1, 64, 64, 16, 64, True)])
def test_fused_fold_fp32_result_at_the_bottom_of_the_chain(n, k, c, h, w, up):
    """The bottom layer hands fp32 NCHW back to AdaIN (no gate, no bf16 form): same bits from both forms."""
    from style_transfer_amd import ops
    g = torch.Generator().manual_seed(h * 1000 + w + k)
    wt = torch.randn(k, c, 3, 3, generator=g) * (2.0 / (9 * k)) ** 0.5
    wpt = ops.pack_conv3x3_weights(wt.cuda(), transpose_flip=True, compute="bf16")
    dy = ops.to_g8((torch.randn(n, k, h, w, generator=g) * (torch.rand(n, k, h, w, generator=g) > 0.3)).cuda())
    out = []
    for fused in (False, True):
        if not fused:
            os.environ["ST_NO_FUSED_FOLD"] = "1"
        try:
            out.append(ops.conv3x3_dgrad_g8_chain(dy, wpt, c, up, None, want_f32=True, want_g8=False, want_g16=False,
                                                  want_bias=False)[0])
        finally:
            os.environ.pop("ST_NO_FUSED_FOLD", None)
    torch.cuda.synchronize()
    assert float(out[0].abs().max()) > 0 and torch.equal(out[1], out[0])


def test_bias_slab_count_covers_both_forms():
    from style_transfer_amd import _lib
    lib = _lib.load()
    for (n, cin, h, w, up) in [(8, 64, 512, 512, 0), (8, 64, 512, 512, 1), (8, 256, 128, 128, 1), (2, 128, 20, 96, 0)]:
        rows = lib.st_conv3x3_dgrad_chain_bias_slabs(n, cin, h, w, up)
        assert rows >= n * lib.st_conv3x3_dgrad_fold8_bias_slabs(h, w, up) > 0
