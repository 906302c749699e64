"""GPU parity of the sub-pixel form of the upsample layers' forward (st_conv3x3_up2_fwd, csrc/conv3x3.hip): four
2 x 2 convolutions over the low-resolution input with pre-summed filter rows / columns, against the direct 3 x 3
convolution over the nearest-upsampled, replicate-padded input (generator.py:92-93, 105-106, 112-113).  Same products
regrouped: agreement to fp32 summation order (tolerance 5e-6 of the output's magnitude at K = 2304), and against the float64
reference to the same 1e-5 bar as the direct kernel."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from style_transfer_amd import ops as o
    return o


CASES = [
    # n, cin, cout, h, w (source dims)
    (2, 64, 64, 32, 32),
    (1, 128, 128, 17, 23),      # ragged source tile both ways
    (1, 256, 256, 8, 40),
    (3, 16, 8, 5, 33),          # one chunk, channel tail in the tile
    (1, 20, 70, 9, 9),          # cin not a multiple of the K chunk, cout not of the tile
    (2, 64, 64, 128, 128),
]


@pytest.mark.parametrize("variant", ["0", "1"])
@pytest.mark.parametrize("n,cin,cout,h,w", CASES)
@pytest.mark.parametrize("relu", [True, False])
def test_up2_matches_direct_and_float64(ops, monkeypatch, n, cin, cout, h, w, relu, variant):
    monkeypatch.setenv("ST_UP2_VARIANT", variant)             # both tile shapes of the kernel
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(F.pad(F.interpolate(x.double(), scale_factor=2, mode="nearest"), (1, 1, 1, 1), mode="replicate"),
                   wt.double(), b.double())
    if relu:
        ref = torch.relu(ref)
    xd, wd, bd = x.cuda(), wt.cuda(), b.cuda()
    direct = ops.conv3x3(xd, ops.pack_conv3x3_weights(wd, False, "f32"), bd, cout, True, "replicate", relu)
    wp = ops.pack_conv3x3_weights(wd, False, "f32up2")
    assert ops.conv_family(wp, cout, cin) == "f32up2"
    got = ops.conv3x3(xd, wp, bd, cout, True, "replicate", relu)
    scale = float(ref.abs().max())
    assert got.shape == direct.shape == ref.shape
    assert float((got - direct).abs().max()) <= 5e-6 * scale
    assert float((got.double().cpu() - ref).abs().max()) <= 1e-5 * scale


def test_decoder_forward_only_calls_take_the_sub_pixel_form_and_are_as_close_to_float64():
    """Decoder.forward without gradients (inference, eval plots) runs the three upsample layers in the sub-pixel form;
    its distance to the float64 oracle is that of the direct form (both far inside the 1e-3 bar).  Under autograd the
    forward is the same sub-pixel form (round 3; backward = the direct form's adjoint)."""
    from oracle import ref_ops as R
    from style_transfer_amd.generator import Decoder, _PackedWeightCache
    dec_sd = R.make_decoder_state(1)
    dec = Decoder().cuda()
    dec.load_state_dict(dec_sd)
    x0 = torch.relu(torch.randn(2, 512, 8, 12, generator=torch.Generator().manual_seed(2)))
    ref = R.decoder_forward(x0.double(), {k: v.double() for k, v in dec_sd.items()})

    def run(up2):
        _PackedWeightCache.UP2 = up2
        dec._packed.clear()
        with torch.no_grad():
            y = dec(x0.cuda())
        fams = {k[2] for k in dec._packed._c}
        return y, fams
    try:
        y_direct, fam_direct = run(False)
        y_sub, fam_sub = run(True)
        x = x0.cuda().requires_grad_(True)
        y_grad = dec(x)                                   # autograd path: the same sub-pixel forward
    finally:
        _PackedWeightCache.UP2 = True
        dec._packed.clear()
    assert "f32up2" in fam_sub and "f32up2" not in fam_direct
    scale = float(ref.abs().max())
    e_direct = float((y_direct.double().cpu() - ref).abs().max()) / scale
    e_sub = float((y_sub.double().cpu() - ref).abs().max()) / scale
    print(f"decoder forward vs float64: direct {e_direct:.2e}, sub-pixel {e_sub:.2e}")
    assert e_direct <= 1e-5 and e_sub <= 1e-5
    assert torch.equal(y_grad.detach(), y_sub)
