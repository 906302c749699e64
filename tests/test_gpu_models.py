"""GPU parity of the module-level drop-ins (Encoder / Decoder / AdaIN / stylize) against
the reference-generated golden vectors and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
RTOL = 1e-3   # north_star: within 1e-3 relative in fp32


def rel_err(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def models():
    from style_transfer_amd.generator import Decoder, Encoder
    from style_transfer_amd.transformer import AdaIN
    enc = Encoder(weights=R.make_encoder_state(0)).cuda().eval()
    dec = Decoder().cuda().eval()
    dec.load_state_dict(R.make_decoder_state(1))
    return enc, dec, AdaIN().cuda()


def test_state_dict_keys(models):
    enc, dec, ada = models
    assert list(enc.encode.state_dict().keys()) == list(R.make_encoder_state(0).keys())
    assert list(dec.state_dict().keys()) == list(R.make_decoder_state(1).keys())
    assert sum(p.numel() for p in dec.parameters()) == 3505219
    assert list(ada.state_dict().keys()) == ["content_eps", "style_eps"]


def test_encoder_decoder_golden(models, golden_dir):
    enc, dec, _ = models
    g = np.load(os.path.join(golden_dir, "generator.npz"))
    for name in ("a", "b"):
        shape = tuple(int(v) for v in g[f"{name}_shape"])
        x = torch.rand(*shape, generator=torch.Generator().manual_seed(int(g[f"{name}_seed"])))
        with torch.no_grad():
            f = enc(x.cuda())
            y = dec(torch.from_numpy(g[f"{name}_feat"]).cuda())
        assert f.shape == g[f"{name}_feat"].shape and y.shape == g[f"{name}_dec"].shape
        assert rel_err(f, torch.from_numpy(g[f"{name}_feat"])) < RTOL
        assert rel_err(y, torch.from_numpy(g[f"{name}_dec"])) < RTOL


def test_sympad_golden(golden_dir):
    from style_transfer_amd.generator import SymmetricPadding2D
    g = np.load(os.path.join(golden_dir, "sympad.npz"))
    x = torch.from_numpy(g["x"]).cuda()
    assert np.array_equal(SymmetricPadding2D((1, 1, 1, 1))(x).cpu().numpy(), g["pad1"])
    assert np.array_equal(SymmetricPadding2D((2, 2, 2, 2))(x).cpu().numpy(), g["pad2"])
    assert np.array_equal(SymmetricPadding2D((1, 3, 0, 6))(x).cpu().numpy(), g["pad_asym"])


@pytest.mark.parametrize("alpha", [1.0, 0.5, 1.5])
def test_stylize_vs_oracle(models, alpha):
    from style_transfer_amd.pipeline import stylize
    enc, dec, ada = models
    g = torch.Generator().manual_seed(77)
    c = torch.rand(2, 3, 64, 72, generator=g)
    s = torch.rand(2, 3, 48, 40, generator=g)
    ref = R.stylize(c, s, R.make_encoder_state(0), R.make_decoder_state(1), alpha=alpha)
    out = stylize(enc, dec, ada, c.cuda(), s.cuda(), alpha=alpha)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < RTOL


def test_cpu_tensor_fails_loudly(models):
    from style_transfer_amd._lib import StHipError
    enc, _, _ = models
    with pytest.raises(StHipError):
        enc(torch.rand(1, 3, 8, 8))
