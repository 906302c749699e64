"""Parity at BASELINE's full sizes (configs[1] 512x512 batch 16, configs[4] 2048x2048), where the CPU oracle
cannot run the whole tensor in seconds: crop-consistency against the oracle (a conv stack's interior does not
depend on pixels outside its receptive field) and size-independent properties of AdaIN."""
import pytest
import torch

from oracle import ref_ops as R

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def models():
    from style_transfer_amd.generator import Decoder, Encoder
    enc_sd, dec_sd = R.make_encoder_state(0), R.make_decoder_state(1)
    enc = Encoder(weights=enc_sd).cuda().eval()
    dec = Decoder().cuda().eval()
    dec.load_state_dict(dec_sd)
    return enc, dec, enc_sd, dec_sd


@pytest.mark.parametrize("side,batch", [(2048, 1), (512, 16)])
def test_encoder_full_size_matches_oracle_on_crops(models, side, batch):
    """relu4_1 features of the full image vs the oracle on 320x320 corner crops.  A feature at (i, j) depends on
    input pixels within < 64 px of (8i, 8j) (three pools, nine 3x3 convs), so the 32x32 features nearest the image
    corner of a crop are exact; the image-border side exercises the edge-replicate padding at full size."""
    enc, _, enc_sd, _ = models
    g = torch.Generator().manual_seed(side)
    x = torch.rand(batch, 3, side, side, generator=g)
    with torch.no_grad():
        f = enc(x.cuda())
    torch.cuda.synchronize()
    assert f.shape == (batch, 512, side // 8, side // 8)
    n = batch - 1
    tl = R.encoder_forward(x[n:n + 1, :, :320, :320], enc_sd)
    br = R.encoder_forward(x[n:n + 1, :, -320:, -320:], enc_sd)
    assert rel_err(f[n:n + 1, :, :32, :32], tl[:, :, :32, :32]) < 1e-3
    assert rel_err(f[n:n + 1, :, -32:, -32:], br[:, :, -32:, -32:]) < 1e-3


def test_decoder_full_size_matches_oracle_on_crops(models):
    """Decoder 256x256 features -> 2048x2048 image vs the oracle on a 48x48 feature corner (output pixels more than
    ~6 feature cells away from the crop's cut edges are exact)."""
    _, dec, _, dec_sd = models
    g = torch.Generator().manual_seed(7)
    t = torch.rand(1, 512, 256, 256, generator=g)
    with torch.no_grad():
        y = dec(t.cuda())
    torch.cuda.synchronize()
    assert y.shape == (1, 3, 2048, 2048)
    tl = R.decoder_forward(t[:, :, :48, :48], dec_sd)
    br = R.decoder_forward(t[:, :, -48:, -48:], dec_sd)
    assert rel_err(y[:, :, :320, :320], tl[:, :, :320, :320]) < 1e-3
    assert rel_err(y[:, :, -320:, -320:], br[:, :, -320:, -320:]) < 1e-3


@pytest.mark.parametrize("alpha", [0.5, 1.0, 1.5])
def test_adain_config5_planes(alpha):
    """AdaIN on 256x256 feature planes (a 2048x2048 image; 256 KB per plane: the streaming kernel) for the alpha
    sweep of configs[4], against the oracle on the full tensor, plus the statistics property: at alpha = 1 every
    output plane has the style plane's mean and (unbiased, eps-shifted) standard deviation."""
    from style_transfer_amd import ops
    g = torch.Generator().manual_seed(11)
    c = torch.randn(1, 512, 256, 256, generator=g) * 2 + 1
    s = torch.randn(1, 512, 256, 240, generator=g) * 0.5 - 3       # style plane of a different size
    out, _ = ops.adain(c.cuda(), s.cuda(), 1e-5, 1e-5, alpha)
    torch.cuda.synchronize()
    ref = R.blend(R.adain_forward(c, s, torch.tensor([1e-5]), torch.tensor([1e-5]), 1e-5), c, alpha)
    assert rel_err(out, ref) < 1e-4
    if alpha == 1.0:
        m, sd = ops.mean_std(out, 1e-5)
        sm, ssd = R.calc_mean_std(s, 1e-5)
        assert rel_err(m, sm) < 1e-4 and rel_err(sd, ssd) < 1e-3


def test_stylize_2048_alpha_linearity(models):
    """Whole pipeline at 2048x2048: decoding alpha-blended features is what stylize(alpha) does -- the blend is
    fused into the AdaIN kernel, so stylize(alpha) must equal decoder(alpha*t + (1-alpha)*fc) built from the
    alpha=1 features."""
    from style_transfer_amd.pipeline import stylize
    from style_transfer_amd.transformer import AdaIN
    enc, dec, _, _ = models
    ada = AdaIN().cuda()
    g = torch.Generator().manual_seed(3)
    c = torch.rand(1, 3, 2048, 2048, generator=g).cuda()
    s = torch.rand(1, 3, 2048, 2048, generator=g).cuda()
    with torch.no_grad():
        fc, fs = enc(c), enc(s)
        t = ada(fc, fs, alpha=1.0)
        for alpha in (0.5, 1.5):
            want = dec((alpha * t + (1 - alpha) * fc).contiguous())
            got = stylize(enc, dec, ada, c, s, alpha=alpha)
            assert got.shape == (1, 3, 2048, 2048)
            assert rel_err(got, want) < 1e-4


def test_stylize_configs1_batch_matches_oracle_on_decoder_side_crops(models):
    """BASELINE configs[1] (batch 16 at 512x512) through the WHOLE pipeline: the stylised image against the oracle's
    decoder run on corner crops of the pipeline's own AdaIN features (output pixels more than ~6 feature cells from
    a crop's cut edges depend on nothing outside it), and those AdaIN features against the oracle's AdaIN of the
    full-size encoder features (exact per plane: AdaIN statistics need the whole plane, which is why the crops are
    taken after it).  Encoder side at this size: test_encoder_full_size_matches_oracle_on_crops."""
    from style_transfer_amd.pipeline import stylize
    from style_transfer_amd.transformer import AdaIN
    enc, dec, enc_sd, dec_sd = models
    ada = AdaIN().cuda()
    g = torch.Generator().manual_seed(16)
    c = torch.rand(16, 3, 512, 512, generator=g)
    s = torch.rand(16, 3, 512, 512, generator=g)
    with torch.no_grad():
        out = stylize(enc, dec, ada, c.cuda(), s.cuda(), alpha=1.0)
        fc, fs = enc(c.cuda()), enc(s.cuda())
        t = ada(fc, fs, alpha=1.0)
    torch.cuda.synchronize()
    assert out.shape == (16, 3, 512, 512)
    for n in (0, 15):
        t_ref = R.adain_forward(fc[n:n + 1].cpu(), fs[n:n + 1].cpu(), torch.tensor([1e-5]), torch.tensor([1e-5]), 1e-5)
        assert rel_err(t[n:n + 1], t_ref) < 1e-4
        tl = R.decoder_forward(t[n:n + 1, :, :24, :24].cpu(), dec_sd)          # 24 cells -> 192 px, 128 px exact
        br = R.decoder_forward(t[n:n + 1, :, -24:, -24:].cpu(), dec_sd)
        assert rel_err(out[n:n + 1, :, :128, :128], tl[:, :, :128, :128]) < 1e-3
        assert rel_err(out[n:n + 1, :, -128:, -128:], br[:, :, -128:, -128:]) < 1e-3


def test_bf16_g8_training_chains_at_the_configs3_size():
    """BASELINE configs[3] per-GPU shape (B=8, 512x512, bf16): decoder + loss VGG on the decoded image through the G8
    chains (autograd_conv.DecoderChainG8Function, autograd_vgg.VGGChainFunction) against the per-layer /
    fp32-activation forms -- bit-identical taps, input gradient and weight gradients at the size where the persistent
    kernels walk thousands of tiles and offsets pass 2^31 bytes."""
    import warnings
    from style_transfer_amd import precision
    from style_transfer_amd.generator import Decoder
    from style_transfer_amd.loss_network import VGG19FeatureExtractor
    sets = [[21], [2, 7, 14, 19, 25]]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vgg = VGG19FeatureExtractor([2, 7, 14, 19, 21, 25], weights=R.make_loss_vgg_state(2, upto=25)).cuda()
    dec = Decoder().cuda()
    dec.load_state_dict(R.make_decoder_state(1))
    g = torch.Generator().manual_seed(11)
    x0 = torch.relu(torch.randn(8, 512, 64, 64, generator=g)).cuda()
    wts = [0.7, -0.3, 0.5, 1.1, -0.9, 0.4]

    def run(chains):
        dec.G8_CHAIN = vgg.GRAD_CHAIN = chains
        for p in dec.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        with precision.compute_dtype("bf16"):
            img = dec(x)
            taps = [t for s in vgg.forward_multi(img, sets) for t in s]
            loss = sum(w * (t * t).mean() for w, t in zip(wts, taps))
            loss.backward()
        out = ([t.detach().clone() for t in taps[-2:]], x.grad.clone(), [p.grad.clone() for p in dec.parameters()])
        del taps, img, loss
        return out
    try:
        ref = run(False)
        torch.cuda.empty_cache()
        got = run(True)
    finally:
        dec.G8_CHAIN = vgg.GRAD_CHAIN = True
    for a, b in zip(got[0], ref[0]):
        assert torch.equal(a, b)
    assert float(ref[1].abs().max()) > 0 and torch.equal(got[1], ref[1])
    for k, (a, b) in enumerate(zip(got[2], ref[2])):
        if a.dim() == 4 and k == len(got[2]) - 2:
            # the 64 -> 3 layer's vector-ALU weight gradient reads its input in G8: bf16 rounding of one operand against
            # the fp32-activation chain (bit-level check of that kernel: tests/test_gpu_decoder_chain_g8.py)
            assert float((a - b).norm() / b.norm()) < 5e-3
        elif a.dim() == 4:
            assert torch.equal(a, b)
        else:
            torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-5 * float(b.abs().max()))
