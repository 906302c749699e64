import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import ref_ops as R
from tests.test_gpu_train import _build, rel_err
from style_transfer_amd.autograd import conv3x3_apply
from style_transfer_amd import ops
warnings.simplefilter("ignore")
content_idx, style_idx = [21], [2, 7, 14, 19, 25]
step, enc_sd, dec_sd, vgg_sd = _build(False, content_idx, style_idx, 0.1)
g = torch.Generator().manual_seed(5)
content = torch.rand(2, 3, 32, 32, generator=g); style = torch.rand(2, 3, 32, 32, generator=g)
fc = R.encoder_forward(content, enc_sd); fs = R.encoder_forward(style, enc_sd)
t = R.adain(fc, fs, 1e-5 ** 0.5, 1e-5 ** 0.5).detach()
# oracle decoder, layer by layer
xs_ref = [t.clone().requires_grad_(True)]
x = xs_ref[0]
for idx, cin, cout in R.DECODER_CONVS:
    x = F.conv2d(R.symmetric_pad(x), dec_sd[f"decoder.{idx}.weight"], dec_sd[f"decoder.{idx}.bias"])
    if idx != 28: x = F.relu(x)
    x.retain_grad(); xs_ref.append(x)
    if idx in R.DECODER_UP_AFTER: x = F.interpolate(x, scale_factor=2, mode="nearest")
sf = zip(R.loss_vgg_features(R.normalize(x), vgg_sd, style_idx), R.loss_vgg_features(R.normalize(style), vgg_sd, style_idx))
sum(R.adain_style_loss(a, b) for a, b in sf).backward()
# HIP decoder, layer by layer
xs = [t.clone().cuda().requires_grad_(True)]
x = xs[0]; up = False
seq = step.decoder.decoder
for idx, cin, cout in R.DECODER_CONVS:
    x = conv3x3_apply(x, seq[idx].weight, seq[idx].bias, step.decoder._packed, idx, upsample2x=up, pad_mode=ops.PAD_REPLICATE, relu=(idx != 28))
    x.retain_grad(); xs.append(x)
    up = idx in R.DECODER_UP_AFTER
gs = step.loss_vgg.forward_multi(step.norm(x), [style_idx])[0]
with torch.no_grad():
    os_ = step.loss_vgg.forward_multi(step.norm(style.cuda()), [style_idx])[0]
sum(step.style_loss(a, b) for a, b in zip(gs, os_)).backward()
for i, (a, b) in enumerate(zip(xs, xs_ref)):
    name = "input" if i == 0 else f"dec.{R.DECODER_CONVS[i-1][0]} out"
    print(f"{name:12s} shape {tuple(b.shape)} fwd rel {rel_err(a, b):.2e} grad rel {rel_err(a.grad, b.grad):.2e} |grad|max {float(b.grad.abs().max()):.3e}")
print("---- element statistics of the gradient mismatch ----")
for i, (a, b) in enumerate(zip(xs, xs_ref)):
    d = (a.grad.cpu().double() - b.grad.double()).abs()
    mx = float(b.grad.abs().max())
    fwd_a, fwd_b = a.detach().cpu(), b.detach()
    flips = int(((fwd_a > 0) != (fwd_b > 0)).sum())
    near0 = int((fwd_b.abs() < 1e-5).sum() - (fwd_b == 0).sum())
    print(i, "n>1e-3*max:", int((d > 1e-3 * mx).sum()), "of", d.numel(), "relu flips:", flips, "near-zero nonzero:", near0,
          "median rel", float((d / mx).median()))
