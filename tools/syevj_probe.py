"""Sweeps, time and accuracy of the Jacobi kernel on the Gram loss's hardest input (root.cov.root of real loss-VGG
features, C = 64) for the current ST_SYEVJ_ABS_TOL."""
import os, sys, warnings, contextlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_ops as R
from style_transfer_amd import eigh, _lib
from style_transfer_amd.loss_network import VGG19FeatureExtractor, Normalize
from style_transfer_amd.losses import VincentStyleLoss
dev = torch.device("cuda")
with warnings.catch_warnings(), contextlib.redirect_stdout(sys.stderr):
    warnings.simplefilter("ignore")
    vgg = VGG19FeatureExtractor([2, 7], weights=R.make_loss_vgg_state(2, upto=25)).to(dev)
    vl = VincentStyleLoss()
g = torch.Generator().manual_seed(5)
for side in (64, 256):
    a = torch.rand(8, 3, side, side, generator=g).to(dev); b = torch.rand(8, 3, side, side, generator=g).to(dev)
    with torch.no_grad():
        fa = vgg.forward_multi(Normalize()(a), [[2, 7]])[0]
        fb = vgg.forward_multi(Normalize()(b), [[2, 7]])[0]
        for f1, f2 in zip(fa, fb):
            _, c1 = vl.calc_2_moments(f1); _, c2 = vl.calc_2_moments(f2)
            root, _ = vl._style_side(c2, int(f2.shape[2] * f2.shape[3]))
            m = root @ c1 @ root
            m = 0.5 * (m + m.transpose(1, 2))
            C = m.shape[-1]
            w, v = eigh.syevj_small(m)
            sw = eigh._small_pending[-1].tolist(); eigh._small_pending.clear()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                eigh.syevj_small(m)
            e1.record(); e1.synchronize(); eigh._small_pending.clear()
            w64 = torch.linalg.eigvalsh(m.double().cpu())
            err = ((w.double().cpu() - w64).abs().max(dim=1).values / w64.abs().max(dim=1).values).max().item()
            rec = (v * w.unsqueeze(-2)) @ v.transpose(1, 2)
            rerr = ((rec - m).flatten(1).norm(dim=1) / m.flatten(1).norm(dim=1)).max().item()
            ov = (torch.sqrt(w.clip(0.1)).sum(1).double().cpu() - torch.sqrt(w64.clip(0.1)).sum(1)).abs().max().item() / torch.sqrt(w64.clip(0.1)).sum(1).max().item()
            print(f"side {side} C={C}: sweeps {sw}  {e0.elapsed_time(e1) / 5:.3f} ms  max eig err/lmax {err:.2e}  recon {rerr:.2e}  var_overlap rel err {ov:.2e}", flush=True)
