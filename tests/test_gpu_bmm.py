"""st_bmm_f32 (csrc/bmm.hip): the batched C x C products of VincentStyleLoss.l2wass_dist (src/models/losses.py:67-81)
and of the Newton-Schulz square root on the fp32 matrix pipe, against float64 matmul.  fp32 bar: 1e-5 relative (the
kernel multiplies exact fp32 products and accumulates in fp32 in k order, like an fmaf chain)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("batch,n", [(16, 512), (8, 256), (3, 128), (5, 64), (1, 192)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_bmm_matches_float64(batch, n, ta, tb):
    from style_transfer_amd import ops
    g = torch.Generator().manual_seed(n + batch)
    a = torch.randn(batch, n, n, generator=g)
    b = torch.randn(batch, n, n, generator=g)
    got = ops.bmm_raw(a.cuda(), b.cuda(), ta, tb)
    a64, b64 = a.double(), b.double()
    ref = (a64.transpose(1, 2) if ta else a64) @ (b64.transpose(1, 2) if tb else b64)
    assert _rel(got, ref) < 1e-5
    got2 = ops.bmm_raw(a.cuda(), b.cuda(), ta, tb, alpha=-0.5, diag=1.5)        # the Newton-Schulz step's epilogue
    assert _rel(got2, -0.5 * ref + 1.5 * torch.eye(n, dtype=torch.float64)) < 1e-5


def test_bmm_rejects_what_it_cannot_do():
    from style_transfer_amd import _lib, ops
    a = torch.randn(2, 96, 96).cuda()
    assert not ops.bmm_ok(a)
    with pytest.raises((AssertionError, _lib.StHipError)):
        ops.bmm_raw(a, a)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_bmm_gradients(ta, tb):
    from style_transfer_amd import ops
    g = torch.Generator().manual_seed(7)
    a0 = torch.randn(4, 128, 128, generator=g)
    b0 = torch.randn(4, 128, 128, generator=g)
    w = torch.randn(4, 128, 128, generator=g)
    a, b = a0.cuda().requires_grad_(True), b0.cuda().requires_grad_(True)
    (ops.bmm(a, b, ta, tb) * w.cuda()).sum().backward()
    a64, b64 = a0.double().requires_grad_(True), b0.double().requires_grad_(True)
    (((a64.transpose(1, 2) if ta else a64) @ (b64.transpose(1, 2) if tb else b64)) * w.double()).sum().backward()
    assert _rel(a.grad, a64.grad) < 1e-5 and _rel(b.grad, b64.grad) < 1e-5


def test_gram_loss_is_the_same_on_both_product_routes(monkeypatch):
    """VincentStyleLoss with its C x C products on st_bmm_f32 against the same loss on torch.matmul (rocBLAS): value
    and gradient agree to fp32 rounding (both routes are pinned to the reference's golden value in
    tests/test_gpu_backward.py)."""
    import warnings
    from style_transfer_amd import eigh
    from style_transfer_amd.losses import VincentStyleLoss
    g = torch.Generator().manual_seed(3)
    x0 = torch.relu(torch.randn(2, 128, 24, 24, generator=g)).cuda()
    t = torch.relu(torch.randn(2, 128, 24, 24, generator=g)).cuda()
    out = []
    for hip in (True, False):
        monkeypatch.setattr(VincentStyleLoss, "HIP_BMM", hip)
        monkeypatch.setattr(eigh, "HIP_BMM", hip)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            crit = VincentStyleLoss()
        x = x0.clone().requires_grad_(True)
        loss = crit(x, t)
        loss.backward()
        out.append((float(loss), x.grad.clone()))
    assert abs(out[0][0] - out[1][0]) <= 1e-4 * abs(out[1][0])
    assert _rel(out[0][1], out[1][1].double().cpu()) < 1e-3
