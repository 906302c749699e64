"""Host side of the GPU augmentation (gpu_augment.GpuAugment.draw): the per-image draws follow the distributions
of the reference's transform lists (src/train.py:116-135).  CPU only."""
import torch

from style_transfer_amd.gpu_augment import GpuAugment


def test_draw_ranges_and_rates():
    n = 4000
    c, s = GpuAugment.for_content(seed=0).draw(n), GpuAugment.for_style(seed=0).draw(n)
    for p in (c, s):
        assert p["angle_deg"].abs().max() <= 90 and p["angle_deg"].std() > 40          # U(-90, 90): std 52
        on = p["sharp_factor"] >= 0
        assert 0.45 < on.float().mean() < 0.55 and torch.all(p["sharp_factor"][on] == 1.5)
        assert 0.45 < p["flip"].float().mean() < 0.55
        assert torch.equal(p["jitter_order"].sort(dim=1).values, torch.arange(4, dtype=torch.int32).expand(n, 4))
        first = torch.bincount(p["jitter_order"][:, 0].long(), minlength=4).float() / n
        assert torch.all((first - 0.25).abs() < 0.04)                                   # a uniform permutation
        b, _, sat, hue = p["jitter_param"].unbind(1)
        assert 0.8 <= b.min() and b.max() <= 1.2 and 0.7 <= sat.min() and sat.max() <= 1.3 and hue.abs().max() <= 0.1
    assert 0.6 <= c["jitter_param"][:, 1].min() and c["jitter_param"][:, 1].max() <= 1.4     # content contrast 0.4
    assert 0.75 <= s["jitter_param"][:, 1].min() and s["jitter_param"][:, 1].max() <= 1.25   # style contrast 0.25
    assert "noise_sigma" not in c
    on = s["noise_sigma"] >= 0
    assert 0.45 < on.float().mean() < 0.55 and s["noise_sigma"][on].max() <= 0.08


def test_switched_off_adjustments_and_seeding():
    p = GpuAugment(rotation_degrees=None, sharpness_factor=None, brightness=0, hue=0, seed=1).draw(16)
    assert torch.all(p["angle_deg"] == 0) and torch.all(p["sharp_factor"] < 0)
    assert torch.isnan(p["jitter_param"][:, 0]).all() and torch.isnan(p["jitter_param"][:, 3]).all()
    a, b = GpuAugment.for_style(seed=5).draw(8), GpuAugment.for_style(seed=5).draw(8)
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_cli_flag_moves_the_tensor_transforms_off_the_workers(tmp_path):
    from style_transfer_amd import train
    d = str(tmp_path)
    base = ["--output_dir", d, "--content_datapath", d, "--style_datapath", d]
    a0, a1 = train.parse_args(base), train.parse_args(base + ["--gpu_augment", "--grayscale_content_transform"])
    assert train.build_gpu_augment(a0) is None and len(train.build_transforms(a0)[0].transforms) == 8
    c, s = train.build_gpu_augment(a1, rank=1)
    assert len(train.build_transforms(a1)[0].transforms) == 3 and len(train.build_transforms(a1)[1].transforms) == 3
    assert c.grayscale and c.noise is None and s.noise == {"mean": 0.5, "sigma_range": (0.0, 0.08), "p": 0.5}
    assert c.jitter["contrast"] == 0.4 and s.jitter["contrast"] == 0.25
