import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_amd import ops
n = 8
k, c = 3, 64
for S in (512, 480, 544):
    g = torch.Generator(device="cuda").manual_seed(0)
    wt = torch.randn(k, c, 3, 3, device="cuda", generator=g) * 0.05
    wpt = ops.pack_conv3x3_weights(wt, transpose_flip=True, compute="bf16")
    dy = torch.randn(n, k, S, S, device="cuda", generator=g)
    gate = torch.relu(torch.randn(n, c, S, S, device="cuda", generator=g))
    for abl in (0, 16, 8 + 16 + 32 + 64):
        os.environ["ST_BF16_ABLATE"] = str(abl)
        f = lambda: ops.conv3x3_dgrad_g8_chain(dy, wpt, c, False, gate, False, True, True, True)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"ablate side {S} abl {abl:3d}: {us:8.1f} us   {us / (S * S) * 512 * 512:8.1f} us per 512^2-equivalent")
