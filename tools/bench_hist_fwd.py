import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from style_transfer_amd.losses import HistLoss
h = HistLoss()
x = torch.rand(8, 3, 512, 512, device="cuda")
for _ in range(3): h.histogram_block(x)
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): h.histogram_block(x)
e.record(); e.synchronize()
print(f"hist fwd B=8 512^2: {a.elapsed_time(e)/50*1e3:.1f} us")
