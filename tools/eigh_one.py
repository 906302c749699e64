import sys, torch
sys.path.insert(0, ".")
from style_transfer_amd import eigh
c, b = int(sys.argv[1]), int(sys.argv[2])
x = torch.randn(b, c, 1024, device="cuda"); x = x - x.mean(-1, keepdim=True); a = x @ x.transpose(-1, -2)
for _ in range(3):
    eigh.syevd(a)
torch.cuda.synchronize()
