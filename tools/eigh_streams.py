import sys, torch
sys.path.insert(0, ".")
from style_transfer_amd import eigh
def spd(b, c, n):
    x = torch.randn(b, c, n, device="cuda"); x = x - x.mean(-1, keepdim=True); return x @ x.transpose(-1, -2)
group = [spd(8, 64, 4096), spd(8, 128, 4096), spd(8, 256, 4096), spd(16, 512, 1024)]
streams = [torch.cuda.Stream() for _ in group]
for rep in range(3):
    cur = torch.cuda.current_stream()
    for m, s in zip(group, streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            eigh.syevd(m)
    for s in streams:
        cur.wait_stream(s)
    torch.cuda.synchronize()
