#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/kernel_regs.py style_transfer_amd/csrc/conv3x3_bf16.hip [name filter]"""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(REPO, "include"),
       "-I", os.path.join(REPO, "style_transfer_amd", "csrc"), "-x", "hip", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
for ln in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", ln)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        if cur and flt in cur["name"]:
            print(cur)
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": name.replace("(anonymous namespace)::", "")[:110]}
    else:
        k, _, v = t.partition(":")
        if k.strip() in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
            cur[k.strip().split()[0]] = v.strip()
if cur and flt in cur["name"]:
    print(cur)
