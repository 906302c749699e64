#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc --kernel-trace CSV pair per dispatch:
duration, effective clock (GRBM_GUI_ACTIVE / 8 / time), MFMA busy fraction, wait shares."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
cc = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"])
cnt = defaultdict(dict)
for r in csv.DictReader(open(cc)):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
filt = sys.argv[2] if len(sys.argv) > 2 else "conv3x3"
for did, c in cnt.items():
    ns, name, gx, gy = dur[did]
    if filt not in name:
        continue
    short = name.split("(")[0][-40:] if "conv3x3" not in name else name[name.index("conv3x3"):name.index(">") + 1]
    line = f"{short:34s} grid={gx:>8s}x{gy:<3s} {ns/1e3:9.1f} us"
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        line += f" clk={cyc/ns:5.2f}GHz"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            line += f" mfma_busy={c['SQ_VALU_MFMA_BUSY_CYCLES']/(cyc*256*4):6.3f}"
    for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES"):
        if k in c and "SQ_WAVE_CYCLES" in c and k != "SQ_WAVE_CYCLES":
            line += f" {k[3:]}={c[k]/c['SQ_WAVE_CYCLES']:.3f}"
    for k, v in c.items():
        if k.startswith("TCC") or k in ("FETCH_SIZE", "WRITE_SIZE"):
            line += f" {k}={v:.4g}"
    print(line)
