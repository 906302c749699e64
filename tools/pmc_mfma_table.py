#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace run:
launches, average duration, effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and the matrix pipes' busy fraction
(SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 256 CUs x 4 SIMDs)), averaged over the launches of each kernel instantiation.
    python tools/pmc_mfma_table.py <rocprof output dir> [name filter ...]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
filters = sys.argv[2:] or ["conv3x3", "cov_", "wgrad"]
cc = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
cnt = defaultdict(dict)
for r in csv.DictReader(open(cc)):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for did, c in cnt.items():
    if did not in dur or "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        continue
    ns, name = dur[did]
    if not any(f in name for f in filters):
        continue
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    a = agg[name]
    a[0] += 1
    a[1] += ns
    a[2] += cyc
    a[3] += c["SQ_VALU_MFMA_BUSY_CYCLES"]
print(f"{'kernel':78s} {'calls':>6s} {'avg us':>9s} {'clk GHz':>8s} {'MFMA busy':>10s}")
for name, (n, ns, cyc, busy) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:78]:78s} {n:6d} {ns / n / 1e3:9.1f} {cyc / ns:8.2f} {busy / (cyc * 256 * 4):10.3f}")
