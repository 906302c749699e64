#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 `--pmc <counter> --kernel-trace --output-format csv` passes.

    python tools/pmc_hbm_csv.py <dir of pass 1> [<dir of pass 2> ...] > profiles/<round>_pmc_hbm.csv

One row per kernel name per counter: calls, average counter value per dispatch (FETCH_SIZE / WRITE_SIZE
are KiB as rocprofv3 reports them; the gfx950 reader-side correction -- double FETCH_SIZE for 16-B-per-lane
streaming reads, MI355X_MICROARCH.md HBM section -- is NOT applied here) and the average dispatch duration
of that profiled pass."""
import csv
import glob
import sys
from collections import defaultdict


def short(name: str) -> str:
    return name.replace("(anonymous namespace)::", "").replace("void ", "")[:90]


w = csv.writer(sys.stdout)
w.writerow(["Name", "Counter", "Calls", "AvgValueKiBPerDispatch", "AvgDurationNs"])
for d in sys.argv[1:]:
    cc = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(cc)):
        a = acc[(r["Kernel_Name"], r["Counter_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += dur.get(r["Dispatch_Id"], 0)
    for (name, ctr), (n, v, t) in sorted(acc.items(), key=lambda kv: -kv[1][2]):
        if t / n < 15000:       # leave out the sub-15-us kernels
            continue
        w.writerow([short(name), ctr, n, round(v / n, 1), round(t / n, 1)])
