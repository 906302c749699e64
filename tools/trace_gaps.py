#!/usr/bin/env python3
"""Idle gaps between kernels in a rocprofv3 --kernel-trace CSV: where the GPU waits for the host.

    python tools/trace_gaps.py <dir-with-*kernel_trace.csv> [min_gap_us] [last_n_kernels]
"""
import csv
import glob
import sys

d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 0
f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
if last_n:
    rows = rows[-last_n:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} kernels, span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms ({100*busy/span:.1f}%)")
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = (s1 - e0) / 1e3
    if g >= min_gap:
        gaps.append((g, n0[:60], n1[:60]))
print(f"{len(gaps)} gaps >= {min_gap} us, total {sum(g for g, _, _ in gaps)/1e3:.3f} ms")
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print(f"{g:10.1f} us  after {a}  before {b}")
