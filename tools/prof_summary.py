#!/usr/bin/env python3
"""Per-step view of a rocprofv3 kernel_stats.csv of a training leg: python tools/prof_summary.py FILE [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = sum(float(r['TotalDurationNs']) for r in rows)
steps = next((int(r['Calls']) for r in rows if 'adain_bwd' in r['Name']), 1)
print(f"steps {steps}  GPU-busy ms/step {tot / steps / 1e6:.3f}")
cat = {}
def bucket(n):
    if 'wgrad' in n or 'slab_reduce' in n: return 'weight gradient (+ reduces, prep)'
    if 'conv3x3' in n: return 'conv launches'
    if 'band' in n or 'fold' in n: return 'pad/upsample adjoint (fold, band)'
    if 'join' in n or 'pool' in n: return 'joins + pools'
    if 'at::' in n or 'rocclr' in n or 'elementwise' in n: return 'ATen / copies'
    return 'loss + other kernels'
for r in rows:
    cat[bucket(r['Name'])] = cat.get(bucket(r['Name']), 0.0) + float(r['TotalDurationNs'])
for k, v in sorted(cat.items(), key=lambda kv: -kv[1]):
    print(f"  {v / steps / 1e6:7.3f} ms  {100 * v / tot:5.1f} %  {k}")
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs'])/steps/1e6:7.3f} ms {int(r['Calls'])/steps:5.1f}x avg {float(r['AverageNs'])/1e3:8.1f}us  {r['Name'].replace('(anonymous namespace)::','')[:120]}")
