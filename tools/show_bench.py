#!/usr/bin/env python3
"""Print the headline numbers of a bench.py JSON line (last line of the given file)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])


def leg(name, o):
    if o:
        print(f"{name:22s} {o['value']:9.1f} img/s  {o['ms_per_step']:8.3f} ms/step")


if "value" in d and "metric" in d:
    leg("infer " + d.get("dtype", ""), d)
    if "roofline" in d:
        print("  roofline", d["roofline"].get("achieved"), d["roofline"].get("frac"))
leg("train f32", d.get("train"))
if d.get("train"):
    leg("train f32 graph", d["train"].get("hip_graph"))
    leg("train f32 gram", d["train"].get("gram"))
for k in ("bf16", "bf16x3"):
    if d.get(k):
        leg(k + " infer", d[k].get("inference"))
        if d[k].get("inference"):
            print("  roofline", d[k]["inference"]["roofline"].get("achieved"), d[k]["inference"]["roofline"].get("frac"))
        leg(k + " train", d[k].get("train"))
