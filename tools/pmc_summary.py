#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (averages per dispatch)."""
import collections, csv, glob, re, sys

path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
f = (glob.glob(path + "/*/*counter_collection.csv") + glob.glob(path + "/*counter_collection.csv"))[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:40]
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if pat and pat not in k:
        continue
    d = {c: sum(x) / len(x) for c, x in v.items()}
    wc = d.get("SQ_WAVE_CYCLES")
    line = f"{k:28s} n={len(next(iter(v.values())))} " + " ".join(f"{c.replace('SQ_', '')}={x:.3g}" for c, x in sorted(d.items()))
    print(line)
    if wc:
        print("   per wave-cycle: " + " ".join(f"{c.replace('SQ_', '')}={x / wc:.3f}" for c, x in sorted(d.items()) if c != "SQ_WAVE_CYCLES" and ("ACTIVE" in c or "WAIT" in c)))
