#!/usr/bin/env python3
"""mean of every counter per kernel from rocprofv3 --pmc output directories: pmc_dump.py <kernel substr> dir..."""
import collections, csv, glob, os, sys
sub = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(vals):
            v = sorted(vals[k])
            print(f"{k:32s} n={len(v):3d} max={v[-1]:.4g} median={v[len(v)//2]:.4g}")
