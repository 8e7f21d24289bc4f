#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection CSV: mean counter value per kernel name (filter substring optional)."""
import csv, sys, collections, glob, re
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
filt = sys.argv[2] if len(sys.argv) > 2 else "gemm_kernel"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if filt in name:
            short = re.sub(r".*gemm_kernelI", "gemm<", name)[:60]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
