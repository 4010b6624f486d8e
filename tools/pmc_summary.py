#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per launch of one kernel from counter_collection.csv files."""
import csv, sys, glob, collections
kernel = sys.argv[1]
acc = collections.defaultdict(list)
for path in sys.argv[2:]:
    for f in glob.glob(path):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]; v2 = v[len(v)//4:]          # skip the first quarter (warm-up launches)
    print(f"{k:28s} n={len(v2):4d} mean={sum(v2)/len(v2):16.1f}")
