#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per dispatch of one kernel.
usage: pmc_summary.py <kernel substring> <counter_collection.csv> [...]"""
import csv, sys, collections
kern = sys.argv[1]
for path in sys.argv[2:]:
    acc = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if kern in row['Kernel_Name']:
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
    for k, v in acc.items():
        print(f'{k:32s} n={len(v):4d} mean={sum(v)/len(v):.6g}')
