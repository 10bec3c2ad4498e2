#!/usr/bin/env python3
"""Print the kernel_stats.csv rows of a rocprofv3 --kernel-trace --stats run whose kernel name contains a pattern.
usage: kstats.py <output dir> <pattern>"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Name"]:
            print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"]) / 1e3:8.2f} min_us {float(r["MinNs"]) / 1e3:8.2f} max_us {float(r["MaxNs"]) / 1e3:8.2f}')
