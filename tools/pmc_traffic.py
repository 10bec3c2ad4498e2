#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE).

usage: pmc_traffic.py FETCH_DIR WRITE_DIR [--json OUT]

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
  * both counters are in KiB-like units of 1024 B as rocprofv3 reports them;
  * FETCH_SIZE tallies the 128-B requests of wide (16 B/lane) loads at 64 B -> it is DOUBLED here;
  * WRITE_SIZE is exact for 16-B/lane stores and float atomics -> taken as is.
Infinity-Cache hits are included in both (memory-side request counters of the L2), so "traffic" is an upper bound
of what reaches HBM.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def family(name):
    m = re.search(r"(nast_[a-z0-9_]+)", name)
    if m:
        return m.group(1)
    if "rocclr" in name:
        return "rocclr_copy/fill"
    return "torch:" + re.sub(r"<.*", "", name.replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", ""))[:40]


def load(d, counter):
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = family(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"]) * 1024.0
        n[k] += 1
    return tot, n


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    ft, fn = load(fd, "FETCH_SIZE")
    wt, wn = load(wd, "WRITE_SIZE")
    rows = []
    for k in set(ft) | set(wt):
        n = max(fn.get(k, 0), wn.get(k, 0))
        fetch = 2.0 * ft.get(k, 0.0)
        write = wt.get(k, 0.0)
        rows.append((fetch + write, k, n, fetch, write))
    rows.sort(reverse=True)
    print(f"{'kernel family':44s} {'launches':>8s} {'fetch MB/launch (x2)':>22s} {'write MB/launch':>16s} {'total MB/launch':>16s} {'total GB':>9s}")
    res = {}
    for tot, k, n, fetch, write in rows[:40]:
        print(f"{k:44s} {n:8d} {fetch / n / 1e6:22.3f} {write / n / 1e6:16.3f} {tot / n / 1e6:16.3f} {tot / 1e9:9.3f}")
        res[k] = {"launches": n, "fetch_bytes_per_launch": fetch / n, "write_bytes_per_launch": write / n,
                  "bytes_per_launch": tot / n}
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
