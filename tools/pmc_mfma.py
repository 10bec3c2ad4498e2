#!/usr/bin/env python3
"""MFMA utilisation per kernel family from one rocprofv3 PMC pass over the training step.

usage: pmc_mfma.py PMC_DIR
The pass must hold SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY,
SQ_INSTS_VALU, SQ_INSTS_LDS, SQ_LDS_BANK_CONFLICT (8 SQ counters) and GRBM_GUI_ACTIVE.
Derivations as in profiles/r1_gemm_pmc.txt (checked there against the event-timed TFLOP/s of one GEMM):
  cycles           = GRBM_GUI_ACTIVE / 8                     (the counter sums the 8 XCDs)
  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)
  waves per SIMD   = 4 x SQ_WAVE_CYCLES / (1024 x cycles)    (SQ wave counters tick in quad-cycles)
  LDS conflict     = SQ_LDS_BANK_CONFLICT / (256 CUs x cycles)   (extra LDS-array cycles per CU cycle)
"""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import family  # noqa: E402


def main():
    d = sys.argv[1]
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
    tot = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        k = family(name)
        if "nast_gemm_f32_kernel" in name:      # keep the tile variants apart
            k = "nast_gemm_f32_kernel<" + name.split("<", 1)[1].split(">", 1)[0].replace(" ", "") + ">"
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    rows = []
    for k, c in tot.items():
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc <= 0:
            continue
        rows.append((cyc, k, len(n[k]), c))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    print(f"{'kernel family':58s} {'launches':>8s} {'GPU cycles %':>12s} {'MFMA util %':>11s} {'waves/SIMD':>10s} {'wait-inst %':>11s} {'LDS conflict %':>14s}")
    for cyc, k, nl, c in rows[:24]:
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc)
        wv = 4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / (1024.0 * cyc)
        wi = c.get("SQ_WAIT_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        # extra LDS cycles per CU cycle.  (Until round 2 this column divided by SQ_BUSY_CYCLES, which ticks per shader engine,
        # not per CU: it overstated the share 8-10x -- 20.7 % for the attention backward where the CUs spend 2.2 %.)
        lc = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / (256.0 * cyc)
        print(f"{k[:58]:58s} {nl:8d} {100 * cyc / total:12.1f} {100 * mf:11.1f} {wv:10.2f} {100 * wi:11.1f} {100 * lc:14.2f}")


if __name__ == "__main__":
    main()
