#!/bin/bash
# SQ counters of the attention backward kernel (run on the GPU box): where its wave-cycles go.
# usage: [MODE=packed] [NAST_ATTN_BWD_FORM=n] tools/attn_pmc.sh "<B> <n>"
cd /tmp && export TMPDIR=/tmp
out=/tmp/attn_pmc; rm -rf $out
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/attn_bench.py $1 256 4 20 bwd $MODE > /dev/null 2>&1 || exit 1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "nast_mha_bwd" not in k: continue
    acc[k[:70]][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k[:70]] += 1
for k, c in acc.items():
    print(k, "launches", n[k])
    for name, v in c.items(): print(f"   {name:28s} {v / n[k]:14.0f} per launch")
PY
