#!/bin/bash
# Kernel-only durations of the attention backward variants (run on the GPU box): rocprofv3 kernel trace of tools/attn_bench.py
# usage: [MODE=packed] tools/attn_prof.sh "<B> <n>" <split 0|1>:x ...
cd /tmp && export TMPDIR=/tmp
shape="$1"; shift
for v in "$@"; do
  form=${v%%:*}; dbg=${v##*:}
  out=/tmp/ap_${form}_${dbg}
  rm -rf $out
  NAST_ATTN_SPLIT=$form timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/attn_bench.py $shape 256 4 100 bwd $MODE > /dev/null 2>&1 || exit 1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "form=$form dbg=$dbg shape=$shape: $(grep -E 'nast_mha_bwd' $f | python3 -c 'import sys,csv
for r in csv.reader(sys.stdin): print(r[0][28:60], "calls", r[1], "avg_us", float(r[3])/1e3)')"
done
