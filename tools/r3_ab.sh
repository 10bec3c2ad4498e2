# usage: bash tools/r3_ab.sh name ENV=VAL ... ; quick A/B of the headline step: ms/step and tiled-GEMM ms/step
name=$1; shift
mkdir -p gpurun_out/r3_ab
env "$@" timeout -k 10 200 python bench.py --no-also --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r3_ab/$name.json 2> gpurun_out/r3_ab/$name.err
echo "$name rc=$? $(python -c "
import json; d=json.load(open('gpurun_out/r3_ab/$name.json')); r=d['roofline']
print('ms/step', d['ms_per_step'], 'gemm ms/step', r['gemm_ms_per_step'], 'TF', r['achieved'], 'frac', r['frac'])" 2>&1 | tail -1)"
