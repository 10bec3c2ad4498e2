set -u
O=gpurun_out/r3_ddp; mkdir -p $O
B="--steps 20 --warmup 5 --no-also --no-cpu-baseline --no-roofline"
run() { name=$1; shift
  env "$@" timeout -k 10 240 python bench.py $B > $O/$name.json 2> $O/$name.err
  rc=$?; echo "$name rc=$rc $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(d['ms_per_step'], d['value'])" 2>/dev/null) $(grep -o 'reducer host profile.*' $O/$name.err | cut -c1-700)"; return $rc; }
run group_reducer_off NAST_DIST_SINGLE=1 NAST_DDP_DISABLE=1 &&
run group_overlap_prof NAST_DIST_SINGLE=1 NAST_DDP_PROF=1 &&
run group_overlap_nocheck NAST_DIST_SINGLE=1 NAST_DDP_CHECK=0 &&
run group_after_prof NAST_DIST_SINGLE=1 NAST_DDP_OVERLAP=0 NAST_DDP_PROF=1 &&
run group_overlap_1bucket NAST_DIST_SINGLE=1 NAST_DDP_BUCKET_MB=1024 &&
{ NAST_DIST_BACKEND=gloo NAST_DDP_BUCKET_MB=1 NAST_DDP_TIMEOUT_S=40 NAST_BENCH_TIMEOUT_S=150 timeout -k 10 180 python bench.py --gpus 2 --batch-per-gpu 128 --steps 4 --warmup 2 --no-also --no-cpu-baseline > $O/selflaunch_n2c.json 2> $O/selflaunch_n2c.err
  echo "selflaunch_n2 rc=$? $(cut -c1-300 $O/selflaunch_n2c.json)"; tail -3 $O/selflaunch_n2c.err | cut -c1-300; }
