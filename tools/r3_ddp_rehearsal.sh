# Round 3: the data-parallel path on a one-GPU box.  (1) one-rank RCCL group: reducer off / overlapped buckets (default) /
# after-backward; (2) `bench.py --gpus 2` launching its own ranks (gloo: two ranks share the card).
set -u
O=gpurun_out/r3_ddp; mkdir -p $O
B="--steps 20 --warmup 5 --no-also --no-cpu-baseline --no-roofline"
run() { name=$1; shift
  env "$@" timeout -k 10 240 python bench.py $B > $O/$name.json 2> $O/$name.err
  rc=$?; echo "$name rc=$rc $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(d['ms_per_step'], d['value'])" 2>/dev/null)"; return $rc; }
run plain NAST_X=0 &&
run group_reducer_off NAST_DIST_SINGLE=1 NAST_DDP_DISABLE=1 &&
run group_overlap NAST_DIST_SINGLE=1 NAST_BENCH_DEBUG=0 &&
run group_after_backward NAST_DIST_SINGLE=1 NAST_DDP_OVERLAP=0 &&
{ NAST_DIST_BACKEND=gloo NAST_DDP_TIMEOUT_S=40 NAST_BENCH_TIMEOUT_S=200 timeout -k 10 230 python bench.py --gpus 2 --batch-per-gpu 128 --steps 4 --warmup 2 --no-also --no-cpu-baseline > $O/selflaunch_n2.json 2> $O/selflaunch_n2.err
  echo "selflaunch_n2 rc=$? $(cut -c1-300 $O/selflaunch_n2.json)"; }
