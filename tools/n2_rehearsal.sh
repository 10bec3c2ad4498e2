# two ranks on ONE GPU through gloo: rehearses bench.py's multi-process path (rendezvous, sharded batches, gradient all-reduce,
# barrier, max-over-ranks, one JSON line from rank 0).  Not a performance figure: the ranks share the card.
export NAST_DIST_BACKEND=gloo NAST_DDP_TIMEOUT_S=${NAST_DDP_TIMEOUT_S:-40}
N=${1:-2}; B=${2:-128}
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29710 + RANDOM % 50)) bench.py --gpus $N --steps 4 --warmup 2 --batch-per-gpu $B > gpurun_out/n${N}_gloo.json 2> gpurun_out/n${N}_gloo.err
echo "N=$N rc=$? stuck=$(grep -c 'did not complete' gpurun_out/n${N}_gloo.err) lines=$(wc -l < gpurun_out/n${N}_gloo.json)"; cut -c1-600 gpurun_out/n${N}_gloo.json
