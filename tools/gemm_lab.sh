#!/bin/bash
# Ablation builds of the tiled GEMM (results are wrong by construction; timing only): which part of the K loop costs what.
# Run on the GPU box from the repo root:  bash tools/gemm_lab.sh "filter"
set -e
cd "$(dirname "$0")/.."
F="${1:-M=9728}"
for lab in ${LABS:-0 1 3 7}; do
  mkdir -p /tmp/lab$lab
  for f in core gemm_f32 layernorm attention attention_mfma rowops gather align lenpred decode optim noise pack; do
    if [ $f = gemm_f32 ]; then /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DNAST_GEMM_LAB=$lab -c nast_amd/csrc/$f.hip -o /tmp/lab$lab/$f.o 2>/dev/null; else cp nast_amd/csrc/$f.o /tmp/lab$lab/$f.o; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lab$lab/libnast_hip.so /tmp/lab$lab/*.o
  echo "== NAST_GEMM_LAB=$lab (1: no global loads, 2: no LDS stores, 4: no fragment reads, 8: no atomic C adds)"
  LD_LIBRARY_PATH=/tmp/lab$lab GB_M=9728 GB_ROT=4 timeout -k 5 200 tools/gemm_bench_lab "$F" | grep -v decode
done
