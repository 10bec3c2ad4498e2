#!/bin/bash
# Where a cluster decode (csrc/decode_cluster.hip) spends its time: builds a STAMPED copy of the library (-DCLUSTER_STAMPS:
# wall-clock stamps summed per phase by two workgroups of cluster 0, printed after every decode) next to the shipped one.
#   here (CPU):   tools/cluster_stamps.sh build
#   GPU box:      tools/cluster_stamps.sh run [B ...]
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  make -C nast_amd/csrc -j8 >/dev/null
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCLUSTER_STAMPS -c nast_amd/csrc/decode_cluster.hip -o /tmp/decode_cluster_stamps.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o nast_amd/lib/libnast_hip_stamps.so $(ls nast_amd/csrc/*.o | grep -v decode_cluster.o) /tmp/decode_cluster_stamps.o
  ls -la nast_amd/lib/libnast_hip_stamps.so
else
  shift || true
  NAST_HIP_LIB=$PWD/nast_amd/lib/libnast_hip_stamps.so python -u tools/decode_bench.py "$@" 2>&1 | grep -v amdgpu.ids
fi
