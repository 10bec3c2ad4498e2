#!/bin/bash
# Regenerates profiles/${R}_* from gpurun_out/ (bench_${R}_default.log, prof_${R}/, prof_${R}_noov/[, pmc_fetch/, pmc_write/]).
set -e
R=${1:-r2}          # round tag: profiles/<tag>_*  from gpurun_out/{bench_<tag>_default.log, prof_<tag>/, prof_<tag>_noov/, pmc_<tag>_{fetch,write,mfma}/}
RN=${R#r}
cd "$(dirname "$0")/.."
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-also   (round ${RN}, MI355X, 512 sentences/GPU)"
  echo "# whole run = 5 warm-up + 20 timed (two-stream overlap) + 5 roofline steps (overlap off) ; 'per step' figures divide by the number of f_steps in the run (= launches of the NoiseLayer kernel)."
  echo "# Kernels of the two streams run concurrently in 25 of the 30 steps, so per-kernel durations here are longer than a kernel alone;"
  echo "# the un-overlapped durations that bench.py's roofline block measures are in ${R}_bench_kernel_stats_no_overlap.txt"
  echo "# bench line of the profiled run:"; grep "^{" gpurun_out/prof_${R}.log; echo
  python tools/prof_summary.py gpurun_out/prof_${R} 0 40; } > profiles/${R}_bench_kernel_stats.txt
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-also --no-overlap   (round ${RN}, MI355X, 512 sentences/GPU)"
  echo "# single stream: every kernel runs alone, durations comparable with roofline.avg_launch_us of the bench line"
  echo "# bench line of the profiled run:"; grep "^{" gpurun_out/prof_${R}_noov.log; echo
  python tools/prof_summary.py gpurun_out/prof_${R}_noov 0 40; echo
  echo "# GEMM launches grouped by workgroup count"; python tools/trace_groups.py gpurun_out/prof_${R}_noov 0 nast_gemm 25; } > profiles/${R}_bench_kernel_stats_no_overlap.txt
cp gpurun_out/prof_${R}/p_kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
cp gpurun_out/prof_${R}_noov/p_kernel_stats.csv profiles/${R}_bench_kernel_stats_no_overlap.csv
grep "^{" gpurun_out/bench_${R}_default.log > profiles/${R}_bench_line.json
if [ -d gpurun_out/pmc_${R}_fetch ] && [ -d gpurun_out/pmc_${R}_write ]; then
  python tools/pmc_traffic.py gpurun_out/pmc_${R}_fetch gpurun_out/pmc_${R}_write --json profiles/${R}_pmc_traffic.json > /tmp/traffic.txt
  { echo "# HBM-side traffic per launch, round ${RN}, MI355X: two separate PMC passes over"
    echo "#   rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-also --no-overlap --no-roofline"
    echo "#   rocprofv3 --pmc WRITE_SIZE -- (same)"
    echo "# FETCH_SIZE doubled (gfx950: 128-B requests of 16 B/lane loads tallied at 64 B), WRITE_SIZE as is; both include Infinity-Cache hits."
    echo "# check: nast_log_softmax_fwd_rr_kernel reads and writes rows x 9600 fp32 once each"
    cat /tmp/traffic.txt; } > profiles/${R}_pmc_traffic.txt
fi
