#!/bin/bash
# Run on the GPU box (gpurun --timeout 1100 -- 'tools/collect_profiles.sh r2'): the default bench line, two rocprofv3 kernel traces
# (two-stream and single-stream) and three counter passes of the same command, all under gpurun_out/.  tools/make_profiles.sh <tag>
# then turns them into profiles/<tag>_*.
R=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
B="python3 bench.py"
S="--steps 3 --warmup 2 --no-cpu-baseline --no-also --no-overlap --no-roofline"
rm -rf gpurun_out/prof_${R} gpurun_out/prof_${R}_noov gpurun_out/pmc_${R}_fetch gpurun_out/pmc_${R}_write gpurun_out/pmc_${R}_mfma
timeout -k 10 300 $B > gpurun_out/bench_${R}_default.log 2> gpurun_out/bench_${R}_default.err &&
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R} -o p -- $B --no-cpu-baseline --no-also > gpurun_out/prof_${R}.log 2>&1 &&
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}_noov -o p -- $B --no-cpu-baseline --no-also --no-overlap > gpurun_out/prof_${R}_noov.log 2>&1 &&
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${R}_fetch -o p -- $B $S > gpurun_out/pmc_${R}_fetch.log 2>&1 &&
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${R}_write -o p -- $B $S > gpurun_out/pmc_${R}_write.log 2>&1 &&
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${R}_mfma -o p -- $B $S > gpurun_out/pmc_${R}_mfma.log 2>&1
rc=$?
echo rc=$rc
grep "^{" gpurun_out/bench_${R}_default.log | cut -c1-300
exit $rc
