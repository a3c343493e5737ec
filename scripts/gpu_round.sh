#!/bin/bash
# One GPU-box visit: gpu tests, bench line, rocprofv3 kernel trace of the same bench command.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -5 gpurun_out/pytest_gpu.log
( timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err )
cat gpurun_out/bench.json
rm -rf gpurun_out/prof_stats
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_stats -o ks -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_stats.log 2>&1 )
find gpurun_out/prof_stats -name "*.db" | head -3
DB=$(find gpurun_out/prof_stats -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_summary.py stats $DB > gpurun_out/kernel_stats.txt 2>&1 && head -40 gpurun_out/kernel_stats.txt
find gpurun_out/prof_stats -name "*.db" -size +20M -delete
