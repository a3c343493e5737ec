#!/bin/bash
# Round 5, visit 10: the whole GPU suite under the opt-in switches (what a user who sets them can rely on).
set -u
O=gpurun_out/r05_v10
mkdir -p $O
export TMPDIR=/tmp
python -m memvul_amd.build > /dev/null 2>&1 || exit 1
( MEMVUL_STREAM_LO8=1 timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu_lo8.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_lo8.log ); grep -E "passed|failed|^FAILED" $O/pytest_gpu_lo8.log | tail -12
( MEMVUL_COMPUTE=f16 timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu_f16_default.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_f16_default.log ); grep -E "passed|failed|^FAILED" $O/pytest_gpu_f16_default.log | tail -12
