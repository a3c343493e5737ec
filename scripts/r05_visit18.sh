#!/bin/bash
# Round 5, visit 18: counter passes of the shipped library for the other two records — precise at cfg 3 (S 512, B 128) and f16 at cfg 2 (merged into pmc_current.json).
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cp profiles/pmc_current.json $O/pmc_current.json
bash scripts/gpu_pmc.sh r05 precise cfg3 > $O/r05_v18_pmc_precise_cfg3.log 2>&1; tail -2 $O/r05_v18_pmc_precise_cfg3.log
bash scripts/gpu_pmc.sh r05 f16 cfg2 > $O/r05_v18_pmc_f16_cfg2.log 2>&1; tail -2 $O/r05_v18_pmc_f16_cfg2.log
