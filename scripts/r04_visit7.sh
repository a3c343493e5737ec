#!/bin/bash
# round-4 visit 7: the shipped binary — full GPU suite, counter passes (both modes at cfg 2, precise at cfg 3), default bench line
mkdir -p gpurun_out/v7
O=gpurun_out/v7
export TMPDIR=/tmp
python -m memvul_amd.build > /dev/null || exit 1
rm -f gpurun_out/diag.jsonl gpurun_out/pmc_current.json
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
cp gpurun_out/diag.jsonl $O/diag.jsonl 2>/dev/null
bash scripts/gpu_pmc.sh r04 precise cfg2 > $O/pmc_precise.log 2>&1; tail -2 $O/pmc_precise.log
bash scripts/gpu_pmc.sh r04 f16 cfg2 > $O/pmc_f16.log 2>&1; tail -2 $O/pmc_f16.log
bash scripts/gpu_pmc.sh r04 precise cfg3 > $O/pmc_cfg3.log 2>&1; tail -2 $O/pmc_cfg3.log
cp gpurun_out/pmc_current.json profiles/pmc_current.json
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; tail -c 300 $O/bench_driver_form.json
