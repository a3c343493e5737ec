#!/bin/bash
# Round 5, final evidence visit on the committed tree: full GPU suite, smoke, the bench lines (default invocation + the driver's form), rocprofv3 kernel trace and
# counter passes (precise cfg 2, f16 cfg 2, precise cfg 3) -> profiles/pmc_current.json, the 24-draw error distribution and the precision envelope on the
# shipped binary, and the N > 1 entry of bench.py on a one-GPU box (the refusal, and the self-launched two-rank control flow with both ranks on the one GPU).
set -u
O=gpurun_out
V=$O/r05_v4
mkdir -p $V
export TMPDIR=/tmp
python -m memvul_amd.build > $V/build.log 2>&1 || { echo BUILD FAILED; tail -20 $V/build.log; exit 1; }
( timeout 1500 python -m pytest tests -m gpu -q > $V/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $V/pytest_gpu.log ); tail -6 $V/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $V/smoke.txt 2>&1; tail -2 $V/smoke.txt
( timeout 600 python bench.py > $V/bench_line.json 2> $V/bench_line.err; echo "rc=$?" >> $V/bench_line.err )
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $V/bench_line_driver_form.json 2> $V/bench_line_driver_form.err; echo "rc=$?" >> $V/bench_line_driver_form.err )
python - <<'PY'
import json
for f in ("bench_line", "bench_line_driver_form"):
    try:
        d = json.loads(open("gpurun_out/r05_v4/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "IR/s", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "err", d.get("logit_max_abs_err_trained_like"),
              "fast", d.get("fast", {}).get("value"), "lo8", d.get("precise_lo8_stream", {}).get("value"), d.get("precise_lo8_stream", {}).get("logit_max_abs_err_trained_like"),
              "cfg3", d.get("cfg3", {}).get("value"), "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"), d.get("cpu_baseline", {}).get("host_limits"),
              "sat", d.get("x8_saturated_elements"), "matcher", d.get("matcher", {}).get("avg_us"))
    except Exception as e:
        print(f, "FAILED", e); print(open("gpurun_out/r05_v4/%s.err" % f).read()[-1500:])
PY
bash scripts/gpu_pmc.sh r05 precise cfg2 > $V/pmc_precise_cfg2.log 2>&1; tail -3 $V/pmc_precise_cfg2.log
bash scripts/gpu_pmc.sh r05 f16 cfg2 > $V/pmc_f16_cfg2.log 2>&1; tail -3 $V/pmc_f16_cfg2.log
bash scripts/gpu_pmc.sh r05 precise cfg3 > $V/pmc_precise_cfg3.log 2>&1; tail -3 $V/pmc_precise_cfg3.log
timeout 900 python scripts/r05_error_distribution.py --json $V/error_distribution.json > $V/error_distribution.txt 2>&1; tail -4 $V/error_distribution.txt
timeout 600 python scripts/r05_precision_envelope.py $V/precision_envelope.json > $V/precision_envelope.txt 2>&1; grep -v amdgpu.ids $V/precision_envelope.txt
( timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 > $V/bench_gpus2_one_gpu_box.out 2> $V/bench_gpus2_one_gpu_box.err; echo "rc=$?" >> $V/bench_gpus2_one_gpu_box.err ); tail -2 $V/bench_gpus2_one_gpu_box.err
( MEMVUL_BENCH_ONE_GPU_SMOKE=rccl timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --shard-irs 2048 --cpu-sample 0 > $V/bench_self_launched_two_ranks_one_gpu.json 2> $V/bench_self_launched_two_ranks_one_gpu.err; echo "rc=$?" >> $V/bench_self_launched_two_ranks_one_gpu.err ); tail -1 $V/bench_self_launched_two_ranks_one_gpu.err; cut -c1-700 $V/bench_self_launched_two_ranks_one_gpu.json
