#!/bin/bash
# Round 5, final evidence visit on the shipped library (the [CLS]-row form as default): full GPU suite, smoke, the bench line in the driver's form, rocprofv3 kernel trace
# + counter passes of the precise mode at cfg 2 -> pmc_current.json, the 24-draw error distribution and the precision envelope.
set -u
O=gpurun_out
V=$O/r05_v17
mkdir -p $V
export TMPDIR=/tmp
python -m memvul_amd.build > $V/build.log 2>&1 || { echo BUILD FAILED; tail -20 $V/build.log; exit 1; }
cat memvul_amd/lib/libmemvul_hip.so.stamp > $V/lib_stamp.txt
( timeout 900 python -m pytest tests -m gpu -q > $V/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $V/pytest_gpu.log ); tail -5 $V/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $V/smoke.txt 2>&1; tail -2 $V/smoke.txt
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $V/bench_line_driver_form.json 2> $V/bench_line_driver_form.err; echo "rc=$?" >> $V/bench_line_driver_form.err )
python - <<'PY'
import json
for f in ("bench_line_driver_form",):
    try:
        d = json.loads(open("gpurun_out/r05_v17/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "IR/s", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "err", d.get("logit_max_abs_err_trained_like"),
              "fast", d.get("fast", {}).get("value"), "lo8", d.get("precise_lo8_stream", {}).get("value"), d.get("precise_lo8_stream", {}).get("logit_max_abs_err_trained_like"),
              "cls_off", d.get("precise_cls_aside_off", {}).get("value"), d.get("precise_cls_aside_off", {}).get("logit_max_abs_err_trained_like"),
              "cfg3", d.get("cfg3", {}).get("value"), "cpu", d.get("cpu_baseline", {}).get("value"), "sat", d.get("x8_saturated_elements"), "kernels", {k: v["avg_us"] for k, v in d.get("kernels", {}).items()})
    except Exception as e:
        print(f, "FAILED", e); print(open("gpurun_out/r05_v17/%s.err" % f).read()[-1500:])
PY
bash scripts/gpu_pmc.sh r05 precise cfg2 > $V/pmc_precise_cfg2.log 2>&1; tail -3 $V/pmc_precise_cfg2.log
timeout 600 python scripts/r05_error_distribution.py --f16-seeds 0 --json $V/error_distribution.json > $V/error_distribution.txt 2>&1; tail -2 $V/error_distribution.txt
timeout 400 python scripts/r05_precision_envelope.py $V/precision_envelope.json > $V/precision_envelope.txt 2>&1; grep -v amdgpu.ids $V/precision_envelope.txt | tail -12
