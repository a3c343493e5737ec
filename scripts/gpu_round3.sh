#!/bin/bash
# One GPU-box visit of round 3: focused kernel tests first (fast signal on the new fp8 path), the whole GPU suite, the default
# bench line (with its `precise` object), optional extra commands given as arguments.  Every step under its own timeout.
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
note() { echo "== $* ($(date +%H:%M:%S))"; }
python -m memvul_amd.build > /dev/null || exit 1   # no-op when the binary that travelled matches the sources; never profile a stale one
note "focused: persistent GEMM (fp16 / fp8 correction sweep), tile variants"
( timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "persistent_gemm or gemm_variants" > $O/pytest_focus.log 2>&1; echo "rc=$?" >> $O/pytest_focus.log ); tail -15 $O/pytest_focus.log
note "GPU suite"
( timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -25 $O/pytest_gpu.log
note "default bench line"
( timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err )
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
    print("default IR/s", d["value"], "ms/step", d["ms_per_step"], "| " + " ".join(f"{n}={v['avg_us']}" for n, v in d.get("kernels", {}).items()))
    print("roofline", d.get("roofline", {}).get("frac"), "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("with_host_loop"), "logit err", d.get("logit_max_abs_err_vs_cpu"))
    p = d.get("precise", {})
    print("precise IR/s", p.get("value"), "trained-like err precise / f16:", p.get("logit_max_abs_err_trained_like"), p.get("logit_max_abs_err_trained_like_f16"), p.get("kernels_avg_us"))
except Exception as e:
    print("bench FAILED", e); print(open("gpurun_out/bench_default.err").read()[-3000:])
PY
for cmd in "$@"; do
  note "$cmd"
  bash -c "$cmd"
done
note done
