#!/bin/bash
# Short GPU-box visit: GPU test suite, default bench line, optional extra commands given as arguments.
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -12 $O/pytest_gpu.log
( timeout 300 python bench.py --cpu-sample 0 > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?" >> $O/bench_default.err )
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
    print("default IR/s", d["value"], "ms/step", d["ms_per_step"], "| " + " ".join(f"{n}={v['avg_us']}" for n, v in d.get("kernels", {}).items()))
except Exception as e:
    print("bench FAILED", e); print(open("gpurun_out/bench_default.err").read()[-2000:])
PY
for cmd in "$@"; do
  echo "== $cmd"
  bash -c "$cmd"
done
