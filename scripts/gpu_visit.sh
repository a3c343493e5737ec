#!/bin/bash
# One parameterised GPU-box visit (replaces the one-shot scripts/r0[45]_visit*.sh of earlier rounds; their records live under profiles/).
#   usage: scripts/gpu_visit.sh <tag> <step> [<step> ...]        outputs: gpurun_out/<tag>/
# steps:
#   suite[:K]    the GPU test suite (pytest -m gpu; K = a -k expression)
#   smoke        __graft_entry__.smoke()
#   bench        the default bench line            driver     the driver's form (--gpus 1 --steps 20 --warmup 5)
#   quick        bench without the CPU leg / second engine (value + kernels only)
#   ab           bench --ab (adds the both-terms form of the precise mode)
#   pmc[:mode[:cfg]]  rocprofv3 kernel trace + counter passes (scripts/gpu_pmc.sh; mode precise|f16, cfg cfg2|cfg3) -> pmc_current.json
#   errdist      24-draw trained-like error distribution (scripts/r05_error_distribution.py)
#   envelope     matcher-norm / outlier envelope (scripts/r05_precision_envelope.py)        lengths  sequence-length axis (scripts/r05_length_envelope.py)
#   sink[:args]  attention-concentration axis (scripts/r06_sink_envelope.py; args e.g. "--only sep_all_80")
#   configs      other configurations (scripts/gpu_configs.sh)      e2e[:args]  the drop-in end to end (scripts/r06_e2e_dropin.py [n] [record workers])
#   cmd:<shell>  anything else
set -u
TAG=${1:?tag}; shift
O=gpurun_out; V=$O/$TAG
mkdir -p $V
export TMPDIR=/tmp
python -m memvul_amd.build > $V/build.log 2>&1 || { echo BUILD FAILED; tail -20 $V/build.log; exit 1; }
cat memvul_amd/lib/libmemvul_hip.so.stamp > $V/lib_stamp.txt
line() {  # one-line digest of a bench JSON line
python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "IR/s", d["value"], "ms/step", d["ms_per_step"], "roofline", d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("avg_launch_us"),
          "err", d.get("logit_max_abs_err_trained_like"), "1-in-flight", d.get("value_one_batch_in_flight"), "sustained", (d.get("value_sustained") or {}).get("value"),
          "fast", d.get("fast", {}).get("value"), "cls_off", d.get("precise_cls_aside_off", {}).get("value"), "cfg3", d.get("cfg3", {}).get("value"),
          "cpu", d.get("cpu_baseline", {}).get("value"), "sat", d.get("x8_saturated_elements"), "kernels", {k: v["avg_us"] for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print(f, "FAILED", e)
    try:
        print(open(f.replace(".json", ".err")).read()[-1500:])
    except Exception:
        pass
PY
}
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  echo "== $step ($(date +%H:%M:%S))"
  case $name in
    suite) if [ -n "$arg" ]; then ( timeout 1500 python -m pytest tests -m gpu -q -k "$arg" > $V/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $V/pytest_gpu.log )
           else ( timeout 1500 python -m pytest tests -m gpu -q > $V/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $V/pytest_gpu.log ); fi
           grep -E "passed|failed|^FAILED|^ERROR|rc=" $V/pytest_gpu.log | tail -15 ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $V/smoke.txt 2>&1; tail -2 $V/smoke.txt ;;
    bench) ( T0=$(date +%s); timeout 600 python bench.py > $V/bench_line.json 2> $V/bench_line.err; echo "rc=$? wall $(( $(date +%s) - T0 )) s" >> $V/bench_line.err ); line $V/bench_line.json; tail -1 $V/bench_line.err ;;
    driver) ( T0=$(date +%s); timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $V/bench_line_driver_form.json 2> $V/bench_line_driver_form.err; echo "rc=$? wall $(( $(date +%s) - T0 )) s" >> $V/bench_line_driver_form.err ); line $V/bench_line_driver_form.json; tail -1 $V/bench_line_driver_form.err ;;
    quick) ( timeout 300 python bench.py --cpu-sample 0 --no-second --matcher-anchors 0 --sustain-s 0 $arg > $V/bench_quick.json 2> $V/bench_quick.err ); line $V/bench_quick.json ;;
    ab) ( timeout 900 python bench.py --ab > $V/bench_line_ab.json 2> $V/bench_line_ab.err ); line $V/bench_line_ab.json ;;
    pmc) m=${arg%%:*}; c=cfg2; [ "$arg" != "$m" ] && c=${arg#*:}; [ -z "$m" ] && m=precise
         bash scripts/gpu_pmc.sh r06 $m $c > $V/pmc_${m}_${c}.log 2>&1; tail -3 $V/pmc_${m}_${c}.log ;;
    errdist) timeout 700 python scripts/r05_error_distribution.py --f16-seeds 0 --json $V/error_distribution.json > $V/error_distribution.txt 2>&1; tail -2 $V/error_distribution.txt ;;
    envelope) timeout 500 python scripts/r05_precision_envelope.py $V/precision_envelope.json > $V/precision_envelope.txt 2>&1; grep -v amdgpu.ids $V/precision_envelope.txt | tail -12 ;;
    lengths) timeout 500 python scripts/r05_length_envelope.py > $V/length_envelope.txt 2>&1; tail -12 $V/length_envelope.txt ;;
    sink) timeout 1500 python scripts/r06_sink_envelope.py --json $V/sink_envelope.json $arg > $V/sink_envelope.txt 2>&1; grep -v amdgpu.ids $V/sink_envelope.txt | tail -14 ;;
    configs) bash scripts/gpu_configs.sh > $V/other_configs.txt 2>&1; tail -12 $V/other_configs.txt ;;
    e2e) timeout 1500 python scripts/r06_e2e_dropin.py $arg > $V/e2e_dropin.txt 2>&1; tail -20 $V/e2e_dropin.txt ;;
    cmd) bash -c "$arg" ;;
    *) echo "unknown step $name" ;;
  esac
done
echo "== done ($(date +%H:%M:%S))"
