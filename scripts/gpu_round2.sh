#!/bin/bash
# One GPU-box visit (round 1, second pass): fail-fast parity of the new kernels, A/B benches of every engine switch,
# the full GPU suite, rocprofv3 kernel trace + PMC (HBM traffic) passes of the bench command, cfg 3 / cfg 5 lines.
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
note() { echo "== $* ($(date +%H:%M:%S))"; }

note "hang guard: one tiny attention_v2 case under a short timeout"
timeout 240 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "layer0_stages and 2-64-False" > $O/t_guard.log 2>&1
GRC=$?
echo "guard rc=$GRC" >> $O/t_guard.log; tail -3 $O/t_guard.log
if [ $GRC -eq 124 ]; then
  echo "attention_v2 timed out: forcing MEMVUL_ATTN=0 for the rest of this visit" | tee -a $O/t_guard.log
  export MEMVUL_FORCE_ATTN=0 MEMVUL_ATTN=0
fi

note "full GPU suite"
( timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -15 $O/pytest_gpu.log

bench() {  # name, env assignments..., then -- bench args
  local name=$1; shift
  local envs=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
  [ $# -gt 0 ] && shift
  ( env "${envs[@]}" timeout 300 python bench.py --cpu-sample 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "rc=$?" >> $O/bench_$name.err )
  python - "$O/bench_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d.get("kernels", {})
    print(sys.argv[2], "IR/s", d["value"], "ms/step", d["ms_per_step"], "| " + " ".join(f"{n}={v['avg_us']}" for n, v in k.items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
note "A/B benches"
bench default MEMVUL_X=1
if [ "${MEMVUL_VISIT:-full}" = "full" ]; then  # MEMVUL_VISIT=short: default + one_stream only (saves ~2.5 GPU-minutes)
bench attn_v1 MEMVUL_ATTN=0
bench res_f32 MEMVUL_RES_HILO=0
bench ln_explicit MEMVUL_LN_VIRTUAL=0
bench no_lnfuse MEMVUL_LN_FUSE=0
bench no_prune MEMVUL_CLS_PRUNE=0
bench all_off MEMVUL_ATTN=0 MEMVUL_LN_FUSE=0 MEMVUL_CLS_PRUNE=0 -- --streams 1
bench noprof MEMVUL_X=1 -- --no-profile
fi
bench one_stream MEMVUL_X=1 -- --streams 1
bench ragged MEMVUL_X=1 -- --ragged --steps 10

note "rocprofv3 kernel trace of the bench command"
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
( cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats -d $R/$O/prof_stats -o ks -- python $R/bench.py --steps 10 --warmup 3 --cpu-sample 0 --streams 1 > $R/$O/prof_stats.log 2>&1 )
DB=$(find $O/prof_stats -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_summary.py stats $DB > $O/kernel_stats.txt 2>&1 && head -34 $O/kernel_stats.txt

note "PMC passes (HBM traffic), one counter group per run"
( cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/prof_fetch -o pf -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-profile --streams 1 > $R/$O/prof_fetch.log 2>&1 )
( cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/prof_write -o pw -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-profile --streams 1 > $R/$O/prof_write.log 2>&1 )
DBS=$(find $O/prof_fetch $O/prof_write -name "*.db" | tr '\n' ' ')
[ -n "$DBS" ] && python scripts/rocpd_summary.py pmc $DBS > $O/pmc_hbm.txt 2>&1 && head -30 $O/pmc_hbm.txt

note "cfg 3 (S=512, B=128) and cfg 5 (G=1000 anchors)"
bench cfg3_s512 MEMVUL_X=1 -- --seq-len 512 --batch 128 --steps 20
bench cfg5_g1000 MEMVUL_X=1 -- --anchors 1000 --anchor-len 128 --steps 10

note "contract line (with the CPU baseline)"
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); cat $O/bench.json
find $O -name "*.db" -size +20M -delete
note done
