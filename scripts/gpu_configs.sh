#!/bin/bash
# The other BASELINE configs and the precise mode as bench lines + rocprofv3 evidence (one GPU-box visit).
#   cfg 3 (S = 512, B = 128), cfg 5 (1000 anchors), ragged corpus; kernel trace + SQ / GRBM counters of `--compute precise`.
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
note() { echo "== $* ($(date +%H:%M:%S))"; }
python -m memvul_amd.build > /dev/null || exit 1   # no-op when the binary that travelled matches the sources; never profile a stale one
Q="--cpu-sample 0 --sustain-s 0 --no-precise"
: > $O/other_configs_bench_lines.jsonl
note "cfg 3: S=512, B=128 (f16, precise)"
timeout 300 python bench.py --seq-len 512 --batch 128 --matcher-anchors 0 $Q 2>> $O/cfg.err | tail -1 >> $O/other_configs_bench_lines.jsonl
timeout 300 python bench.py --seq-len 512 --batch 128 --matcher-anchors 0 --compute precise $Q 2>> $O/cfg.err | tail -1 >> $O/other_configs_bench_lines.jsonl
note "cfg 5: 1000 anchors"
timeout 300 python bench.py --anchors 1000 $Q 2>> $O/cfg.err | tail -1 >> $O/other_configs_bench_lines.jsonl
note "ragged corpus"
timeout 300 python bench.py --ragged --matcher-anchors 0 $Q 2>> $O/cfg.err | tail -1 >> $O/other_configs_bench_lines.jsonl
note "precise as the headline mode"
timeout 300 python bench.py --compute precise --matcher-anchors 0 $Q 2>> $O/cfg.err | tail -1 >> $O/other_configs_bench_lines.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/other_configs_bench_lines.jsonl"):
    try:
        d = json.loads(l)
        print(d["dtype"][:12], d["config"]["seq_len"], d["config"]["global_batch"], d["config"]["anchors"], "IR/s", d["value"], "frac", d.get("roofline", {}).get("frac"),
              {k: v["avg_us"] for k, v in d.get("kernels", {}).items()}, d.get("ragged"))
    except Exception as e:
        print("bad line", e, l[:200])
PY
COMMON="--cpu-sample 0 --sustain-s 0 --matcher-anchors 0 --streams 1 --no-precise --compute precise"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
rm -rf $O/px_stats $O/px_sq
note "precise: kernel trace"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/px_stats -o ks -- python $R/bench.py --steps 8 --warmup 3 $COMMON > $R/$O/px_stats.log 2>&1 )
DB=$(find $O/px_stats -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_summary.py stats $DB > $O/r03_precise_kernel_stats_one_batch_in_flight.txt 2>&1 && head -14 $O/r03_precise_kernel_stats_one_batch_in_flight.txt
note "precise: SQ / GRBM counters"
( cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc $SQ -d $R/$O/px_sq -o sq -- python $R/bench.py --steps 2 --warmup 1 --no-profile $COMMON > $R/$O/px_sq.log 2>&1 )
DBS=$(find $O/px_sq -name "*.db" | tr '\n' ' ')
[ -n "$DBS" ] && python scripts/rocpd_summary.py pmc $DBS > $O/r03_precise_pmc_sq_grbm.txt 2>&1 && head -12 $O/r03_precise_pmc_sq_grbm.txt
find $O -name "*.db" -size +1M -delete
note done
