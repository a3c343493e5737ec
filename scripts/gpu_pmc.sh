#!/bin/bash
# Counter evidence behind the SHIPPED kernels, one batch in flight: rocprofv3 kernel trace + SQ / GRBM pass + two TCC passes
# (FETCH_SIZE, WRITE_SIZE), one counter group per run (never combined with the trace domains gpurun refuses).
#   usage: scripts/gpu_pmc.sh [tag] [mode: precise|f16] [cfg: cfg2|cfg3]
# cfg2 = the default bench workload (S=256, B=256): writes the text summaries AND merges this mode's classes into
# gpurun_out/pmc_current.json (traffic / MFMA-busy / clock with the stamp of the library profiled) — copy that to
# profiles/pmc_current.json: bench.py reports the figures only while the stamp matches the loaded library.
# cfg3 = BASELINE.json configs[2] (S=512, B=128): text summaries only.
set -u
TAG=${1:-r04}
MODE=${2:-precise}
CFG=${3:-cfg2}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
note() { echo "== $* ($(date +%H:%M:%S))"; }
python -m memvul_amd.build > /dev/null || exit 1   # no-op when the binary that travelled matches the sources; never profile a stale one
SHAPE=""
[ "$CFG" = "cfg3" ] && SHAPE="--seq-len 512 --batch 128"
COMMON="--compute $MODE --cpu-sample 0 --sustain-s 0 --matcher-anchors 0 --streams 1 --no-second $SHAPE"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P=$O/p_${MODE}_${CFG}
rm -rf ${P}_stats ${P}_sq ${P}_fetch ${P}_write
note "kernel trace ($MODE $CFG)"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/${P}_stats -o ks -- python $R/bench.py --steps 8 --warmup 3 $COMMON > $R/${P}_stats.log 2>&1 )
DB=$(find ${P}_stats -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_summary.py stats $DB > $O/${TAG}_${CFG}_${MODE}_kernel_stats_one_batch_in_flight.txt 2>&1 && head -16 $O/${TAG}_${CFG}_${MODE}_kernel_stats_one_batch_in_flight.txt
tail -1 ${P}_stats.log > $O/${TAG}_${CFG}_${MODE}_bench_line_under_profiler.json
note "SQ / GRBM counters"
( cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc $SQ -d $R/${P}_sq -o sq -- python $R/bench.py --steps 2 --warmup 1 --no-profile $COMMON > $R/${P}_sq.log 2>&1 )
note "TCC counters (HBM traffic)"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/${P}_fetch -o pf -- python $R/bench.py --steps 2 --warmup 1 --no-profile $COMMON > $R/${P}_fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/${P}_write -o pw -- python $R/bench.py --steps 2 --warmup 1 --no-profile $COMMON > $R/${P}_write.log 2>&1 )
DBS=$(find ${P}_sq -name "*.db" | tr '\n' ' ')
[ -n "$DBS" ] && python scripts/rocpd_summary.py pmc $DBS > $O/${TAG}_${CFG}_${MODE}_pmc_sq_grbm.txt 2>&1 && head -12 $O/${TAG}_${CFG}_${MODE}_pmc_sq_grbm.txt
DBH=$(find ${P}_fetch ${P}_write -name "*.db" | tr '\n' ' ')
[ -n "$DBH" ] && python scripts/rocpd_summary.py pmc $DBH > $O/${TAG}_${CFG}_${MODE}_pmc_hbm.txt 2>&1 && head -10 $O/${TAG}_${CFG}_${MODE}_pmc_hbm.txt
if [ "$CFG" = "cfg2" ]; then
  python scripts/rocpd_summary.py json memvul_amd/lib/libmemvul_hip.so.stamp --mode $MODE --into $O/pmc_current.json $DBS $DBH > $O/pmc_current.new 2> $O/pmc_current.err && mv $O/pmc_current.new $O/pmc_current.json && cat $O/pmc_current.json
fi
find $O -name "*.db" -size +1M -delete
note done
