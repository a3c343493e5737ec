#!/bin/bash
# Counter evidence behind the SHIPPED kernels at the default bench workload (cfg 2: S=256, B=256), one batch in flight:
# rocprofv3 kernel trace + SQ / GRBM pass + two TCC passes (FETCH_SIZE, WRITE_SIZE), one counter group per run (never combined
# with the trace domains gpurun refuses).  Writes the text summaries AND gpurun_out/pmc_current.json (the classes' traffic /
# MFMA-busy / clock with the stamp of the library profiled) — copy that to profiles/pmc_current.json: bench.py reports the
# figures only while the stamp matches the loaded library.   usage: scripts/gpu_pmc.sh [tag]
set -u
TAG=${1:-r03}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
note() { echo "== $* ($(date +%H:%M:%S))"; }
python -m memvul_amd.build > /dev/null || exit 1   # no-op when the binary that travelled matches the sources; never profile a stale one
COMMON="--cpu-sample 0 --sustain-s 0 --matcher-anchors 0 --streams 1 --no-precise"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
rm -rf $O/p_stats $O/p_sq $O/p_fetch $O/p_write
note "kernel trace"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/p_stats -o ks -- python $R/bench.py --steps 8 --warmup 3 $COMMON > $R/$O/p_stats.log 2>&1 )
DB=$(find $O/p_stats -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_summary.py stats $DB > $O/${TAG}_cfg2_kernel_stats_one_batch_in_flight.txt 2>&1 && head -16 $O/${TAG}_cfg2_kernel_stats_one_batch_in_flight.txt
tail -1 $O/p_stats.log > $O/${TAG}_cfg2_bench_line_under_profiler.json
note "SQ / GRBM counters"
( cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc $SQ -d $R/$O/p_sq -o sq -- python $R/bench.py --steps 2 --warmup 1 --no-profile $COMMON > $R/$O/p_sq.log 2>&1 )
note "TCC counters (HBM traffic)"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/p_fetch -o pf -- python $R/bench.py --steps 2 --warmup 1 --no-profile $COMMON > $R/$O/p_fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/p_write -o pw -- python $R/bench.py --steps 2 --warmup 1 --no-profile $COMMON > $R/$O/p_write.log 2>&1 )
DBS=$(find $O/p_sq -name "*.db" | tr '\n' ' ')
[ -n "$DBS" ] && python scripts/rocpd_summary.py pmc $DBS > $O/${TAG}_cfg2_pmc_sq_grbm.txt 2>&1 && head -12 $O/${TAG}_cfg2_pmc_sq_grbm.txt
DBH=$(find $O/p_fetch $O/p_write -name "*.db" | tr '\n' ' ')
[ -n "$DBH" ] && python scripts/rocpd_summary.py pmc $DBH > $O/${TAG}_cfg2_pmc_hbm.txt 2>&1 && head -10 $O/${TAG}_cfg2_pmc_hbm.txt
python scripts/rocpd_summary.py json memvul_amd/lib/libmemvul_hip.so.stamp $DBS $DBH > $O/pmc_current.json 2> $O/pmc_current.err && cat $O/pmc_current.json
find $O -name "*.db" -size +1M -delete
note done
