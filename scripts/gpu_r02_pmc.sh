#!/bin/bash
# Round 2: counter evidence behind the SHIPPED kernels: rocprofv3 kernel trace + SQ / GRBM / TCC
# counter passes of the bench command at cfg 2 (S=256, B=256) and cfg 3 (S=512, B=128), one batch in flight (clean
# per-kernel durations), one counter group per run (never combined with the trace domains gpurun refuses).
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
note() { echo "== $* ($(date +%H:%M:%S))"; }
COMMON="--cpu-sample 0 --sustain-s 0 --matcher-anchors 0 --streams 1"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
run_cfg() {  # tag, bench args...
  local tag=$1; shift
  note "$tag: kernel trace"
  rm -rf $O/p_${tag}_stats $O/p_${tag}_sq $O/p_${tag}_fetch $O/p_${tag}_write
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/p_${tag}_stats -o ks -- python $R/bench.py --steps 8 --warmup 3 $COMMON "$@" > $R/$O/p_${tag}_stats.log 2>&1 )
  DB=$(find $O/p_${tag}_stats -name "*.db" | head -1)
  [ -n "$DB" ] && python scripts/rocpd_summary.py stats $DB > $O/r02_${tag}_kernel_stats.txt 2>&1 && head -24 $O/r02_${tag}_kernel_stats.txt
  note "$tag: SQ / GRBM counters"
  ( cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc $SQ -d $R/$O/p_${tag}_sq -o sq -- python $R/bench.py --steps 2 --warmup 1 --no-profile $COMMON "$@" > $R/$O/p_${tag}_sq.log 2>&1 )
  DBS=$(find $O/p_${tag}_sq -name "*.db" | tr '\n' ' ')
  [ -n "$DBS" ] && python scripts/rocpd_summary.py pmc $DBS > $O/r02_${tag}_pmc_sq.txt 2>&1 && head -16 $O/r02_${tag}_pmc_sq.txt
  if [ "${MEMVUL_PMC_HBM:-1}" = "1" ]; then
    note "$tag: TCC counters (HBM traffic)"
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/p_${tag}_fetch -o pf -- python $R/bench.py --steps 2 --warmup 1 --no-profile $COMMON "$@" > $R/$O/p_${tag}_fetch.log 2>&1 )
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/p_${tag}_write -o pw -- python $R/bench.py --steps 2 --warmup 1 --no-profile $COMMON "$@" > $R/$O/p_${tag}_write.log 2>&1 )
    DBS=$(find $O/p_${tag}_fetch $O/p_${tag}_write -name "*.db" | tr '\n' ' ')
    [ -n "$DBS" ] && python scripts/rocpd_summary.py pmc $DBS > $O/r02_${tag}_pmc_hbm.txt 2>&1 && head -14 $O/r02_${tag}_pmc_hbm.txt
  fi
  find $O -name "*.db" -size +1M -delete
}
run_cfg cfg2
MEMVUL_PMC_HBM=0 run_cfg cfg3 --seq-len 512 --batch 128
note done
