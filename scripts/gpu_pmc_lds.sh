#!/bin/bash
# LDS / instruction-mix counters of the default bench workload (separate --pmc passes, kernel trace only):
#   pass 1: LDS array activity and conflicts; pass 2: instruction counts by type; pass 3: L2 hit / miss.   usage: scripts/gpu_pmc_lds.sh [tag]
set -u
TAG=${1:-r03}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
note() { echo "== $* ($(date +%H:%M:%S))"; }
python -m memvul_amd.build > /dev/null || exit 1
COMMON="--cpu-sample 0 --sustain-s 0 --matcher-anchors 0 --streams 1 --no-precise --steps 2 --warmup 1 --no-profile"
rm -rf $O/p_lds $O/p_inst $O/p_l2
note "LDS counters"
( cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/$O/p_lds -o lds -- python $R/bench.py $COMMON > $R/$O/p_lds.log 2>&1 )
note "instruction mix"
( cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES -d $R/$O/p_inst -o inst -- python $R/bench.py $COMMON > $R/$O/p_inst.log 2>&1 )
note "L2"
( cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $R/$O/p_l2 -o l2 -- python $R/bench.py $COMMON > $R/$O/p_l2.log 2>&1 )
for p in lds inst l2; do
  DBS=$(find $O/p_$p -name "*.db" | tr '\n' ' ')
  [ -n "$DBS" ] && python scripts/rocpd_summary.py pmc $DBS > $O/${TAG}_cfg2_pmc_$p.txt 2>&1 && head -10 $O/${TAG}_cfg2_pmc_$p.txt
  tail -3 $O/p_$p.log | cut -c 1-300
done
find $O -name "*.db" -size +1M -delete
note done
