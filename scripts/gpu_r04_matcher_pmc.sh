#!/bin/bash
# Round 4: the counters that say what bounds the fused matcher at configs[4] (B = 256, G = 1000, k = 10):
# VALU instructions / VALU-active cycles / LDS-issue stalls / LDS instructions / waves, of tools/match_probe (the shipped kernels).
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ -x tools/match_probe ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/match_probe.hip -o tools/match_probe
rm -rf $O/p_match_a $O/p_match_b
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $R/$O/p_match_a -o sq -- $R/tools/match_probe > $R/$O/p_match_a.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $R/$O/p_match_b -o sq -- $R/tools/match_probe > $R/$O/p_match_b.log 2>&1 )
for p in a b; do
  DBS=$(find $O/p_match_$p -name "*.db" | tr '\n' ' ')
  [ -n "$DBS" ] && python scripts/rocpd_summary.py pmc $DBS > $O/r04_match_pmc_$p.txt 2>&1 && head -14 $O/r04_match_pmc_$p.txt
done
tail -20 $O/p_match_a.log
find $O -name "*.db" -size +1M -delete
