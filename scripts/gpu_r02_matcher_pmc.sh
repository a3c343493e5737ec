#!/bin/bash
# Round 2: rocprofv3 evidence for the fused matcher (configs[4]): kernel trace + one SQ / GRBM counter pass of tools/match_probe.
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf $O/p_match_stats $O/p_match_sq
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/p_match_stats -o ks -- $R/tools/match_probe > $R/$O/p_match_stats.log 2>&1 )
DB=$(find $O/p_match_stats -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_summary.py stats $DB > $O/r02_match_kernel_stats.txt 2>&1 && head -20 $O/r02_match_kernel_stats.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $R/$O/p_match_sq -o sq -- $R/tools/match_probe > $R/$O/p_match_sq.log 2>&1 )
DBS=$(find $O/p_match_sq -name "*.db" | tr '\n' ' ')
[ -n "$DBS" ] && python scripts/rocpd_summary.py pmc $DBS > $O/r02_match_pmc_sq.txt 2>&1 && head -16 $O/r02_match_pmc_sq.txt
find $O -name "*.db" -size +1M -delete
