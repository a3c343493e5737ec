#!/bin/bash
# Round 5, visit 8: (1) rocprofv3 kernel trace + HBM-traffic counter passes of the precise mode WITH the opt-in lo8 residual stream (what the bytes of the
# residual GEMMs become), (2) the other configurations on the shipped binary: ragged corpus (padded / length-bucketed), S = 128, 1 000 anchors, in the default mode.
set -u
O=gpurun_out
V=$O/r05_v8
mkdir -p $V
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
python -m memvul_amd.build > /dev/null || exit 1
COMMON="--compute precise --cpu-sample 0 --sustain-s 0 --matcher-anchors 0 --streams 1 --no-second"
export MEMVUL_STREAM_LO8=1
P=$O/p_lo8
rm -rf ${P}_stats ${P}_fetch ${P}_write
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/${P}_stats -o ks -- python $R/bench.py --steps 8 --warmup 3 $COMMON > $R/${P}_stats.log 2>&1 )
DB=$(find ${P}_stats -name "*.db" | head -1)
[ -n "$DB" ] && python scripts/rocpd_summary.py stats $DB > $V/r05_cfg2_precise_lo8_stream_kernel_stats_one_batch_in_flight.txt 2>&1 && head -8 $V/r05_cfg2_precise_lo8_stream_kernel_stats_one_batch_in_flight.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/${P}_fetch -o pf -- python $R/bench.py --steps 2 --warmup 1 --no-profile $COMMON > $R/${P}_fetch.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/${P}_write -o pw -- python $R/bench.py --steps 2 --warmup 1 --no-profile $COMMON > $R/${P}_write.log 2>&1 )
DBH=$(find ${P}_fetch ${P}_write -name "*.db" | tr '\n' ' ')
[ -n "$DBH" ] && python scripts/rocpd_summary.py pmc $DBH > $V/r05_cfg2_precise_lo8_stream_pmc_hbm.txt 2>&1 && head -10 $V/r05_cfg2_precise_lo8_stream_pmc_hbm.txt
unset MEMVUL_STREAM_LO8
find $O -name "*.db" -size +1M -delete
Q="--cpu-sample 0 --sustain-s 0 --no-second --matcher-anchors 0 --steps 20 --warmup 5"
{
for MODE in precise f16; do
  echo -n "$MODE ragged (lengths uniform in [16, 256]) : "; timeout 300 python bench.py --compute $MODE --ragged $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ragged'])"
  echo -n "$MODE S=128 B=256 : "; timeout 300 python bench.py --compute $MODE --seq-len 128 $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['e2e_mfma_frac'])"
  echo -n "$MODE 1000 anchors : "; timeout 300 python bench.py --compute $MODE --anchors 1000 $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels'].get('match'), d['kernels'].get('topk'))"
done; } > $V/other_configs.txt 2>&1; cat $V/other_configs.txt
