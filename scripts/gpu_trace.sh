set -u
O=gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf $O/prof_stats
( cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats -d $R/$O/prof_stats -o ks -- python $R/bench.py --steps 10 --warmup 3 --cpu-sample 0 --streams 1 > $R/$O/prof_stats.log 2>&1 )
DB=$(find $O/prof_stats -name "*.db" | head -1)
python scripts/rocpd_summary.py stats $DB > $O/kernel_stats.txt 2>&1; head -30 $O/kernel_stats.txt
tail -1 $O/prof_stats.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline'])"
find $O -name "*.db" -size +20M -delete
