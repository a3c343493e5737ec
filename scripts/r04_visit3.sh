#!/bin/bash
# round-4 visit 3: raster A/B with fetch counters, full GPU suite, counter passes of both modes + cfg 3, matcher counters, default bench line
mkdir -p gpurun_out/v3
O=gpurun_out/v3
export TMPDIR=/tmp
R=$(pwd)
python -m memvul_amd.build > /dev/null || exit 1
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v['avg_us'],1) for k, v in d['kernels'].items() if k.startswith('gemm')})"; }
for MODE in f16 precise; do
  Q="--compute $MODE --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --streams 1 --steps 10 --warmup 3"
  for rep in 1 2; do
    echo -n "$MODE base        : "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
    echo -n "$MODE raster=1    : "; MEMVUL_RASTER=1 timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
    echo -n "$MODE gn<=6       : "; MEMVUL_GN_MAX=6 timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
    echo -n "$MODE gn<=12      : "; MEMVUL_GN_MAX=12 timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
    echo -n "$MODE raster=1,gn6: "; MEMVUL_RASTER=1 MEMVUL_GN_MAX=6 timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  done
done > $O/raster_ab.txt 2>&1
cat $O/raster_ab.txt
# fetch / clock counters of the raster variants (f16), one counter group per pass
QP="--compute f16 --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --streams 1 --steps 2 --warmup 1 --no-profile"
for V in base raster1 gn12; do
  case $V in base) E="";; raster1) E="MEMVUL_RASTER=1";; gn12) E="MEMVUL_GN_MAX=12";; esac
  rm -rf $O/p_$V
  ( cd /tmp && env $E timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/p_$V/f -o pf -- python $R/bench.py $QP > /dev/null 2>&1 )
  ( cd /tmp && env $E timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $R/$O/p_$V/s -o ps -- python $R/bench.py $QP > /dev/null 2>&1 )
  DBS=$(find $O/p_$V -name "*.db" | tr '\n' ' ')
  echo "== $V"; python scripts/rocpd_summary.py pmc $DBS 2>&1 | grep -v "^#" | cut -c1-200 | head -8
done > $O/raster_fetch_pmc.txt 2>&1
cat $O/raster_fetch_pmc.txt
find $O -name "*.db" -delete
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
cp gpurun_out/diag.jsonl $O/diag.jsonl 2>/dev/null
bash scripts/gpu_pmc.sh r04 precise cfg2 > $O/pmc_precise.log 2>&1; tail -3 $O/pmc_precise.log
bash scripts/gpu_pmc.sh r04 f16 cfg2 > $O/pmc_f16.log 2>&1; tail -3 $O/pmc_f16.log
bash scripts/gpu_pmc.sh r04 precise cfg3 > $O/pmc_cfg3.log 2>&1; tail -3 $O/pmc_cfg3.log
bash scripts/gpu_r04_matcher_pmc.sh > $O/matcher_pmc.log 2>&1; tail -5 $O/matcher_pmc.log
cp gpurun_out/pmc_current.json profiles/pmc_current.json 2>/dev/null
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
