#!/bin/bash
mkdir -p gpurun_out/v9
O=gpurun_out/v9
python -m memvul_amd.build > /dev/null || exit 1
Q="--compute precise --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --streams 1 --steps 10 --warmup 3"
P=$PWD/tools/probe_x8half
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v['avg_us'],1) for k, v in d['kernels'].items() if k.startswith('gemm')})"; }
for rep in 1 2; do
  echo -n "base    : "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  for m in halfbar nobar8; do
    echo -n "$m : "; MEMVUL_HIP_LIB=$P/lib_$m.so timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  done
done > $O/sweep_sync_ablation.txt 2>&1; cat $O/sweep_sync_ablation.txt
