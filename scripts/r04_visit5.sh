#!/bin/bash
mkdir -p gpurun_out/v5
O=gpurun_out/v5
python -m memvul_amd.build > /dev/null || exit 1
timeout 900 python scripts/r04_qkv_terms_errors.py > $O/qkv_aside_errors.txt 2>&1; tail -6 $O/qkv_aside_errors.txt
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v['avg_us'],1) for k, v in d['kernels'].items()})"; }
Q="--compute precise --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --streams 1 --steps 10 --warmup 3"
for rep in 1 2; do
  for m in q none qkv v; do
    echo -n "precise, A-side blocks = $m: "; MEMVUL_QKV_ASIDE=$m timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  done
done > $O/precise_aside_ab.txt 2>&1; cat $O/precise_aside_ab.txt
timeout 900 python -m pytest tests -m gpu -q -k "precise or parity or layer0 or reference or golden" > $O/pytest_gpu_subset.txt 2>&1; tail -4 $O/pytest_gpu_subset.txt
