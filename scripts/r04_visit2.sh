#!/bin/bash
# round-4 visit 2: GPU test suite on the new default (precise), the QKV weight-side-only sweep A/B, the new default bench line
mkdir -p gpurun_out/v2
O=gpurun_out/v2
python -m memvul_amd.build > /dev/null || exit 1
Q="--compute precise --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --streams 1 --steps 10 --warmup 3"
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v['avg_us'],1) for k, v in d['kernels'].items()})"; }
for rep in 1 2; do
  echo -n "qkv both terms : "; MEMVUL_QKV_X8_TERMS=2 timeout 300 python bench.py $Q 2>$O/err.txt | tail -1 | one
  echo -n "qkv w-side only: "; timeout 300 python bench.py $Q 2>$O/err.txt | tail -1 | one
done > $O/qkv_terms_ab.txt 2>&1
cat $O/qkv_terms_ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json; tail -5 $O/bench_default.err
cp gpurun_out/diag.jsonl $O/diag.jsonl 2>/dev/null
