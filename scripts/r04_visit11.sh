#!/bin/bash
mkdir -p gpurun_out/v11
O=gpurun_out/v11
python -m memvul_amd.build > /dev/null || exit 1
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v['avg_us'],1) for k, v in d['kernels'].items()})"; }
for rep in 1 2; do
for MODE in f16 precise; do
  Q="--compute $MODE --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --streams 1 --steps 10 --warmup 3"
  echo -n "$MODE S256: "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  echo -n "$MODE S512: "; timeout 300 python bench.py $Q --seq-len 512 --batch 128 2>/dev/null | tail -1 | one
done; done > $O/attn_o_image_ab.txt 2>&1; cat $O/attn_o_image_ab.txt
timeout 900 python -m pytest tests -m gpu -q -k "attention or layer0 or golden or parity or precise" > $O/pytest_attn.txt 2>&1; grep -E "passed|failed" $O/pytest_attn.txt
