#!/bin/bash
mkdir -p gpurun_out/v10
O=gpurun_out/v10
python -m memvul_amd.build > /dev/null || exit 1
Q="--cpu-sample 0 --sustain-s 0 --no-second --matcher-anchors 0 --steps 16 --warmup 4"
for MODE in precise f16; do
  timeout 300 python bench.py --compute $MODE $Q --ragged 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$MODE ragged', d['value'], d['ragged'])"
  timeout 300 python bench.py --compute $MODE $Q --seq-len 128 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$MODE S128', d['value'], d['e2e_mfma_frac'])"
  timeout 300 python bench.py --compute $MODE $Q --anchors 1000 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$MODE G1000', d['value'], d['kernels']['match'])"
done > $O/other_configs.txt 2>&1; cat $O/other_configs.txt
