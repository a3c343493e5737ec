#!/bin/bash
Q="--matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-precise"
P=$PWD/tools/probe_order
MEMVUL_HIP_LIB=$P/libpp_order_v4.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "persistent" 2>&1 | tail -1
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_one_batch_in_flight'], {k: v['avg_us'] for k, v in d['kernels'].items() if k.startswith('gemm')})"; }
for rep in 1 2 3; do
  echo -n "base: "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  echo -n "v4:   "; MEMVUL_HIP_LIB=$P/libpp_order_v4.so timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
done
