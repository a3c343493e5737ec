#!/bin/bash
# A/B: order of the LDS-DMA issue against the fragment reads inside a barrier interval of gemm_pp (v1: DMA first in both wave groups,
# v2: in the fragments-first group only, v3: in the matrix-pipe-first group only).  Functionally neutral; checked on v1.
Q="--matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-precise"
P=$PWD/tools/probe_order
MEMVUL_HIP_LIB=$P/libpp_order_v1.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "persistent or layer0" 2>&1 | tail -2
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_one_batch_in_flight'], {k: v['avg_us'] for k, v in d['kernels'].items() if k.startswith('gemm')})"; }
for rep in 1 2; do
  echo -n "base: "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  for v in 1 2 3; do echo -n "v$v:   "; MEMVUL_HIP_LIB=$P/libpp_order_v$v.so timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one; done
done
echo -n "base precise: "; timeout 300 python bench.py $Q --compute precise 2>/dev/null | tail -1 | one
echo -n "v1   precise: "; MEMVUL_HIP_LIB=$P/libpp_order_v1.so timeout 300 python bench.py $Q --compute precise 2>/dev/null | tail -1 | one
