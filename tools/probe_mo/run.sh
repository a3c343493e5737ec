#!/bin/bash
# A/B: issue order of the 16 MFMAs of a quadrant (base: K-step > column block > token block; o1: K-step > token block > column block;
# o2: base with the token blocks walked back and forth so that consecutive MFMAs always share an operand register)
Q="--matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-precise --streams 1"
P=$PWD/tools/probe_mo
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['avg_us'] for k, v in d['kernels'].items() if k.startswith('gemm')})"; }
for v in o1 o2; do MEMVUL_HIP_LIB=$P/libpp_$v.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "persistent" 2>&1 | tail -1; done
for rep in 1 2 3; do
  echo -n "base: "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  for v in o1 o2; do echo -n "$v:   "; MEMVUL_HIP_LIB=$P/libpp_$v.so timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one; done
done
