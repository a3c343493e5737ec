#!/bin/bash
# A/B of s_setprio placement in gemm_pp's interval: base = priority 1 around the first wave group's MFMA block only;
# p0 = no s_setprio; p1 = priority 1 around BOTH groups' MFMA blocks; p2 = priority 1 around the fragment reads + DMA issue instead.
Q="--matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-precise --streams 1"
P=$PWD/tools/probe_prio
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['avg_us'] for k, v in d['kernels'].items() if k.startswith('gemm')})"; }
MEMVUL_HIP_LIB=$P/libpp_p2.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "persistent" 2>&1 | tail -1
for rep in 1 2; do
  echo -n "base: "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  for v in p0 p1 p2; do echo -n "$v:   "; MEMVUL_HIP_LIB=$P/libpp_$v.so timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one; done
done
