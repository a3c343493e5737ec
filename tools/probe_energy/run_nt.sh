#!/bin/bash
# A/B: cache policy of the operand LDS-DMA loads (functionally neutral): `nt` on the A-panel loads (streamed once per column group),
# on the W-panel loads (re-read by every tile of a column group), on both
Q="--matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-precise --streams 1"
P=$PWD/tools/probe_energy
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['avg_us'] for k, v in d['kernels'].items() if k.startswith('gemm')})"; }
MEMVUL_HIP_LIB=$P/libpp_ntA.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "persistent" 2>&1 | tail -1
for rep in 1 2; do
  echo -n "base: "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  for v in ntA ntW ntAW; do echo -n "$v: "; MEMVUL_HIP_LIB=$P/libpp_$v.so timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one; done
done
