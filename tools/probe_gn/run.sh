#!/bin/bash
# A/B: raster group width of the persistent GEMM (tiles of GN consecutive N-tiles share an XCD's L2): 4 (shipped), 6, 12
Q="--matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-precise --streams 1"
P=$PWD/tools/probe_gn
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['avg_us'] for k, v in d['kernels'].items() if k.startswith('gemm')})"; }
for rep in 1 2; do
  echo -n "GN<=4:  "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  for v in 6 12; do echo -n "GN<=$v: "; MEMVUL_HIP_LIB=$P/libpp_g$v.so timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one; done
done
