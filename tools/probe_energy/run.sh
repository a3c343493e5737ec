#!/bin/bash
# ENERGY-share ablations of the (power-bound) persistent GEMM, wrong results by construction:
#   noread = fragment ds_reads only during the first two K-tiles of a workgroup (the registers keep their values afterwards)
#   nodma  = no LDS-DMA after the prologue (the ring keeps its first contents)
Q="--matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-precise --streams 1"
P=$PWD/tools/probe_energy
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['avg_us'] for k, v in d['kernels'].items() if k.startswith('gemm')})"; }
for rep in 1 2; do
  echo -n "base:   "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  for v in noread nodma; do echo -n "$v: "; MEMVUL_HIP_LIB=$P/libpp_$v.so timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one; done
done
