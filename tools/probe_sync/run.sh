#!/bin/bash
# timing ablations of the persistent GEMM's per-interval synchronisation (WRONG results by construction):
#   nobar  = no s_barrier at the end of an interval (waves free-run inside a tile; the counted vmcnt wait stays)
#   nowait = the barrier stays, the counted `s_waitcnt vmcnt` in front of it is gone
Q="--matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-precise --streams 1"
P=$PWD/tools/probe_sync
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['avg_us'] for k, v in d['kernels'].items() if k.startswith('gemm')})"; }
for rep in 1 2; do
  echo -n "base:   "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  for v in nobar nowait; do echo -n "$v: "; MEMVUL_HIP_LIB=$P/libpp_$v.so timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one; done
done
