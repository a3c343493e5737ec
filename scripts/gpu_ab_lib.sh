#!/bin/bash
# Same-box A/B of two builds of the library: bench.py (value + kernel classes only) alternately on $1 (a second .so, e.g. one built from another commit's sources
# into memvul_amd/lib/) and on the tree's own library.   usage: scripts/gpu_ab_lib.sh memvul_amd/lib/libmemvul_hip_head.so [rounds]
OTHER=$(readlink -f "$1"); R=${2:-2}
for r in $(seq $R); do
  for which in other tree; do
    if [ $which = other ]; then export MEMVUL_HIP_LIB=$OTHER; else unset MEMVUL_HIP_LIB; fi
    python bench.py --cpu-sample 0 --no-second --matcher-anchors 0 --sustain-s 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which', d['value'], d['value_one_batch_in_flight'], {k: v['avg_us'] for k, v in d['kernels'].items() if k.startswith('gemm') or k in ('embed_ln', 'attention')})"
  done
done
unset MEMVUL_HIP_LIB
