#!/bin/bash
# same-box A/B of the precise mode against the timing probes of build.py (numerically meaningless libraries; MEMVUL_HIP_LIB selects them)
Q="--compute precise --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-precise --streams 1 --steps 10 --warmup 3"
P=$PWD/tools/probe_x8half
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v['avg_us'],1) for k, v in d['kernels'].items()})"; }
for rep in 1 2; do
  echo -n "base   : "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  for m in half half6 nocorr; do
    echo -n "$m : "; MEMVUL_HIP_LIB=$P/lib_$m.so timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  done
done
echo -n "f16    : "; timeout 300 python bench.py --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-precise --streams 1 --steps 10 --warmup 3 2>/dev/null | tail -1 | one
