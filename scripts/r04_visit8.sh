#!/bin/bash
mkdir -p gpurun_out/v8
O=gpurun_out/v8
python -m memvul_amd.build > /dev/null || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
m() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['matcher']['avg_us'], d['matcher']['valu_frac'], d['matcher']['launches'])"; }
Q="--compute f16 --cpu-sample 0 --sustain-s 0 --no-second --steps 4 --warmup 2 --no-profile"
for rep in 1 2 3; do
  echo -n "base (GC 256, MI 32): "; timeout 200 python bench.py $Q 2>/dev/null | tail -1 | m
  echo -n "MI 64               : "; MEMVUL_MATCH_MI=64 timeout 200 python bench.py $Q 2>/dev/null | tail -1 | m
  echo -n "GC 128              : "; MEMVUL_MATCH_GC=128 timeout 200 python bench.py $Q 2>/dev/null | tail -1 | m
done > $O/matcher_ab.txt 2>&1; cat $O/matcher_ab.txt
MEMVUL_MATCH_GC=128 timeout 600 python -m pytest tests -m gpu -q -k "match or topk" > $O/pytest_gc128.txt 2>&1; tail -2 $O/pytest_gc128.txt
MEMVUL_MATCH_MI=64 timeout 600 python -m pytest tests -m gpu -q -k "match or topk" > $O/pytest_mi64.txt 2>&1; tail -2 $O/pytest_mi64.txt
