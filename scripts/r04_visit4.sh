#!/bin/bash
mkdir -p gpurun_out/v4
O=gpurun_out/v4
export TMPDIR=/tmp
python -m memvul_amd.build > /dev/null || exit 1
timeout 600 python scripts/r04_qkv_terms_errors.py > $O/qkv_terms_errors.txt 2>&1; cat $O/qkv_terms_errors.txt | tail -4
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v['avg_us'],1) for k, v in d['kernels'].items()})"; }
Q="--compute precise --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --streams 1 --steps 10 --warmup 3"
for rep in 1 2 3; do
  echo -n "precise (FFN-2 writes hi8 only): "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  echo -n "precise, QKV both terms (FFN-2 writes both planes): "; MEMVUL_QKV_X8_TERMS=2 timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
done > $O/precise_ab.txt 2>&1; cat $O/precise_ab.txt
( cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "mall|hbm|dram|EA0_RD|EA_RD|MC_RD|TCC_REQ|TCC_HIT|TCC_MISS|TCP_TCC" | head -60 ) > $O/counters_list.txt 2>&1; wc -l $O/counters_list.txt; head -40 $O/counters_list.txt
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
