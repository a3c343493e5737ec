#!/bin/bash
# Round 5, visit 16: the [CLS]-row form as the library's default — the whole GPU suite on it, and the row term in one launch (cls_corr_kernel)
# against the two-launch form (MEMVUL_CLS_FIX=ring).
set -u
O=gpurun_out/r05_v16
mkdir -p $O
export TMPDIR=/tmp
python -c "import memvul_amd.build as b; print('stale:', b.is_stale())" > $O/build.log 2>&1; cat $O/build.log
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(round(d['value']), 'one-in-flight', round(d['value_one_batch_in_flight']), ' '.join('%s=%.1f' % (n, k[n]['avg_us']) for n in ('gemm_qkv','attention','gemm_attn_out','gemm_ffn1_gelu','gemm_ffn2','gemm_kv_last','cls_tail','other') if n in k))"; }
Q="--compute precise --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --steps 20 --warmup 5"
{
for rep in 1 2; do
  echo -n "default ([CLS]-row form, row term in one launch): "; timeout 200 python bench.py $Q 2>/dev/null | tail -1 | one
  echo -n "row term in two launches (MEMVUL_CLS_FIX=ring)  : "; MEMVUL_CLS_FIX=ring timeout 200 python bench.py $Q 2>/dev/null | tail -1 | one
done
echo -n "both terms (MEMVUL_CLS_ASIDE=0)                 : "; MEMVUL_CLS_ASIDE=0 timeout 200 python bench.py $Q 2>/dev/null | tail -1 | one
} > $O/ab_rates.txt 2>&1; cat $O/ab_rates.txt
R05_MARGIN_CONFIGS="q/0/1" timeout 300 python scripts/r05_margin_table.py 8 > $O/errors8.txt 2>&1; tail -1 $O/errors8.txt
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log
