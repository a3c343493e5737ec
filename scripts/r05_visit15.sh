#!/bin/bash
# Round 5, visit 15: the [CLS]-row A-side form decided per sequence (GemmArgs::tile_both), no row term for the QKV projection when its Q block carries the
# A-side term: parity tests, rate against the both-terms form, trained-like logit error over 24 draws.
set -u
O=gpurun_out/r05_v15
mkdir -p $O
export TMPDIR=/tmp
python -c "import memvul_amd.build as b; print('stale:', b.is_stale())" > $O/build.log 2>&1; cat $O/build.log
timeout 500 python -m pytest tests/test_gpu_parity.py -q -k "cls_row_aside" > $O/pytest_targeted.log 2>&1; tail -15 $O/pytest_targeted.log
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(round(d['value']), 'one-in-flight', round(d['value_one_batch_in_flight']), ' '.join('%s=%.1f' % (n, k[n]['avg_us']) for n in ('gemm_qkv','attention','gemm_attn_out','gemm_ffn1_gelu','gemm_ffn2','gemm_kv_last','cls_tail','other') if n in k))"; }
Q="--compute precise --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --steps 20 --warmup 5"
{
for rep in 1 2; do
  echo -n "both terms (MEMVUL_CLS_ASIDE=0)     : "; MEMVUL_CLS_ASIDE=0 timeout 200 python bench.py $Q 2>/dev/null | tail -1 | one
  echo -n "[CLS]-row form, aside q             : "; MEMVUL_CLS_ASIDE=1 timeout 200 python bench.py $Q 2>/dev/null | tail -1 | one
done
echo -n "[CLS]-row form, aside none          : "; MEMVUL_CLS_ASIDE=1 MEMVUL_QKV_ASIDE=none timeout 200 python bench.py $Q 2>/dev/null | tail -1 | one
echo -n "[CLS]-row form, aside q, S 512 B 128: "; MEMVUL_CLS_ASIDE=1 timeout 200 python bench.py $Q --seq-len 512 --batch 128 2>/dev/null | tail -1 | one
echo -n "both terms, S 512 B 128             : "; MEMVUL_CLS_ASIDE=0 timeout 200 python bench.py $Q --seq-len 512 --batch 128 2>/dev/null | tail -1 | one
} > $O/ab_rates.txt 2>&1; cat $O/ab_rates.txt
R05_MARGIN_CONFIGS="q/0/1" timeout 600 python scripts/r05_margin_table.py 24 $O/errors.json > $O/errors.txt 2>&1; tail -2 $O/errors.txt
