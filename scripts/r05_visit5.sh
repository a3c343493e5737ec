#!/bin/bash
# Round 5, visit 5: the margin table — error over 16 draws and rate of every (QKV A-side blocks) x (stream low part) combination of the precise mode.
set -u
O=gpurun_out/r05_v5
mkdir -p $O
export TMPDIR=/tmp
python -m memvul_amd.build > $O/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(round(d['value']), 'one-in-flight', round(d['value_one_batch_in_flight']), ' '.join('%s=%.1f' % (n, k[n]['avg_us']) for n in ('gemm_qkv','gemm_attn_out','gemm_ffn1_gelu','gemm_ffn2','gemm_kv_last') if n in k))"; }
Q="--compute precise --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --steps 20 --warmup 5"
for rep in 1 2; do
for LO8 in 0 1; do for AS in none q qv qkv; do
  echo -n "aside=$AS lo8=$LO8 : "; MEMVUL_QKV_ASIDE=$AS MEMVUL_STREAM_LO8=$LO8 timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
done; done; done > $O/margin_rates.txt 2>&1; cat $O/margin_rates.txt
timeout 1500 python scripts/r05_margin_table.py 16 $O/margin_errors.json > $O/margin_errors.txt 2>&1; tail -9 $O/margin_errors.txt
