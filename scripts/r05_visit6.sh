#!/bin/bash
# Round 5, visit 6: second pass of the margin table — the V block alone (and K alone) against Q alone / Q + V, both streams, all 24 draws.
set -u
O=gpurun_out/r05_v6
mkdir -p $O
export TMPDIR=/tmp
python -m memvul_amd.build > $O/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(round(d['value']), 'one-in-flight', round(d['value_one_batch_in_flight']), ' '.join('%s=%.1f' % (n, k[n]['avg_us']) for n in ('gemm_qkv','gemm_attn_out','gemm_ffn1_gelu','gemm_ffn2','gemm_kv_last') if n in k))"; }
Q="--compute precise --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --steps 20 --warmup 5"
for rep in 1 2; do
for CFG in q/0 v/0 v/1 qv/1; do AS=${CFG%/*}; LO8=${CFG#*/}
  echo -n "aside=$AS lo8=$LO8 : "; MEMVUL_QKV_ASIDE=$AS MEMVUL_STREAM_LO8=$LO8 timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
done; done > $O/margin_rates2.txt 2>&1; cat $O/margin_rates2.txt
R05_MARGIN_CONFIGS="q/0,v/0,k/0,v/1,qv/1,kv/1" timeout 1500 python scripts/r05_margin_table.py 24 $O/margin_errors2.json > $O/margin_errors2.txt 2>&1; tail -7 $O/margin_errors2.txt
