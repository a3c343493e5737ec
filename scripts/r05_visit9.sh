#!/bin/bash
# Round 5, visit 9: V and P as two fp16 planes through the attention of short passes (padded length <= 128, precise mode).
set -u
O=gpurun_out/r05_v9
mkdir -p $O
export TMPDIR=/tmp
python -m memvul_amd.build > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; exit 1; }
timeout 600 python scripts/r05_length_envelope.py $O/length_envelope.json 2>&1 | grep -v amdgpu.ids | tee $O/length_envelope.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "short_sequences or trained_like_rows or precise or lo8" > $O/pytest_targeted.log 2>&1
tail -4 $O/pytest_targeted.log; grep -h "AssertionError" $O/pytest_targeted.log | cut -c1-400
grep -h "short_sequences\|full_batch_trained_like" gpurun_out/diag.jsonl | tail -3 | cut -c1-400
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(round(d['value']), 'one-in-flight', round(d['value_one_batch_in_flight']), ' '.join('%s=%.1f' % (n, k[n]['avg_us']) for n in ('gemm_qkv','attention','gemm_attn_out','gemm_ffn1_gelu','gemm_ffn2','gemm_kv_last') if n in k))"; }
Q="--compute precise --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --steps 20 --warmup 5"
for rep in 1 2; do
  echo -n "visit-4 binary S256 : "; MEMVUL_HIP_LIB=$PWD/tools/probe_r5/lib_r5_visit4.so timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  echo -n "this tree      S256 : "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
done > $O/ab.txt 2>&1
for rep in 1; do
  echo -n "visit-4 binary S128 : "; MEMVUL_HIP_LIB=$PWD/tools/probe_r5/lib_r5_visit4.so timeout 300 python bench.py $Q --seq-len 128 2>/dev/null | tail -1 | one
  echo -n "this tree      S128 : "; timeout 300 python bench.py $Q --seq-len 128 2>/dev/null | tail -1 | one
  echo -n "visit-4 binary S64 B512 : "; MEMVUL_HIP_LIB=$PWD/tools/probe_r5/lib_r5_visit4.so timeout 300 python bench.py $Q --seq-len 64 --batch 512 2>/dev/null | tail -1 | one
  echo -n "this tree      S64 B512 : "; timeout 300 python bench.py $Q --seq-len 64 --batch 512 2>/dev/null | tail -1 | one
done >> $O/ab.txt 2>&1; cat $O/ab.txt
