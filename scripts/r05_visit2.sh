#!/bin/bash
# Round 5, visit 2: saturation counter as a per-lane maximum (no scalar state) + the lo8 stream as an opt-in kernel instantiation.
# Targeted tests, then A/B: round-4 kernels / this tree / this tree with MEMVUL_STREAM_LO8=1, precise mode, same box, two alternations.
set -u
O=gpurun_out/r05_v2
mkdir -p $O
export TMPDIR=/tmp
python -m memvul_amd.build > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "precise or saturation or trained_like_rows" > $O/pytest_targeted.log 2>&1
tail -5 $O/pytest_targeted.log; grep -h "AssertionError: {" $O/pytest_targeted.log | cut -c1-300
MEMVUL_STREAM_LO8=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -q -k "precise or saturation or trained_like_rows or layer0" > $O/pytest_targeted_lo8.log 2>&1
tail -5 $O/pytest_targeted_lo8.log; grep -h "AssertionError: {" $O/pytest_targeted_lo8.log | cut -c1-300
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(round(d['value']), 'one-in-flight', round(d['value_one_batch_in_flight']), ' '.join('%s=%.1f' % (n, k[n]['avg_us']) for n in ('embed_ln','gemm_qkv','attention','gemm_attn_out','gemm_ffn1_gelu','gemm_ffn2','cls_tail') if n in k))"; }
Q="--compute precise --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --steps 20 --warmup 5"
for rep in 1 2; do
  echo -n "r4 kernels (base)        : "; MEMVUL_HIP_LIB=$PWD/tools/probe_r5/lib_r4_base.so timeout 300 python bench.py $Q 2>$O/err_base.log | tail -1 | one
  echo -n "r5 tree (lo16 + counter) : "; timeout 300 python bench.py $Q 2>$O/err_new.log | tail -1 | one
  echo -n "r5 tree, STREAM_LO8=1    : "; MEMVUL_STREAM_LO8=1 timeout 300 python bench.py $Q 2>$O/err_lo8.log | tail -1 | one
done > $O/ab_counter_and_stream.txt 2>&1; cat $O/ab_counter_and_stream.txt
