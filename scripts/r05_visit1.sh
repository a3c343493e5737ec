#!/bin/bash
# Round 5, visit 1: (1) the hi16 + lo8 residual stream and the saturation counter: targeted parity tests first (stop if they fail),
# (2) the full GPU suite, (3) A/B of the round-4 kernels (tools/probe_r5/lib_r4_base.so) against the tree's library, precise mode,
# (4) the 24-draw error distribution, (5) the precision envelope.
set -u
O=gpurun_out/r05_v1
mkdir -p $O
export TMPDIR=/tmp
python -m memvul_amd.build > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "precise or saturation or trained_like_rows or full_batch" > $O/pytest_targeted.log 2>&1
rc=$?; tail -15 $O/pytest_targeted.log; grep -h "AssertionError: {" $O/pytest_targeted.log | cut -c1-300
if grep -q "hipError\|Segmentation\|core dumped\|nan" $O/pytest_targeted.log; then echo "TARGETED TESTS BROKEN"; tail -60 $O/pytest_targeted.log; exit 0; fi
( timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -8 $O/pytest_gpu.log
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(round(d['value']), 'one-in-flight', round(d['value_one_batch_in_flight']), ' '.join('%s=%.1f' % (n, k[n]['avg_us']) for n in ('embed_ln','gemm_qkv','attention','gemm_attn_out','gemm_ffn1_gelu','gemm_ffn2','cls_tail') if n in k))"; }
Q="--compute precise --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --steps 20 --warmup 5"
for rep in 1 2; do
  echo -n "r4 kernels (base)      : "; MEMVUL_HIP_LIB=$PWD/tools/probe_r5/lib_r4_base.so timeout 300 python bench.py $Q 2>$O/err_base.log | tail -1 | one
  echo -n "r5 hi16+lo8 stream+cnt : "; timeout 300 python bench.py $Q 2>$O/err_new.log | tail -1 | one
done > $O/ab_stream_lo8.txt 2>&1; cat $O/ab_stream_lo8.txt
timeout 900 python scripts/r05_error_distribution.py --json $O/error_distribution.json > $O/error_distribution.txt 2>&1; tail -6 $O/error_distribution.txt
MEMVUL_HIP_LIB=$PWD/tools/probe_r5/lib_r4_base.so timeout 900 python scripts/r05_error_distribution.py --f16-seeds 0 --json $O/error_distribution_r4_kernels.json > $O/error_distribution_r4_kernels.txt 2>&1; tail -3 $O/error_distribution_r4_kernels.txt
timeout 600 python scripts/r05_precision_envelope.py $O/precision_envelope.json > $O/precision_envelope.txt 2>&1; cat $O/precision_envelope.txt
