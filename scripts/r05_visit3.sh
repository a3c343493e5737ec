#!/bin/bash
# Round 5, visit 3: plane conversions without clamps on the common path (fp16 roundings taken from the packed output words, v_fma_mix_f32 residuals):
# bit-equality against the clamped build, targeted tests, A/B.
set -u
O=gpurun_out/r05_v3
mkdir -p $O
export TMPDIR=/tmp
python -m memvul_amd.build > $O/build.log 2>&1 || { echo BUILD FAILED; tail -20 $O/build.log; exit 1; }
timeout 600 python scripts/r05_bits_equal.py tools/probe_r5/lib_r5_clamped_planes.so memvul_amd/lib/libmemvul_hip.so 2>&1 | tee $O/bits_equal.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "precise or saturation or trained_like_rows or lo8" > $O/pytest_targeted.log 2>&1
tail -4 $O/pytest_targeted.log; grep -h "AssertionError: {" $O/pytest_targeted.log | cut -c1-300
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(round(d['value']), 'one-in-flight', round(d['value_one_batch_in_flight']), ' '.join('%s=%.1f' % (n, k[n]['avg_us']) for n in ('embed_ln','gemm_qkv','attention','gemm_attn_out','gemm_ffn1_gelu','gemm_ffn2','cls_tail') if n in k))"; }
Q="--compute precise --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --steps 20 --warmup 5"
for rep in 1 2; do
  echo -n "clamped planes (visit 2's tree)   : "; MEMVUL_HIP_LIB=$PWD/tools/probe_r5/lib_r5_clamped_planes.so timeout 300 python bench.py $Q 2>$O/err_a.log | tail -1 | one
  echo -n "in-range planes (this tree)       : "; timeout 300 python bench.py $Q 2>$O/err_b.log | tail -1 | one
  echo -n "in-range planes, STREAM_LO8=1     : "; MEMVUL_STREAM_LO8=1 timeout 300 python bench.py $Q 2>$O/err_c.log | tail -1 | one
done > $O/ab_planes.txt 2>&1; cat $O/ab_planes.txt
for rep in 1; do
  echo -n "S512 clamped planes   : "; MEMVUL_HIP_LIB=$PWD/tools/probe_r5/lib_r5_clamped_planes.so timeout 300 python bench.py $Q --seq-len 512 --batch 128 2>/dev/null | tail -1 | one
  echo -n "S512 in-range planes  : "; timeout 300 python bench.py $Q --seq-len 512 --batch 128 2>/dev/null | tail -1 | one
done > $O/ab_planes_s512.txt 2>&1; cat $O/ab_planes_s512.txt
