#!/bin/bash
# TIMING probe (numerically meaningless): gemm_pp with every v_mfma_f32_32x32x16_f16 of the fp16 sweep replaced by two
# v_mfma_f32_16x16x32_f16 on the same registers — does the power-bound main loop get faster with the more energy-efficient shape?
Q="--matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-precise --streams 1"
P=$PWD/tools/probe_m16
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['avg_us'] for k, v in d['kernels'].items() if k.startswith('gemm')})"; }
for rep in 1 2 3; do
  echo -n "base (32x32x16): "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  echo -n "probe (16x16x32): "; MEMVUL_HIP_LIB=$P/libpp_m16.so timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
done
