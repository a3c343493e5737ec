#!/bin/bash
# A/B: de-phasing the two wave groups of attention_v2 (S = 256: waves 0-3 and 4-7 share the four SIMDs pairwise) by WHERE the
# first group does its per-unit bookkeeping (previous unit's O flush + next unit's LDS-DMA issue):
#   base = both groups before QK^T;  a1 = first group after its QK^T;  a2 = first group after its softmax (before PV)
Q="--matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-precise --streams 1"
P=$PWD/tools/probe_attn
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['avg_us'] for k, v in d['kernels'].items() if k in ('attention','gemm_qkv','gemm_attn_out')})"; }
for v in a1 a2; do echo "tests on $v:"; MEMVUL_HIP_LIB=$P/libattn_$v.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "layer0 or attention" 2>&1 | tail -1; done
for rep in 1 2 3; do
  echo -n "base: "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  for v in a1 a2; do echo -n "$v:   "; MEMVUL_HIP_LIB=$P/libattn_$v.so timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one; done
done
