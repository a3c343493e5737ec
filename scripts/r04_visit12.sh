#!/bin/bash
mkdir -p gpurun_out/v12
O=gpurun_out/v12
python -m memvul_amd.build > /dev/null || exit 1
P=$PWD/tools/probe_attn
one() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['kernels']['attention']['avg_us'],1))"; }
for rep in 1 2 3; do
for MODE in f16 precise; do
  Q="--compute $MODE --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --streams 1 --steps 10 --warmup 3"
  echo -n "$MODE S256 O through the ring slot (rounds 1-3): "; MEMVUL_HIP_LIB=$P/lib_ring_o_image.so timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  echo -n "$MODE S256 own O image, DMA first         : "; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | one
  echo -n "$MODE S512 O through the ring slot (rounds 1-3): "; MEMVUL_HIP_LIB=$P/lib_ring_o_image.so timeout 300 python bench.py $Q --seq-len 512 --batch 128 2>/dev/null | tail -1 | one
  echo -n "$MODE S512 own O image, DMA first         : "; timeout 300 python bench.py $Q --seq-len 512 --batch 128 2>/dev/null | tail -1 | one
done; done > $O/attn_o_image_ab.txt 2>&1; cat $O/attn_o_image_ab.txt
