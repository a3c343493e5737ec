#!/bin/bash
# timing probe: the precise mode's correction sweep issued in the fp6 MFMA format on the same (fp8) bytes — numerically meaningless, same LDS / DMA traffic
Q="--compute precise --matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-precise --streams 1"
for rep in 1 2; do
  echo "== shipped library (fp8 correction sweep)"; timeout 300 python bench.py $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
  echo "== probe: fp6 format"; MEMVUL_HIP_LIB=$PWD/tools/probe_fp6/libmemvul_fp6probe.so timeout 300 python bench.py $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['avg_us'] for k, v in d['kernels'].items()})"
done
