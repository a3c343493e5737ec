#!/bin/bash
mkdir -p gpurun_out/v6
O=gpurun_out/v6
python -m memvul_amd.build > /dev/null || exit 1
timeout 300 tools/match_probe > $O/match_probe.txt 2>&1; grep "MATCH" $O/match_probe.txt | head -24
timeout 600 python -m pytest tests -m gpu -q -k "match or topk or plumbing or reference" > $O/pytest_match.txt 2>&1; tail -3 $O/pytest_match.txt
for rep in 1 2; do timeout 300 python bench.py --compute precise --cpu-sample 0 --sustain-s 0 --no-second --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['matcher']['avg_us'], d['matcher']['valu_frac'], d['kernels']['match'])"; done > $O/bench_matcher.txt 2>&1; cat $O/bench_matcher.txt
