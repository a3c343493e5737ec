#!/bin/bash
# Round 5, visit 19 (the shipped library, no build): numbers for the documentation — the default invocation of bench.py, the ragged corpus (padded / length-bucketed)
# and the sequence-length axis of the precision envelope under the [CLS]-row default.
set -u
O=gpurun_out/r05_v19
mkdir -p $O
export TMPDIR=/tmp
( timeout 400 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "rc=$?" >> $O/bench_line.err ); cut -c1-400 $O/bench_line.json
Q="--matcher-anchors 0 --cpu-sample 0 --sustain-s 0 --no-second --steps 20 --warmup 5"
{
echo -n "precise ragged (lengths uniform in [16, 256]) : "; timeout 300 python bench.py --ragged --compute precise $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('ragged'))"
echo -n "precise ragged, MEMVUL_CLS_ASIDE=0            : "; MEMVUL_CLS_ASIDE=0 timeout 300 python bench.py --ragged --compute precise $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('ragged'))"
} > $O/other_configs.txt 2>&1; cat $O/other_configs.txt
timeout 300 python scripts/r05_length_envelope.py $O/length_envelope.json > $O/length_envelope.txt 2>&1; grep -v amdgpu.ids $O/length_envelope.txt
