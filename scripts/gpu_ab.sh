#!/bin/bash
# Same-box A/B of engine switches: each argument is "name VAR=VAL [VAR=VAL...]"; every variant runs bench.py
# (no CPU sample) ROUNDS times, interleaved, and one line per run is printed / appended to gpurun_out/ab.jsonl.
set -u
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
ROUNDS=${ROUNDS:-2}
BENCH_ARGS=${BENCH_ARGS:---steps 30 --warmup 5}
for r in $(seq 1 $ROUNDS); do
  for spec in "$@"; do
    read -r name envs <<< "$spec"
    env $envs timeout 300 python bench.py --cpu-sample 0 $BENCH_ARGS 2> $O/ab_$name.err | tail -1 > $O/ab_$name.json
    python - "$O/ab_$name.json" "$name" "$r" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    k = d.get("kernels", {})
    print(f"[{sys.argv[3]}] {sys.argv[2]:<14} IR/s {d['value']:>9} | " + " ".join(f"{n}={v['avg_us']}" for n, v in k.items()), flush=True)
    open("gpurun_out/ab.jsonl", "a").write(json.dumps({"variant": sys.argv[2], "round": int(sys.argv[3]), "value": d["value"], "kernels": {n: v["avg_us"] for n, v in k.items()}}) + "\n")
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
  done
done
