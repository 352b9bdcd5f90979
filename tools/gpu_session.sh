#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session G -- front order "interleaved" and the threshold of the speculative rounds.
export TMPDIR=/tmp
O=gpurun_out/r5g
mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = j.get("one_call") or {}
print("%s: value %.1f %s | bulk frac %.3f | one_call %.2f ms (bulk %.2f front %.2f)" % (sys.argv[1], j["value"], [round(v) for v in j["repeats"]],
      j["roofline"]["bulk_kernel_frac"], oc.get("ms_per_call", 0), oc.get("ms_bulk_kernel", 0), oc.get("ms_front_kernel", 0)))
PY
}
AB="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --distinct-scenes 0 --one-call-n 20"
for E in "MI_X=0" "MI_DMRECON_FRONT_ORDER=3" "MI_DMRECON_SPEC_ROUNDS=150000" "MI_DMRECON_SPEC_ROUNDS=1000000" "MI_X=1"; do
  env MI_BENCH_REGION_LOG=1 $E timeout -s KILL 300 python bench.py $AB > $O/bench_$E.json 2> $O/bench_$E.err
  line $O/bench_$E.json; grep "^region" $O/bench_$E.err | tail -1
done
