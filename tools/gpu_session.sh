#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session J -- how many single-attempt follow-up launches a large round should have.
export TMPDIR=/tmp
O=gpurun_out/r5j
mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%s: value %.1f %s | bulk frac %.3f" % (sys.argv[1], j["value"], [round(v) for v in j["repeats"]], j["roofline"]["bulk_kernel_frac"]))
PY
}
timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch" 2>&1 | tail -2
AB="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --distinct-scenes 0 --no-one-call"
for E in 4 1 2 0 4; do
  MI_DMRECON_SINGLE_FOLLOW=$E MI_BENCH_REGION_LOG=1 timeout -s KILL 300 python bench.py $AB > $O/bench_sf$E.json 2> $O/bench_sf$E.err
  line $O/bench_sf$E.json; grep "^region" $O/bench_sf$E.err | tail -1
done
