#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r4m
mkdir -p $OUT
timeout -s KILL 120 python tools/trace_c3.py 2>&1 | grep -E "phase|total|streamed" | head -40
timeout -s KILL 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-call-n 30 2>$OUT/bench.err > $OUT/bench.json
python - $OUT/bench.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = d["one_call"]
print("value %.1f" % d["value"], "repeats", [round(x) for x in d["repeats"]], "one_call %.2f ms (min %.2f) bulk %.2f front %.2f plan %.2f" % (oc["ms_per_call"], oc["ms_per_call_min_max"][0], oc["ms_bulk_kernel"], oc["ms_front_kernel"], oc["ms_host_planning"]))
PY
