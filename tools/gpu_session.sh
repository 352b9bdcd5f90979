#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Session r4l: maps of finished views streamed back during the front kernel.
export TMPDIR=/tmp
OUT=gpurun_out/r4l
mkdir -p $OUT
timeout -s KILL 900 python -m pytest tests -q -m gpu 2>&1 | tail -150 > $OUT/pytest.txt; grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.txt
for G in x 1; do
if [ $G = 1 ]; then export MI_DMRECON_GVS_DEVICE=1; fi
timeout -s KILL 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --one-call-n 30 2>$OUT/bench$G.err > $OUT/bench$G.json
python - $OUT/bench$G.json $G <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = d["one_call"]
print("gvs_device", sys.argv[2], "value %.1f" % d["value"], "repeats", [round(x) for x in d["repeats"]], "one_call %.2f ms (min %.2f) bulk %.2f front %.2f plan %.2f" % (oc["ms_per_call"], oc["ms_per_call_min_max"][0], oc["ms_bulk_kernel"], oc["ms_front_kernel"], oc["ms_host_planning"]))
PY
done
unset MI_DMRECON_GVS_DEVICE
timeout -s KILL 120 python tools/trace_c3.py 2>&1 | grep -E "phase|total"
