#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Session r4a: the GPU suite after the per-view hand-over / give-up changes, the
# instruction-cost micro-benchmark, a lone-call trace and a sweep of the per-view hand-over threshold.
export TMPDIR=/tmp
OUT=gpurun_out/r4a
mkdir -p $OUT
timeout -s KILL 60 build/valu_rate2 > $OUT/valu_rate.txt 2>&1; head -20 $OUT/valu_rate.txt
timeout -s KILL 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -40 > $OUT/pytest.txt; tail -15 $OUT/pytest.txt
timeout -s KILL 120 python tools/trace_c3.py > $OUT/round_trace_c3.txt 2>&1; grep -E "phase|total" $OUT/round_trace_c3.txt
for H in 160 320 640 1280; do
  MI_DMRECON_VIEW_HANDOVER=$H timeout -s KILL 200 python bench.py --gpus 1 --steps 20 --warmup 5 --repeats 3 --one-call-n 20 --no-cpu-baseline 2>$OUT/bench_h$H.err > $OUT/bench_h$H.json
  python - $OUT/bench_h$H.json $H <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    oc = d.get("one_call", {})
    print("handover", sys.argv[2], "value %.1f" % d["value"], "repeats", [round(x) for x in d["repeats"]], "one_call ms %.2f (bulk %.2f front %.2f tail %.2f rounds %s ffr %s)" % (
        oc.get("ms_per_call", 0), oc.get("ms_bulk_kernel", 0), oc.get("ms_front_kernel", 0), oc.get("ms_tail_kernel", 0), oc.get("rounds"), oc.get("front_first_round")),
        "bulk frac %.3f valu_issue %.3f" % (d["roofline"]["bulk_kernel_frac"], d["roofline"]["secondary_roofs"]["valu_issue"]["frac"]), d["config"]["library_batch_log"])
except Exception as e:
    print("handover", sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
