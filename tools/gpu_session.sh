#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session M -- the round's bench lines on the final build (the PMC passes of
# session E stay: profiles/r5_pmc.md, r5_traffic.json).
export TMPDIR=/tmp
O=gpurun_out/r5
mkdir -p $O
MI_BENCH_REGION_LOG=1 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python bench.py --streams 1 --steps-per-call 1 --steps 20 --repeats 3 --no-cpu-baseline --no-one-call --distinct-scenes 0 > $O/bench_1thread.json 2> $O/bench_1thread.err
timeout -s KILL 120 python tools/trace_c3.py > $O/round_trace_c3.txt 2>&1
timeout -s KILL 300 python tools/lone_calls.py C3 12 > $O/lone_calls.json 2> $O/lone_calls.err
timeout -s KILL 300 python bench.py --config C2 --steps 20 --warmup 3 > $O/bench_c2.json 2> $O/bench_c2.err
timeout -s KILL 600 python bench.py --config C5 --steps 4 --warmup 1 --repeats 3 --streams 2 --steps-per-call 1 > $O/bench_c5.json 2> $O/bench_c5.err
MI_DMRECON_TRACE=1 timeout -s KILL 300 python tools/app_c3_timing.py 2>&1 | grep -v "^\[mi_dmrecon\]" > $O/app_c3_timing.txt
for f in driver 1thread c2 c5; do python - <<PY
import json
j = json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
print("$f", round(j["value"], 1), [round(v) for v in j["repeats"]], "one_call", (j.get("one_call") or {}).get("ms_per_call"), "traffic src", (j["roofline"].get("traffic_source") or {}).get("profiled_plan_is_this_runs"))
PY
done
grep -v '(view)' $O/app_c3_timing.txt | tail -6
