#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session D -- the second wind of the front (a one-workgroup-per-view launch
# stopped when few views are left, the rest continued with teams): its test, the parity file, bench lines with and without.
export TMPDIR=/tmp
O=gpurun_out/r5d
mkdir -p $O
line() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
oc = j.get("one_call") or {}
d = j["config"].get("distinct_scenes_variant") or {}
print("%s: value %.1f %s | bulk frac %.3f | one_call %.2f ms (bulk %.2f front %.2f) | distinct %.1f (bulk %.2f ms/step, front %.2f)" % (
      sys.argv[1], j["value"], [round(v) for v in j["repeats"]], j["roofline"]["bulk_kernel_frac"], oc.get("ms_per_call", 0),
      oc.get("ms_bulk_kernel", 0), oc.get("ms_front_kernel", 0), d.get("value", 0), d.get("ms_bulk_kernel_per_step", 0), d.get("ms_front_kernel_per_step", 0)))
PY
}
timeout -s KILL 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
AB="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --distinct-scenes 0 --no-one-call"
MI_DMRECON_TRACE=1 MI_BENCH_REGION_LOG=1 timeout -s KILL 300 python bench.py --steps 20 --warmup 2 --repeats 1 --no-cpu-baseline --distinct-scenes 0 --no-one-call 2> $O/trace.err > /dev/null
grep -E "front:|second launch|phase C|download" $O/trace.err | tail -8
MI_BENCH_REGION_LOG=1 timeout -s KILL 300 python bench.py $AB > $O/bench_wind.json 2> $O/bench_wind.err
line $O/bench_wind.json; grep "^region" $O/bench_wind.err | tail -2
MI_DMRECON_SECOND_WIND=0 MI_BENCH_REGION_LOG=1 timeout -s KILL 300 python bench.py $AB > $O/bench_nowind.json 2> $O/bench_nowind.err
line $O/bench_nowind.json; grep "^region" $O/bench_nowind.err | tail -1
MI_BENCH_REGION_LOG=1 timeout -s KILL 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
line $O/bench_driver.json; grep "^region" $O/bench_driver.err | sed -n '5p;8p'
timeout -s KILL 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py > $O/pytest_rest.log 2>&1; tail -4 $O/pytest_rest.log
