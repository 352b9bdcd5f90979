#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, the closing pass on the final commit -- what the driver runs at the end of a
# round: the whole GPU suite, smoke(), the bench at its defaults' driver form.
export TMPDIR=/tmp
O=gpurun_out/r5end
mkdir -p $O
( timeout -s KILL 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log
( timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log ); tail -2 $O/smoke.log
timeout -s KILL 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-420 $O/bench_driver.json
python - $O/bench_driver.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print("roofline:", {k: r[k] for k in ("bound", "achieved", "peak", "frac", "traffic", "bulk_kernel_frac", "traffic_over_algorithmic")})
print("traffic_source:", r["traffic_source"]["profiled_plan_is_this_runs"], "| cpu_baseline:", {k: j["cpu_baseline"][k] for k in ("value", "cores", "kind")}, "| parity within bounds:", j["parity"]["within_bounds"])
PY
