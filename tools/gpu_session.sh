#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session K -- the seed re-optimisation round (MI_DMRECON_SEED_REOPT=1).
export TMPDIR=/tmp
O=gpurun_out/r5k
mkdir -p $O
timeout -s KILL 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "seed_reopt or batch or more_seeds" > $O/pytest.log 2>&1; grep -E "seed re-opt|passed|failed|Error|assert" $O/pytest.log | head -20
AB="--steps 20 --warmup 3 --repeats 2 --distinct-scenes 0 --one-call-n 10"
MI_DMRECON_SEED_REOPT=1 timeout -s KILL 400 python bench.py $AB > $O/bench_reopt.json 2> $O/bench_reopt.err
python - <<PY
import json
j = json.loads(open("$O/bench_reopt.json").read().strip().splitlines()[-1])
p = j["parity"]
print("reopt: value %.1f one_call %.2f ms | parity" % (j["value"], j["one_call"]["ms_per_call"]), {k: p[k] for k in ("min_fill_iou", "max_rel_depth_median", "max_rel_depth_p99", "max_conf_abs_p99", "within_bounds")})
print(p["fill_iou_per_view"]); print("mean fill", j["config"]["mean_fill"], "n_patch", j["roofline"]["n_patch"], "n_eval", j["roofline"]["n_eval"])
PY
