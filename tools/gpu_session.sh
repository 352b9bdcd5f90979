#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session A -- the GPU suite, the driver's command with its region log, and the
# two new scheduling forms against their switches in the same lease (front order, one attempt per follow-up launch).
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
nproc > $O/host.txt; cat /sys/fs/cgroup/cpu.max >> $O/host.txt 2>/dev/null
timeout -s KILL 1100 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1
tail -25 $O/pytest.log
MI_BENCH_REGION_LOG=1 timeout -s KILL 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
tail -c 1500 $O/bench_driver.json; grep "^region" $O/bench_driver.err | tail -6
AB="--steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-one-call --distinct-scenes 0"
MI_BENCH_REGION_LOG=1 MI_DMRECON_FRONT_ORDER=0 timeout -s KILL 200 python bench.py $AB > $O/bench_front_order0.json 2> $O/bench_front_order0.err
MI_BENCH_REGION_LOG=1 MI_DMRECON_SINGLE_FOLLOW=0 timeout -s KILL 200 python bench.py $AB > $O/bench_single_follow0.json 2> $O/bench_single_follow0.err
MI_BENCH_REGION_LOG=1 timeout -s KILL 200 python bench.py $AB > $O/bench_default_ab.json 2> $O/bench_default_ab.err
for f in front_order0 single_follow0 default_ab; do echo "== $f"; python - <<PY
import json
j = json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
print(j["value"], j["repeats"], "bulk frac", j["roofline"]["bulk_kernel_frac"])
PY
grep "^region" $O/bench_$f.err | tail -2; done
