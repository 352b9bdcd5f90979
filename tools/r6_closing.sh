#!/bin/bash
# Runs ON THE GPU BOX: round 6, the closing pass on the final commit -- what the driver runs at round end (GPU suite, smoke, the
# bench command), then the plain N = 2 command on one GPU, the drop-in binary's timing and the run-to-run spread of the headline.
export TMPDIR=/tmp
O=gpurun_out/r6end
mkdir -p $O
( timeout -s KILL 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log
( timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log ); tail -2 $O/smoke.log
MI_BENCH_REGION_LOG=1 timeout -s KILL 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-400 $O/bench_driver.json
( MI_BENCH_SHARE_GPU=1 timeout -s KILL 400 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/strong_2ranks_one_gpu.json 2> $O/strong_2ranks_one_gpu.err; echo "rc $?" ); tail -c 700 $O/strong_2ranks_one_gpu.json
timeout -s KILL 300 python tools/app_c3_timing.py > $O/app_c3_timing.txt 2>&1; tail -3 $O/app_c3_timing.txt
for R in 1 2 3 4; do
  timeout -s KILL 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant 2>/dev/null > $O/drv_$R.json
  python -c "import json,sys; d=json.loads(open('$O/drv_$R.json').read().strip().splitlines()[-1]); print('run $R', round(d['value'],1), [round(x) for x in d['repeats']])" | tee -a $O/driver_repeat.txt
done
