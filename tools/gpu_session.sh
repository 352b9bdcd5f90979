#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the whole GPU suite, then the driver's command three times.
export TMPDIR=/tmp
mkdir -p gpurun_out/r3av
timeout -s KILL 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for R in 1 2 3; do
  timeout -s KILL 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-one-call 2>/dev/null > gpurun_out/r3av/drv_$R.json
  python - gpurun_out/r3av/drv_$R.json $R <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('run', sys.argv[2], round(d['value'], 1), 'depth-maps/s', d['config'].get('library_batch_log'))
PY
done
