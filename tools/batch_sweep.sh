#!/bin/bash
for spc in 1 2 4; do for st in 1 2 3; do
  v=$(python bench.py --steps 24 --warmup 1 --streams $st --steps-per-call $spc --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f maps/s  %.2f ms/step  opt-share %.2f fill %.4f' % (j['value'], j['ms_per_step'], j['roofline']['kernel_time_share'], j['config']['mean_fill']))" 2>&1 | tail -1)
  echo "steps-per-call=$spc streams=$st: $v"
done; done
