#!/bin/bash
# Runs ON THE GPU BOX (through gpurun).  Middle lane layout: its tests, then a threshold sweep on one host thread.
TAG=${1:-s40}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "lane_layouts or middle_layout" 2>&1 | tail -12
for T in 0 32768 49152 65536; do
  MI_DMRECON_MID_THRESHOLD=$T timeout -s KILL 120 python bench.py --no-cpu-baseline --streams 1 --steps-per-call 1 --steps 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']['per_kernel']
print('mid $T:', round(d['value'],1), 'maps/s', {k[:10]:(round(v['launches']/d['steps'],1), round(v['avg_launch_ms'],3)) for k,v in r.items()})"
done
