#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the whole GPU suite, the smoke test, and the bench with the RCCL path forced on one rank.
export TMPDIR=/tmp
mkdir -p gpurun_out/r3ap
timeout -s KILL 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
MI_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 timeout -s KILL 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3ap/bench_force_dist.json 2> gpurun_out/r3ap/bench_force_dist.err
tail -c 300 gpurun_out/r3ap/bench_force_dist.err; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3ap/bench_force_dist.json').read().strip().splitlines()[-1])
print('forced RCCL path:', round(d['value'], 1), d['unit'], 'n_gpus', d['n_gpus'], 'one_call', round(d['one_call']['depth_maps_per_s'], 1), 'batches', d['config'].get('library_batch_log'))
PY
