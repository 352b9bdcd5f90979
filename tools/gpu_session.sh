#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): front workgroups (next to other calls) with fewer registers, so that bulk wavefronts fit beside them.
TAG=${1:-r3aq}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
show() {
  python - $1 $2 <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); t = d['roofline']['per_kernel']['k_tail + k_front (tail rounds)']; b = d['roofline']['per_kernel']['k_optimize<1> (host-visible rounds)']
    print('%-14s' % sys.argv[2], round(d['value'], 1), 'maps/s', 'ms/step', round(d['ms_per_step'], 2), 'bulk ms', round(b['avg_launch_ms'] * b['launches'] / d['steps'], 2), 'frac', round(d['roofline']['bulk_kernel_frac'], 3), 'k_front', round(t['k_front_ms'] / d['steps'], 2))
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
drv() { L=$1; shift; env "$@" timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-call 2>/dev/null > $OUT/drv_$L.json; show $OUT/drv_$L.json drv_$L; }
dfl() { L=$1; shift; env "$@" timeout -s KILL 300 python bench.py --no-cpu-baseline --no-one-call 2>/dev/null > $OUT/dfl_$L.json; show $OUT/dfl_$L.json dfl_$L; }
dfl base
dfl fs3 MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_fs3.so
dfl fs4 MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_fs4.so
drv base
drv fs3 MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_fs3.so
drv fs4 MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_fs4.so
dfl base_2
dfl fs3_2 MI_DMRECON_LIB=$PWD/build/libmi_dmrecon_fs3.so
