#!/bin/bash
# Runs ON THE GPU BOX: round 6 -- 24 dependent v_fma_f32 per sample on top of the ~107 VALU instructions of the sampling loop
# (-DMI_EXTRA_FMA=24): are the bulk kernels bound by their VALU instructions after all?  (Session H asked with v_mov_b32.)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r6s
mkdir -p $O
NOX="--no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant"
for L in "-" "build/libmi_dmrecon_fma24.so" "-" "build/libmi_dmrecon_fma24.so"; do
  T=$( [ "$L" = "-" ] && echo new || basename $L .so | sed 's/libmi_dmrecon_//' )_$RANDOM
  MI_DMRECON_LIB=$( [ "$L" = "-" ] && echo "" || echo $R/$L ) MI_BENCH_REGION_LOG=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --repeats 3 $NOX > $O/bench_$T.json 2> $O/bench_$T.err
  echo "$T driver plan: $(python -c "import json,sys; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print(round(d['value'],1), [round(x) for x in d['repeats']])")"
  grep region $O/bench_$T.err | tail -1 | sed 's/.*kernels/kernels/'
  MI_DMRECON_LIB=$( [ "$L" = "-" ] && echo "" || echo $R/$L ) timeout -s KILL 200 python bench.py --streams 1 --steps-per-call 1 --steps 20 --warmup 3 --repeats 3 $NOX > $O/lone_$T.json 2> $O/lone_$T.err
  echo "$T lone calls: $(python -c "import json,sys; d=json.loads(open('$O/lone_$T.json').read().strip().splitlines()[-1]); r=d['roofline']['per_kernel']; print(round(d['ms_per_step'],2), 'bulk ms/step', round(r['k_optimize<1> (host-visible rounds)']['avg_launch_ms']*r['k_optimize<1> (host-visible rounds)']['launches']/60,2))")"
done
