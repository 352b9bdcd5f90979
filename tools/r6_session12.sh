#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 6, twelfth session -- the LDS windows (-DMI_LDS_WINDOW, two wavefronts per SIMD) against
# the product and against the product's kernels at two wavefronts per SIMD: same maps (digests), driver's plan, lone calls.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r6l
mkdir -p $O
LIBS=${LIBS:-"- build/libmi_dmrecon_w2.so build/libmi_dmrecon_win.so - build/libmi_dmrecon_win.so"}
for L in "-" "build/libmi_dmrecon_win.so" "build/libmi_dmrecon_winL.so"; do
  T=$( [ "$L" = "-" ] && echo new || basename $L .so | sed 's/libmi_dmrecon_//' )
  MI_DMRECON_LIB=$( [ "$L" = "-" ] && echo "" || echo $R/$L ) timeout -s KILL 300 python tools/maps_digest.py C3 5 > $O/digest_$T.json 2> $O/digest_$T.err; cat $O/digest_$T.json; tail -2 $O/digest_$T.err
done
NOX="--no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant"
for L in $LIBS; do
  T=$( [ "$L" = "-" ] && echo new || basename $L .so | sed 's/libmi_dmrecon_//' )_$RANDOM
  MI_DMRECON_LIB=$( [ "$L" = "-" ] && echo "" || echo $R/$L ) MI_BENCH_REGION_LOG=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --repeats 3 $NOX > $O/bench_$T.json 2> $O/bench_$T.err
  echo "$T driver plan: $(python -c "import json,sys; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print(round(d['value'],1), [round(x) for x in d['repeats']])")"
  grep region $O/bench_$T.err | tail -1
  MI_DMRECON_LIB=$( [ "$L" = "-" ] && echo "" || echo $R/$L ) timeout -s KILL 200 python bench.py --streams 1 --steps-per-call 1 --steps 20 --warmup 3 --repeats 3 $NOX > $O/lone_$T.json 2> $O/lone_$T.err
  echo "$T lone calls: $(python -c "import json,sys; d=json.loads(open('$O/lone_$T.json').read().strip().splitlines()[-1]); r=d['roofline']['per_kernel']; print(round(d['value'],1), round(d['ms_per_step'],2), 'bulk ms/step', round(r['k_optimize<1> (host-visible rounds)']['avg_launch_ms']*r['k_optimize<1> (host-visible rounds)']['launches']/60,2), 'front', round(r['k_tail + k_front (tail rounds)']['k_front_ms']/60,2))")"
done
du -sh $O
