#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 6, ninth session -- latency hiding in the throughput layout: the later rows' cache lines
# asked for at the start of a pass (touch), two rows of gathers in flight at two wavefronts per SIMD (row2w2; w2 = its baseline).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r6i
mkdir -p $O
NOX="--no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant"
for L in "" "build/libmi_dmrecon_touch.so" "build/libmi_dmrecon_w2.so" "build/libmi_dmrecon_row2w2.so" "" "build/libmi_dmrecon_touch.so"; do
  T=$( [ -z "$L" ] && echo new || basename $L .so | sed 's/libmi_dmrecon_//' )_$RANDOM
  MI_DMRECON_LIB=$( [ -z "$L" ] && echo "" || echo $R/$L ) MI_BENCH_REGION_LOG=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --repeats 3 $NOX > $O/bench_$T.json 2> $O/bench_$T.err
  echo "$T driver plan: $(python -c "import json,sys; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print(round(d['value'],1), [round(x) for x in d['repeats']])")"
  grep region $O/bench_$T.err | tail -1
  MI_DMRECON_LIB=$( [ -z "$L" ] && echo "" || echo $R/$L ) timeout -s KILL 200 python bench.py --streams 1 --steps-per-call 1 --steps 20 --warmup 3 --repeats 3 $NOX > $O/lone_$T.json 2> $O/lone_$T.err
  echo "$T lone calls: $(python -c "import json,sys; d=json.loads(open('$O/lone_$T.json').read().strip().splitlines()[-1]); r=d['roofline']['per_kernel']; print(round(d['value'],1), round(d['ms_per_step'],2), 'bulk ms/step', round(r['k_optimize<1> (host-visible rounds)']['avg_launch_ms']*r['k_optimize<1> (host-visible rounds)']['launches']/60,2), 'front', round(r['k_tail + k_front (tail rounds)']['k_front_ms']/60,2))")"
done
du -sh $O
