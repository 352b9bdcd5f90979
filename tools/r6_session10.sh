#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 6, tenth session -- column-pair footprint elements (8 bytes per texel position, one
# 16-byte gather per sample from an 8-byte aligned address) against the 16-byte records; one wavefront per SIMD.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r6j
mkdir -p $O
NOX="--no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant"
MI_DMRECON_LIB=$R/build/libmi_dmrecon_pair.so timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -x -q -k "patch_sampler or vs_reference or pyramid or wider" > $O/pytest_pair.log 2>&1; tail -3 $O/pytest_pair.log
for L in "" "build/libmi_dmrecon_pair.so" "build/libmi_dmrecon_w1.so" "" "build/libmi_dmrecon_pair.so"; do
  T=$( [ -z "$L" ] && echo new || basename $L .so | sed 's/libmi_dmrecon_//' )_$RANDOM
  MI_DMRECON_LIB=$( [ -z "$L" ] && echo "" || echo $R/$L ) MI_BENCH_REGION_LOG=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --repeats 3 $NOX > $O/bench_$T.json 2> $O/bench_$T.err
  echo "$T driver plan: $(python -c "import json,sys; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print(round(d['value'],1), [round(x) for x in d['repeats']])")"
  grep region $O/bench_$T.err | tail -1
  MI_DMRECON_LIB=$( [ -z "$L" ] && echo "" || echo $R/$L ) timeout -s KILL 200 python bench.py --streams 1 --steps-per-call 1 --steps 20 --warmup 3 --repeats 3 $NOX > $O/lone_$T.json 2> $O/lone_$T.err
  echo "$T lone calls: $(python -c "import json,sys; d=json.loads(open('$O/lone_$T.json').read().strip().splitlines()[-1]); r=d['roofline']['per_kernel']; print(round(d['value'],1), round(d['ms_per_step'],2), 'bulk ms/step', round(r['k_optimize<1> (host-visible rounds)']['avg_launch_ms']*r['k_optimize<1> (host-visible rounds)']['launches']/60,2), 'front', round(r['k_tail + k_front (tail rounds)']['k_front_ms']/60,2))")"
done
MI_DMRECON_LIB=$R/build/libmi_dmrecon_pair.so MI_BENCH_REGION_LOG=1 timeout -s KILL 300 python bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-one-call --no-seed-variant > $O/bench_pair_distinct.json 2> $O/bench_pair_distinct.err
python -c "import json; d=json.loads(open('$O/bench_pair_distinct.json').read().strip().splitlines()[-1]); print('pair: value', round(d['value'],1), 'distinct', d['config']['distinct_scenes_variant']['value'])"
du -sh $O
