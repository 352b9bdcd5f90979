#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 6, seventh session -- the passes of iterations 1..3 without colour sums and a colour pass
# after a normal step (GPU suite; A/B against the build before), the 48-byte-record memory emulation with its loads kept in flight.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r6g
mkdir -p $O
NOX="--no-cpu-baseline --no-one-call --distinct-scenes 0 --no-seed-variant"
( timeout -s KILL 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log ); tail -6 $O/pytest_gpu.log
MI_TEST_PRINT=1 timeout -s KILL 100 python -m pytest tests/test_gpu_parity.py -q -s -k "scale1_odd" 2>&1 | grep "G1b"
for L in "" "build/libmi_dmrecon_prev.so" "" "build/libmi_dmrecon_prev.so" "build/libmi_dmrecon_lin48.so" "build/libmi_dmrecon_lin48w2.so"; do
  T=$( [ -z "$L" ] && echo new || basename $L .so | sed 's/libmi_dmrecon_//' )_$RANDOM
  MI_DMRECON_LIB=$( [ -z "$L" ] && echo "" || echo $R/$L ) MI_BENCH_REGION_LOG=1 timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --repeats 3 $NOX > $O/bench_$T.json 2> $O/bench_$T.err
  echo "$T driver plan: $(python -c "import json,sys; d=json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); print(round(d['value'],1), [round(x) for x in d['repeats']])")"
  grep region $O/bench_$T.err | tail -1
  MI_DMRECON_LIB=$( [ -z "$L" ] && echo "" || echo $R/$L ) timeout -s KILL 200 python bench.py --streams 1 --steps-per-call 1 --steps 20 --warmup 3 --repeats 3 $NOX > $O/lone_$T.json 2> $O/lone_$T.err
  echo "$T lone calls: $(python -c "import json,sys; d=json.loads(open('$O/lone_$T.json').read().strip().splitlines()[-1]); r=d['roofline']['per_kernel']; print(round(d['value'],1), round(d['ms_per_step'],2), 'bulk ms/step', round(r['k_optimize<1> (host-visible rounds)']['avg_launch_ms']*r['k_optimize<1> (host-visible rounds)']['launches']/60,2), 'front', round(r['k_tail + k_front (tail rounds)']['k_front_ms']/60,2))")"
done
MI_BENCH_REGION_LOG=1 timeout -s KILL 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
grep "region" $O/bench_driver.err | tail -2
python - <<PY
import json
d = json.loads(open("$O/bench_driver.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"], 1), [round(x) for x in d["repeats"]], "frac", round(r["frac"], 3), "bulk", round(r["bulk_kernel_frac"], 3), "clock", r["shader_clock_mhz_measured"])
print("one_call", d["one_call"]["ms_per_call"], d["one_call"]["ms_front_kernel"], d["one_call"]["ms_bulk_kernel"])
print("distinct", d["config"]["distinct_scenes_variant"]["value"], "seedvar", d["config"]["all_seeds_propagate_variant"]["value"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["cpu_baseline"].get("drop_in_app_same_scene"))
print("parity", json.dumps({k: d["parity"][k] for k in ("within_bounds", "min_fill_iou", "max_rel_depth_median", "max_rel_depth_p99", "max_conf_abs_p99")}), d["parity"]["fill_iou_per_view"])
PY
du -sh $O
