#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): bench lines + rocprofv3 kernel stats + PMC passes for the C3 bench.
# Everything lands in gpurun_out/$TAG/; tools/summarize_profiles.py turns it into profiles/<tag>_*.
# --pmc passes are separate runs with --kernel-trace only (never with sys/runtime/hip traces), each under its own
# timeout: a counter group the hardware cannot collect makes rocprofv3 abort and then hang in its finalisation.
TAG=${1:-r3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err      # the driver's command
python bench.py --no-one-call > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --streams 1 --steps-per-call 1 --steps 20 --no-cpu-baseline --no-one-call > $OUT/bench_1thread.json 2> $OUT/bench_1thread.err
# one host thread, the default five steps (100 reference views) per call: what the size of a launch does to the bulk kernel
python bench.py --streams 1 --steps 20 --warmup 5 --no-cpu-baseline --no-one-call > $OUT/bench_1thread_5steps.json 2> $OUT/bench_1thread_5steps.err
B1="python bench.py --steps 6 --warmup 1 --streams 1 --steps-per-call 1 --no-cpu-baseline --no-one-call"
B3="python bench.py --steps 30 --warmup 1 --no-cpu-baseline --no-one-call"
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s1 -o bench -- $B1 > $OUT/s1.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s3 -o bench -- $B3 > $OUT/s3.log 2>&1
# PMC at the call plan of the 1-thread line (one step per call) and, for the traffic figure of the default line,
# at the default plan (6 threads, 5 steps per call)
BP="python bench.py --steps 2 --warmup 1 --streams 1 --steps-per-call 1 --no-cpu-baseline --no-one-call"
BQ="python bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-one-call"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  D=$OUT/pmc_$(echo $C | tr ' ' '+')
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o bench -- $BP > $D.log 2>&1
done
for C in FETCH_SIZE WRITE_SIZE; do
  D=$OUT/pmcd_$C
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o bench -- $BQ > $D.log 2>&1
done
# keep the merge small: drop the per-dispatch traces of the stats runs (the stats CSV is the summary)
rm -f $OUT/s1/bench_kernel_trace.csv $OUT/s3/bench_kernel_trace.csv
find $OUT -name "*_kernel_trace.csv" -path "*pmc*" -delete
du -sh $OUT
cat $OUT/bench_driver.json $OUT/bench_default.json $OUT/bench_1thread.json $OUT/bench_1thread_5steps.json | cut -c1-600
# a round trace of one lone C3 call, the front-kernel patch probe, the drop-in app on the scene on disk
timeout -s KILL 120 python tools/trace_c3.py > $OUT/round_trace_c3.txt 2>&1
timeout -s KILL 200 python tools/patch_probe.py > $OUT/patch_probe.txt 2>&1
MI_DMRECON_TRACE=1 timeout -s KILL 300 python tools/app_c3_timing.py 2>&1 | grep -v "^\[mi_dmrecon\]" > $OUT/app_c3_timing.txt
# lone calls of 1..20 views (what a rank's share of a scene costs when the rank has its GPU to itself), a cold first call
for N in 1 2 3 5 10 20; do
  timeout -s KILL 100 python tools/trace_c3.py C3 $N 2>&1 | grep -E "phase|total|wall" | sed 's/\[mi_dmrecon\] //' | tr '\n' ';' | cut -c1-420 | sed "s/^/lone call of $N views: /"; echo
done > $OUT/lone_calls.txt
timeout -s KILL 200 python tools/cold_call.py C3 20 2>&1 | grep -E "phase|==|context|staged|total" | cut -c1-120 > $OUT/cold_call.txt
# BASELINE config 4 (the 20 views of ONE scene sharded over the ranks) on the one GPU of this box: two ranks sharing GPU 0
# (development mode of bench.py, gloo): what a rank's share of the scene costs -- the strong-scaling prediction
MI_BENCH_SHARE_GPU=1 timeout -s KILL 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29641 \
  bench.py --gpus 2 --steps 20 --warmup 5 --scaling strong --no-cpu-baseline > $OUT/strong_2ranks_one_gpu.json 2> $OUT/strong_2ranks_one_gpu.err
MI_BENCH_SHARE_GPU=1 timeout -s KILL 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29642 \
  bench.py --gpus 4 --steps 20 --warmup 5 --scaling strong --no-cpu-baseline > $OUT/strong_4ranks_one_gpu.json 2> $OUT/strong_4ranks_one_gpu.err
tail -c 600 $OUT/strong_2ranks_one_gpu.json; tail -c 600 $OUT/strong_4ranks_one_gpu.json
grep -v '(view)' $OUT/app_c3_timing.txt | tail -12; cat $OUT/lone_calls.txt | cut -c1-300
