#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): bench lines + rocprofv3 kernel stats + PMC passes for the C3 bench.
# Everything lands in gpurun_out/$TAG/; tools/summarize_profiles.py turns it into profiles/<tag>_*.
# --pmc passes are separate runs with --kernel-trace only (never with sys/runtime/hip traces), each under its own
# timeout: a counter group the hardware cannot collect makes rocprofv3 abort and then hang in its finalisation.
TAG=${1:-r4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err      # the driver's command
python bench.py --steps 60 --repeats 3 --no-cpu-baseline --no-one-call > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --streams 1 --steps-per-call 1 --steps 20 --repeats 3 --no-cpu-baseline --no-one-call > $OUT/bench_1thread.json 2> $OUT/bench_1thread.err
# one host thread, five steps (100 reference views) per call: what the size of a launch does to the bulk kernel
python bench.py --streams 1 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-one-call > $OUT/bench_1thread_5steps.json 2> $OUT/bench_1thread_5steps.err
B1="python $R/bench.py --steps 6 --warmup 1 --repeats 1 --streams 1 --steps-per-call 1 --no-cpu-baseline --no-one-call"
B3="python $R/bench.py --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-one-call"
cd /tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/s1 -o bench -- $B1 > $R/$OUT/s1.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/s3 -o bench -- $B3 > $R/$OUT/s3.log 2>&1
# PMC at the call plan of the 1-thread line (one step per call) and, for the traffic figure of the driver's line, at its plan
BP="python $R/bench.py --steps 2 --warmup 1 --repeats 1 --streams 1 --steps-per-call 1 --no-cpu-baseline --no-one-call"
BQ="python $R/bench.py --steps 10 --warmup 1 --repeats 1 --no-cpu-baseline --no-one-call"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  D=$R/$OUT/pmc_$(echo $C | tr ' ' '+')
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o bench -- $BP > $D.log 2>&1
done
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  D=$R/$OUT/pmcd_$(echo $C | tr ' ' '+')
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o bench -- $BQ > $D.log 2>&1
done
cd $R
# keep the merge small: drop the per-dispatch traces of the stats runs (the stats CSV is the summary)
rm -f $OUT/s1/bench_kernel_trace.csv $OUT/s3/bench_kernel_trace.csv
find $OUT -name "*_kernel_trace.csv" -path "*pmc*" -delete
du -sh $OUT
cat $OUT/bench_driver.json $OUT/bench_default.json $OUT/bench_1thread.json $OUT/bench_1thread_5steps.json | cut -c1-400
# a round trace of one lone C3 call, the drop-in app on the scene on disk, lone calls of a rank's share, a cold first call
timeout -s KILL 120 python tools/trace_c3.py > $OUT/round_trace_c3.txt 2>&1
MI_DMRECON_TRACE=1 timeout -s KILL 300 python tools/app_c3_timing.py 2>&1 | grep -v "^\[mi_dmrecon\]" > $OUT/app_c3_timing.txt
timeout -s KILL 300 python tools/lone_calls.py C3 12 > $OUT/lone_calls.json 2> $OUT/lone_calls.err
timeout -s KILL 200 python tools/cold_call.py C3 20 2>&1 | grep -E "phase|==|context|staged|total" | cut -c1-120 > $OUT/cold_call.txt
timeout -s KILL 90 build/valu_rate2 > $OUT/valu_rate.txt 2>&1
timeout -s KILL 200 python tools/big_batch_probe.py 20 > $OUT/big_batch.txt 2>&1
# BASELINE config 4 (the 20 views of ONE scene sharded over the ranks) on the one GPU of this box: ranks sharing GPU 0
# (development mode of bench.py, gloo; default environment: the team token and the give-up path do their work)
MI_BENCH_SHARE_GPU=1 timeout -s KILL 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29641 \
  bench.py --gpus 2 --steps 20 --warmup 5 --repeats 3 --scaling strong --no-cpu-baseline > $OUT/strong_2ranks_one_gpu.json 2> $OUT/strong_2ranks_one_gpu.err
tail -c 700 $OUT/strong_2ranks_one_gpu.json
grep -v '(view)' $OUT/app_c3_timing.txt | tail -12; cut -c1-300 $OUT/lone_calls.json
