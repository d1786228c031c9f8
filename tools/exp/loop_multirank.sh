#!/bin/bash
# loop_multirank.sh REPS: the multi-rank-on-one-GPU legs of tests/test_gpu_bench.py, REPS times each, every rank's stderr kept
REPS=${1:-10}
OUT=gpurun_out/mr_loop
rm -rf $OUT; mkdir -p $OUT
export NBP_BENCH_SHA=1
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
fail=0
for i in $(seq 1 $REPS); do
  for case in "5 400 4" "2 300 2" "4 6 2" "4 8 8" "5 400 8" "3 200 4"; do
    set -- $case
    tag="c$1_s$2_w$3_$i"
    timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $3 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
       bench.py --gpus $3 --steps 2 --warmup 1 --config $1 --nvars $2 --dist-backend gloo --no-cpu-baseline > $OUT/$tag.out 2> $OUT/$tag.err
    rc=$?
    echo "$tag rc=$rc $(grep -h 'posterior_max_mean_err' $OUT/$tag.err | sed 's/(.*//; s/posterior_max_mean_err=/err /; s/mode_share=None//' | sort | tr '\n' ' ')" | tee -a $OUT/summary.txt
    if [ $rc -ne 0 ]; then fail=$((fail+1)); grep -h "bench rank" $OUT/$tag.err | tail -30 | tee -a $OUT/summary.txt; else rm -f $OUT/$tag.out; fi
  done
done
echo "failures: $fail" | tee -a $OUT/summary.txt
