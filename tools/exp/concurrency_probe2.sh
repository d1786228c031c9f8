#!/bin/bash
# which part of the speculative fits is timing-dependent?  libnbp variants: patience 1 (roles give up at once: everybody
# searches alone), patience 4M polls (nobody ever gives up), default (512)
OUT=gpurun_out/conc_probe2
mkdir -p $OUT
export NBP_BENCH_SHA=1 MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
one() { python bench.py --steps 2 --warmup 1 --config 2 --nvars 300 --no-cpu-baseline --no-10k --no-profile-pass > $OUT/$1.out 2> $OUT/$1.err
  echo "$1 rc=$? $(grep -h -o 'sha=[0-9a-f]*' $OUT/$1.err)"; }
w2() { timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
     bench.py --gpus 2 --steps 2 --warmup 1 --config 2 --nvars 300 --dist-backend gloo --no-cpu-baseline --no-profile-pass > $OUT/$1.out 2> $OUT/$1.err
  echo "$1 rc=$? $(grep -h -o 'rank [0-9]\] posterior_max_mean_err=[0-9.]* \|sha=[0-9a-f]*' $OUT/$1.err | sort | tr '\n' ' ')"; }
echo "== single process"
one default
NBP_NO_SPECULATIVE_FITS=1 one nospec
NBP_LIB_OVERRIDE=$PWD/tools/exp/libnbp_p1.so one p1
NBP_LIB_OVERRIDE=$PWD/tools/exp/libnbp_pinf.so one pinf
echo "== world 2, default library, no profiling pass"
for i in 1 2 3; do w2 d_$i; done
echo "== world 2, patience 1"
for i in 1 2 3; do NBP_LIB_OVERRIDE=$PWD/tools/exp/libnbp_p1.so w2 p1_$i; done
echo "== world 2, patience 4M"
for i in 1 2 3; do NBP_LIB_OVERRIDE=$PWD/tools/exp/libnbp_pinf.so w2 pinf_$i; done
echo "== world 2, 3 workgroups per fit only"
for i in 1 2 3; do NBP_SPEC_DEPTH3=0 w2 d3_$i; done
echo "== world 2, no speculative fits"
for i in 1 2; do NBP_NO_SPECULATIVE_FITS=1 w2 ns_$i; done
