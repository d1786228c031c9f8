#!/bin/bash
# speculative fits under two lock-step processes on one GPU: rendezvous areas blanked by the library's own kernel (default
# build) against hipMemsetAsync (libnbp_memset.so)
OUT=gpurun_out/conc_probe3
mkdir -p $OUT
export NBP_BENCH_SHA=1 MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
one() { python bench.py --steps 2 --warmup 1 --config 2 --nvars 300 --no-cpu-baseline --no-10k --no-profile-pass > $OUT/$1.out 2> $OUT/$1.err
  echo "$1 rc=$? $(grep -h -o 'sha=[0-9a-f]*' $OUT/$1.err)"; }
w2() { timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
     bench.py --gpus 2 --steps 2 --warmup 1 --config 2 --nvars 300 --dist-backend gloo --no-cpu-baseline --no-profile-pass > $OUT/$1.out 2> $OUT/$1.err
  echo "$1 rc=$? $(grep -h -o 'rank [0-9]\] posterior_max_mean_err=[0-9.]* \|sha=[0-9a-f]*' $OUT/$1.err | sort | tr '\n' ' ')"; }
echo "== single process"; one default; NBP_NO_SPECULATIVE_FITS=1 one nospec
echo "== world 2, blank kernel (default build)"
for i in 1 2 3 4 5; do w2 k_$i; done
echo "== world 2, hipMemsetAsync build"
for i in 1 2 3; do NBP_LIB_OVERRIDE=$PWD/tools/exp/libnbp_memset.so w2 m_$i; done
echo "== world 2, no speculative fits"
NBP_NO_SPECULATIVE_FITS=1 w2 ns_1
