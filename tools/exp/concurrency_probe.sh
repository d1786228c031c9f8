#!/bin/bash
# concurrency_probe.sh: are the posteriors of a solve the same when other processes use the GPU at the same time?
# (sha over the posterior particles; bench.py prints it with NBP_BENCH_SHA=1)
OUT=gpurun_out/conc_probe
mkdir -p $OUT
export NBP_BENCH_SHA=1 MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
one() { # tag config nvars [extra env]
  python bench.py --steps 2 --warmup 1 --config $2 --nvars $3 --no-cpu-baseline --no-10k --no-profile-pass > $OUT/$1.out 2> $OUT/$1.err
  echo "$1 rc=$? $(grep -h -o 'sha=[0-9a-f]*' $OUT/$1.err)"
}
for cfg in "2 300" "5 400" "4 6"; do
  set -- $cfg
  echo "== config $1 size $2: alone, twice"
  one a1_c$1 $1 $2; one a2_c$1 $1 $2
  for rep in 1 2 3; do
    echo "== config $1 size $2: four at once (rep $rep)"
    for k in 1 2 3 4; do one p${rep}_${k}_c$1 $1 $2 & done; wait
  done
  echo "== config $1 size $2: four at once, no speculative fits"
  for k in 1 2 3 4; do NBP_NO_SPECULATIVE_FITS=1 one q${k}_c$1 $1 $2 & done; wait
done
echo "== world 2 (gloo, shared GPU), config 2 size 300: sha per rank, 5 runs"
for i in 1 2 3 4 5; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
     bench.py --gpus 2 --steps 2 --warmup 1 --config 2 --nvars 300 --dist-backend gloo --no-cpu-baseline > $OUT/w2_$i.out 2> $OUT/w2_$i.err
  echo "w2_$i rc=$? $(grep -h -o 'rank [0-9]\] posterior_max_mean_err=[0-9.]* \|sha=[0-9a-f]*' $OUT/w2_$i.err | tr '\n' ' ')"
done
echo "== the same with NBP_NO_SPECULATIVE_FITS=1"
for i in 1 2 3 4 5; do
  NBP_NO_SPECULATIVE_FITS=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
     bench.py --gpus 2 --steps 2 --warmup 1 --config 2 --nvars 300 --dist-backend gloo --no-cpu-baseline > $OUT/w2n_$i.out 2> $OUT/w2n_$i.err
  echo "w2n_$i rc=$? $(grep -h -o 'rank [0-9]\] posterior_max_mean_err=[0-9.]* \|sha=[0-9a-f]*' $OUT/w2n_$i.err | tr '\n' ' ')"
done
