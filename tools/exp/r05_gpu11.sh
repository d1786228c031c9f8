cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
# the bench's own config-5 leg at 400 variables (one prior): single process, seeds 0..15 -- and the 4-rank shared-GPU leg at seeds 0..3
for s in 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15; do
  NBP_BENCH_SEED=$s python bench.py --config 5 --nvars 400 --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass 2>&1 | grep -E "^\{|result invalid" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('single seed $s err', json.loads(l)['posterior_max_mean_err'])
    else: print('single seed $s', l.strip()[-120:])"
done > $O/bench5_seeds.txt 2>&1
for s in 0 1 2 3; do
  NBP_BENCH_SEED=$s HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $((29600+s)) bench.py --gpus 4 --steps 2 --warmup 1 --config 5 --nvars 400 --dist-backend gloo --no-cpu-baseline 2>&1 | grep -E "bench rank . posterior_max|result invalid" | sed "s/^/world4 seed $s /" | cut -c1-200
done >> $O/bench5_seeds.txt 2>&1
cat $O/bench5_seeds.txt
gcc -O2 -Wall -fopenmp -I include examples/solve_by_clique_calls.c -o /tmp/sbcc -L incrementalinference.jl_amd/csrc -lnbp -lm || exit 1
export LD_LIBRARY_PATH=$PWD/incrementalinference.jl_amd/csrc:/opt/rocm/lib:$LD_LIBRARY_PATH
for k in 1 2; do /tmp/sbcc 1000 200 100 -1 2>&1 | grep -v amdgpu.ids | cut -c1-460; done > $O/seam.txt
NBP_SEAM_TIMES=1 /tmp/sbcc 1000 200 100 -1 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/seam.txt
/tmp/sbcc 1000 200 100 0 2>&1 | grep -v amdgpu.ids | cut -c1-460 >> $O/seam.txt
cat $O/seam.txt
