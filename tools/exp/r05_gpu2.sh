# wave-per-proposal geometry against the workgroup geometry; product launches by size and chunks per range
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_wave_proposal_kernels.py tests/test_gpu_uniform_proposal_kernels.py -m gpu -x -q -s 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
{
for spec in "lin2 200" "lin3 200" "lin3 300"; do set -- $spec
  for B in 975 2000 4000 9750; do
    NBP_PROPOSAL_WAVE_MIN=100000000 python tools/exp/prop_batch.py $B $2 $1 | sed 's/^/workgroup /'
    NBP_PROPOSAL_WAVE_MIN=1 python tools/exp/prop_batch.py $B $2 $1 | sed 's/^/wave      /'
  done
done
} > $O/prop_wave.txt 2>&1
cat $O/prop_wave.txt
{
for n in 245 332 488 738 975; do
  python tools/exp/prod_batch.py $n 2 | sed 's/^/nch default  /'
  for k in 2 4 8; do NBP_PRODUCT_NCH=$k python tools/exp/prod_batch.py $n 2 | sed "s/^/nch $k        /"; done
done
for n in 372 488; do
  python tools/exp/prod_batch.py $n 3 | sed 's/^/nch default  /'
  for k in 2 4 8; do NBP_PRODUCT_NCH=$k python tools/exp/prod_batch.py $n 3 | sed "s/^/nch $k        /"; done
done
} > $O/prod_nch.txt 2>&1
cat $O/prod_nch.txt
for w in 100000000 3000 1500; do
  NBP_PROPOSAL_WAVE_MIN=$w python bench.py --config 2p --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wave_min $w', j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['posterior_max_mean_err'])"
done > $O/bench10k.txt 2>&1
cat $O/bench10k.txt
