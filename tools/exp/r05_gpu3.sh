# wave-per-proposal: test, small launches, lin3 at four waves per SIMD; product launches of mixed F after the nch = 1 fallback; config 2 / 5 / 10k
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_wave_proposal_kernels.py tests/test_gpu_uniform_proposal_kernels.py -m gpu -x -q -s 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
{
for B in 128 245 488 700; do
  NBP_PROPOSAL_WAVE_MIN=100000000 python tools/exp/prop_batch.py $B 200 lin2 | sed 's/^/workgroup /'
  NBP_PROPOSAL_WAVE_MIN=1 python tools/exp/prop_batch.py $B 200 lin2 | sed 's/^/wave      /'
done
for spec in "lin3 200" "lin3 300"; do set -- $spec
  for B in 975 2000 4000 9750; do
    NBP_PW_LIN3_W4=1 NBP_PROPOSAL_WAVE_MIN=1 python tools/exp/prop_batch.py $B $2 $1 | sed 's/^/wave w4   /'
  done
done
} 2>/dev/null > $O/prop_wave.txt
cat $O/prop_wave.txt
{
for n in 372 488 738; do
  python tools/exp/prod_batch.py $n 3 | sed 's/^/nch default  /'
  for k in 1 2; do NBP_PRODUCT_NCH=$k python tools/exp/prod_batch.py $n 3 | sed "s/^/nch $k        /"; done
done
NBP_PRODUCT_NCH=1 python tools/exp/prod_batch.py 488 2 | sed "s/^/nch 1        /"
} 2>/dev/null > $O/prod_nch.txt
cat $O/prod_nch.txt
for c in 2 2p 5; do
for w in 100000000 -1; do
  NBP_PROPOSAL_WAVE_MIN=$w python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-10k 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c wave_min $w', j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['posterior_max_mean_err'])"
done; done > $O/bench.txt 2>&1
cat $O/bench.txt
