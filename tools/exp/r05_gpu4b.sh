# after dropping the unused label row of the product LDS (two chunks per range and two workgroups per CU at three densities)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wave_proposal_kernels.py tests/test_golden.py tests/test_gpu_parity_ops.py tests/test_gpu_fused_update.py tests/test_gpu_tree_parity.py -m gpu -x -q 2>&1 | tail -5 > $O/pytest.txt
cat $O/pytest.txt
{
for n in 372 488 738; do python tools/exp/prod_batch.py $n 3 | sed 's/^/nch default  /'; done
python tools/exp/prod_batch.py 488 2 | sed 's/^/nch default  /'
} 2>/dev/null > $O/prod_nch.txt
cat $O/prod_nch.txt
for c in 2 3 4; do
  python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-10k 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c', j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['posterior_max_mean_err'])"
done > $O/bench.txt 2>&1
cat $O/bench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-10k --no-profile-pass > /dev/null 2> $GRAFT_REPO_ROOT/$O/trace.err
cd $GRAFT_REPO_ROOT
python tools/stage_timeline.py $O/trace > $O/solve_launches_in_order.txt
python tools/summarize_trace.py $O/trace 5 > $O/bench_timed_region_summary.txt
rm -rf $O/trace
grep product $O/solve_launches_in_order.txt | head -12
cat $O/bench_timed_region_summary.txt
