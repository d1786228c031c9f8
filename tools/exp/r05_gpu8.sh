# -ffp-contract=on (contraction inside a statement only: the same in every kernel an inlined function lands in) against the default (fast)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wave_proposal_kernels.py tests/test_gpu_uniform_proposal_kernels.py tests/test_golden.py tests/test_gpu_parity_ops.py -m gpu -x -q -s 2>&1 | grep -E "wave vs workgroup|passed|failed|Error|error" | tail -15 > $O/pytest.txt
cat $O/pytest.txt
for lib in on fast on fast; do
  if [ $lib = fast ]; then export NBP_LIB_OVERRIDE=$PWD/tools/exp/libnbp_fast.so; else unset NBP_LIB_OVERRIDE; fi
  for c in 2 3 4 5; do
    python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-10k 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib config $c', round(j['ms_per_step'],2), {k[4:-7]:round(v,2) for k,v in j['roofline']['kernel_ms_per_step'].items()}, j['posterior_max_mean_err'])"
  done
done > $O/bench.txt 2>&1
unset NBP_LIB_OVERRIDE
cat $O/bench.txt
bash tools/exp/clique_seam_rate.sh 2>&1 | grep -E "byte-identical|queued walk" | cut -c1-200 > $O/seam.txt
cat $O/seam.txt
