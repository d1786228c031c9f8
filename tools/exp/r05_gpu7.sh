# explicit search arithmetic (wave == workgroup bit for bit?), _w1 latency product instances with the uniforms handed over
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_wave_proposal_kernels.py tests/test_gpu_uniform_proposal_kernels.py tests/test_golden.py tests/test_gpu_parity_ops.py tests/test_gpu_fused_update.py tests/test_gpu_tree_parity.py tests/test_gpu_stagewise_parity.py tests/test_gpu_native_host.py tests/test_gpu_clique_entry.py -m gpu -x -q -s 2>&1 | grep -E "wave vs workgroup|passed|failed|Error|error" | tail -15 > $O/pytest.txt
cat $O/pytest.txt
for spec in "1 2" "1 5" "1 8" "6 4" "12 3" "20 3" "66 2"; do set -- $spec
  python tools/exp/prod_batch.py $1 $2 2>/dev/null
done > $O/prod.txt
cat $O/prod.txt
for c in 2 3 4 5; do
  python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-10k 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c', j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['posterior_max_mean_err'])"
done > $O/bench.txt 2>&1
cat $O/bench.txt
bash tools/exp/clique_seam_rate.sh 2>&1 | grep -E "byte-identical|queued walk" | cut -c1-330 > $O/seam.txt
cat $O/seam.txt
