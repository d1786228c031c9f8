set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_parity_ops.py tests/test_gpu_stagewise_parity.py tests/test_gpu_fused_update.py tests/test_gpu_analytic_known_answers.py -m gpu -x -q 2>&1 | tail -15
for n in 2 4 8; do NBP_PRODUCT_NCH=$n python tools/exp/prod_batch.py 975 2; done
python tools/exp/prod_batch.py 975 2
python tools/exp/prod_batch.py 975 3
python tools/exp/prod_batch.py 1 2
python tools/exp/prod_batch.py 100 2
bash tools/exp/ab_bench.sh r05a
