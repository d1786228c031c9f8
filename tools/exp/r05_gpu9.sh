cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8) > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
R=$GRAFT_REPO_ROOT
gcc -O2 -Wall -fopenmp -I $R/include $R/examples/solve_by_clique_calls.c -o /tmp/sbcc -L $R/incrementalinference.jl_amd/csrc -lnbp -lm || exit 1
export LD_LIBRARY_PATH=$R/incrementalinference.jl_amd/csrc:/opt/rocm/lib:$LD_LIBRARY_PATH
{
for c in 16 32 64; do for q in 16 32; do GPU_MAX_HW_QUEUES=$q /tmp/sbcc 1000 200 100 $c 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/callers $c queues $q: /" | cut -c1-420; done; done
NBP_SEAM_TIMES=1 /tmp/sbcc 1000 200 100 -1 2>&1 | grep -v amdgpu.ids | tail -3
NBP_SEAM_TIMES=1 /tmp/sbcc 1000 200 100 0 2>&1 | grep -v amdgpu.ids | tail -2
} > $O/seam_more.txt 2>&1
cat $O/seam_more.txt
python bench.py 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['cpu_baseline'], j['north_star_10k']['ms_per_step'], j['north_star_10k']['roofline_valu']['kernel_ms_per_step'])" > $O/bench_default.txt 2>&1
cat $O/bench_default.txt
