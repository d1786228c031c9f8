# the plan cache of the clique seam under concurrent callers and re-seeded queued walks (round 6)
R=${GRAFT_REPO_ROOT:-$PWD}
gcc -O2 -Wall -fopenmp -I $R/include $R/examples/solve_by_clique_calls.c -o /tmp/sbcc -L $R/incrementalinference.jl_amd/csrc -lnbp -lm || exit 1
export LD_LIBRARY_PATH=$R/incrementalinference.jl_amd/csrc:/opt/rocm/lib:$LD_LIBRARY_PATH
for c in 4 16; do NBP_PLAN_CACHE_STATS=1 GPU_MAX_HW_QUEUES=$c /tmp/sbcc 1000 200 100 $c 2>&1 | grep -v amdgpu.ids | cut -c1-420 | sort | uniq -c | tail -6; done
NBP_PLAN_CACHE_STATS=1 NBP_WALKS=6 NBP_WALK_SEEDS=1 /tmp/sbcc 1000 200 100 -2 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tail -6
