#!/bin/bash
# the per-clique seam at the config-2 shape, plain C against libnbp.so (examples/solve_by_clique_calls.c): resident program,
# one call per clique with 1 / 4 / 16 concurrent callers, one batched call per tree level
R=${GRAFT_REPO_ROOT:-$PWD}
gcc -O2 -Wall -fopenmp -I $R/include $R/examples/solve_by_clique_calls.c -o /tmp/sbcc -L $R/incrementalinference.jl_amd/csrc -lnbp -lm || exit 1
export LD_LIBRARY_PATH=$R/incrementalinference.jl_amd/csrc:/opt/rocm/lib:$LD_LIBRARY_PATH
echo "examples/solve_by_clique_calls.c 1000 200 100 <callers> on one MI355X (config-2 shape: 1000-variable Euclid(2) chain, N = 200)"
for c in 1 4 16; do GPU_MAX_HW_QUEUES=$c /tmp/sbcc 1000 200 100 $c 2>&1 | grep -v amdgpu.ids; done
# the same callers on ONE context: the library merges the calls that arrive while a batch is on the device
for c in 4 16 64; do NBP_SHARED_CTX=1 NBP_PLAN_CACHE_STATS=1 GPU_MAX_HW_QUEUES=8 /tmp/sbcc 1000 200 100 $c 2>&1 | grep -v amdgpu.ids; done
/tmp/sbcc 1000 200 100 0 2>&1 | grep -v amdgpu.ids
/tmp/sbcc 1000 200 100 -1 2>&1 | grep -v amdgpu.ids
/tmp/sbcc 1000 200 100 -2 2>&1 | grep -v amdgpu.ids
