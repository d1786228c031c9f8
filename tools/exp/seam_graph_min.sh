#!/bin/bash
# one call per clique: the cached clique programs replayed plainly (default) against replayed as hipGraphs (NBP_PLAN_CACHE_GRAPH_MIN=1)
R=${GRAFT_REPO_ROOT:-$PWD}
gcc -O2 -Wall -fopenmp -I $R/include $R/examples/solve_by_clique_calls.c -o /tmp/sbcc -L $R/incrementalinference.jl_amd/csrc -lnbp -lm || exit 1
export LD_LIBRARY_PATH=$R/incrementalinference.jl_amd/csrc:/opt/rocm/lib:$LD_LIBRARY_PATH
for gm in 8 1; do
  for c in 1 4 16; do
    echo "NBP_PLAN_CACHE_GRAPH_MIN=$gm, $c caller(s): $(NBP_WALKS=4 NBP_PLAN_CACHE_GRAPH_MIN=$gm GPU_MAX_HW_QUEUES=$c /tmp/sbcc 1000 200 100 $c 2>&1 | grep -v amdgpu.ids | grep "byte-identical\|walks in order" | sed 's/.*cliques: //; s/ (means.*//' | tr '\n' ' ')"
  done
done
