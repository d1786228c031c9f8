#!/bin/bash
# soak: the merged single-clique calls walked many times (16 and 64 callers), every walk's posteriors must be the whole-tree program's bytes
R=${GRAFT_REPO_ROOT:-$PWD}
gcc -O2 -Wall -fopenmp -I $R/include $R/examples/solve_by_clique_calls.c -o /tmp/sbcc -L $R/incrementalinference.jl_amd/csrc -lnbp -lm || exit 1
export LD_LIBRARY_PATH=$R/incrementalinference.jl_amd/csrc:/opt/rocm/lib:$LD_LIBRARY_PATH
bad=0
for i in $(seq 1 ${REPS:-10}); do
  for c in 16 64 5; do
    out=$(NBP_SHARED_CTX=1 NBP_WALKS=6 GPU_MAX_HW_QUEUES=8 /tmp/sbcc 1000 200 100 $c 2>&1)
    echo "$out" | grep -q "1000 of 1000 posteriors byte-identical" || { bad=$((bad+1)); echo "run $i callers $c: $(echo "$out" | grep -v amdgpu | head -3 | cut -c1-300)"; }
  done
done
echo "soak: $bad bad runs of $((3 * ${REPS:-10}))"
