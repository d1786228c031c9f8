#!/bin/bash
# where a walk through the clique seam spends its host time (NBP_SEAM_TIMES: a third walk with the library's phase clock)
R=${GRAFT_REPO_ROOT:-$PWD}
gcc -O2 -Wall -fopenmp -I $R/include $R/examples/solve_by_clique_calls.c -o /tmp/sbcc -L $R/incrementalinference.jl_amd/csrc -lnbp -lm || exit 1
export LD_LIBRARY_PATH=$R/incrementalinference.jl_amd/csrc:/opt/rocm/lib:$LD_LIBRARY_PATH
export NBP_SEAM_TIMES=1
/tmp/sbcc 1000 200 100 0 2>&1 | grep -v amdgpu.ids
/tmp/sbcc 1000 200 100 1 2>&1 | grep -v amdgpu.ids
# the queued walks (resident beliefs; -2: the requests of every level kept across walks), second walk + the phase clock of a third
/tmp/sbcc 1000 200 100 -1 2>&1 | grep -v amdgpu.ids
/tmp/sbcc 1000 200 100 -2 2>&1 | grep -v amdgpu.ids
