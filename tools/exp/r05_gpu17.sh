R=${GRAFT_REPO_ROOT:-$PWD}
gcc -O2 -Wall -fopenmp -I $R/include $R/examples/solve_by_clique_calls.c -o /tmp/sbcc -L $R/incrementalinference.jl_amd/csrc -lnbp -lm || exit 1
export LD_LIBRARY_PATH=$R/incrementalinference.jl_amd/csrc:/opt/rocm/lib:$LD_LIBRARY_PATH
NBP_SEAM_TIMES=1 /tmp/sbcc 1000 200 100 -2 2>&1 | grep -v amdgpu.ids | cut -c1-600
NBP_SEAM_TIMES=1 /tmp/sbcc 1000 200 100 -1 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-600
