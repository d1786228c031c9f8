cd $GRAFT_REPO_ROOT
gcc -O2 -Wall -fopenmp -I include examples/solve_by_clique_calls.c -o /tmp/sbcc -L incrementalinference.jl_amd/csrc -lnbp -lm || exit 1
export LD_LIBRARY_PATH=$PWD/incrementalinference.jl_amd/csrc:/opt/rocm/lib:$LD_LIBRARY_PATH
/tmp/sbcc 1000 200 100 0 2>&1 | grep -v amdgpu.ids | tail -2
/tmp/sbcc 1000 200 100 -1 2>&1 | grep -v amdgpu.ids | tail -3
NBP_SEAM_TIMES=1 /tmp/sbcc 1000 200 100 -1 2>&1 | grep -v amdgpu.ids | tail -3
