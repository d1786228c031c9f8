cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_native_host.py tests/test_gpu_clique_entry.py tests/test_gpu_sharded_emulation.py -m gpu -x -q 2>&1 | tail -8
gcc -O2 -Wall -fopenmp -I include examples/solve_by_clique_calls.c -o /tmp/sbcc -L incrementalinference.jl_amd/csrc -lnbp -lm || exit 1
export LD_LIBRARY_PATH=$PWD/incrementalinference.jl_amd/csrc:/opt/rocm/lib:$LD_LIBRARY_PATH
/tmp/sbcc 1000 200 100 0 2>&1 | grep -v amdgpu.ids
/tmp/sbcc 1000 200 100 -1 2>&1 | grep -v amdgpu.ids
/tmp/sbcc 1000 200 100 -1 2>&1 | grep -v amdgpu.ids
