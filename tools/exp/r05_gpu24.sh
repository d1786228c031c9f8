R=${GRAFT_REPO_ROOT:-$PWD}
gcc -O2 -Wall -fopenmp -I $R/include $R/examples/solve_by_clique_calls.c -o /tmp/sbcc -L $R/incrementalinference.jl_amd/csrc -lnbp -lm || exit 1
export LD_LIBRARY_PATH=$R/incrementalinference.jl_amd/csrc:/opt/rocm/lib:$LD_LIBRARY_PATH
for N in 64 100 257 300 320 500; do
  for mode in 1 4 0 -1 -2; do
    out=$(/tmp/sbcc 150 $N 40 $mode 2>&1 | grep -v amdgpu.ids | head -1 | grep -o "[0-9]* of [0-9]* posteriors byte-identical")
    echo "N=$N callers=$mode: $out (exit $?)"
  done
done
