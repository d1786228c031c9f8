#!/bin/bash
# single-clique calls from several callers on ONE context, merged by the library (nbp_host.cpp): walk time by callers and by the
# number of lanes (NBP_COMBINE_LANES: batches side by side on the device) and by the
# moment a leader gives the callers of the last round to come back (NBP_COMBINE_GATHER_US)
R=${GRAFT_REPO_ROOT:-$PWD}
gcc -O2 -Wall -fopenmp -I $R/include $R/examples/solve_by_clique_calls.c -o /tmp/sbcc -L $R/incrementalinference.jl_amd/csrc -lnbp -lm || exit 1
export LD_LIBRARY_PATH=$R/incrementalinference.jl_amd/csrc:/opt/rocm/lib:$LD_LIBRARY_PATH
for l in ${LANES:-1 2 4 8}; do
for g in ${GATHERS:-0 150}; do
  for c in ${CALLERS:-4 16 32 64}; do
    echo "lanes $l, gather $g us, $c callers:"
    NBP_COMBINE_LANES=$l GPU_MAX_HW_QUEUES=${HWQ:-8} NBP_COMBINE_GATHER_US=$g NBP_SHARED_CTX=1 NBP_PLAN_CACHE_STATS=1 /tmp/sbcc 1000 200 100 $c 2>&1 | grep -v amdgpu.ids | sed -e 's/.*one C call per clique, beliefs from and to host memory: /  /' | grep -v "^solve_by" | cut -c1-200
  done
done
done
