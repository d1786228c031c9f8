#!/bin/bash
# kernel trace of the queued walk (requests kept, plan cache on), four walks: the device's busy and idle time in the last one
R=${GRAFT_REPO_ROOT:-$PWD}
gcc -O2 -Wall -fopenmp -I $R/include $R/examples/solve_by_clique_calls.c -o /tmp/sbcc -L $R/incrementalinference.jl_amd/csrc -lnbp -lm || exit 1
export LD_LIBRARY_PATH=$R/incrementalinference.jl_amd/csrc:/opt/rocm/lib:$LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/seamtrace
NBP_WALKS=5 rocprofv3 --kernel-trace -d /tmp/seamtrace -- /tmp/sbcc 1000 200 100 -2 2>&1 | grep -v amdgpu.ids | grep "walks in order\|queued walk"
python $R/tools/exp/seam_walk_timeline.py /tmp/seamtrace
