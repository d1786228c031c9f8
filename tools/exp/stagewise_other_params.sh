#!/bin/bash
# every stage of a tree program on the oracle's state, BIT FOR BIT (round 6), with solver parameters BASELINE does not use:
# four shapes x eleven settings (ONLY="useMsgLikelihoods=1": that setting alone; N=300: another particle count).
# usage (GPU box): tools/exp/stagewise_other_params.sh > gpurun_out/r06/stagewise_other_solver_parameters.txt
R=${GRAFT_REPO_ROOT:-$PWD}
N=${N:-200}
SETTINGS=("productNiter=2" "productNiter=3" "productNiter=8" "inflateCycles=1" "inflateCycles=5" "gibbsIters=1" "gibbsIters=5" "useMsgLikelihoods=1" "limitfixeddown=1" "spreadNH=1.0 inflation=2.0" "nullSurplusAdd=0.0")
[ -n "$ONLY" ] && SETTINGS=("$ONLY")
for sh in 2 3 4 5; do
  size=150; [ $sh = 4 ] && size=3
  for p in "${SETTINGS[@]}"; do
    timeout 600 python $R/tools/exp/stagewise_any_n.py $sh $N $size $p 2>&1 | grep "^shape" | cut -c1-330
  done
done
