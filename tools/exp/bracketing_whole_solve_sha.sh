#!/bin/bash
# The posteriors of whole solves with the bracketed fits and with every evaluation in double precision (NBP_FIT_F64=1): the
# sha of every posterior of BASELINE's configurations must be the same (the bandwidths are bit-identical, so is everything
# computed from them).  Usage (GPU box): tools/exp/bracketing_whole_solve_sha.sh
R=${GRAFT_REPO_ROOT:-$PWD}
for c in 2 2p 3 4 5; do
  a=$(NBP_BENCH_SHA=1 python $R/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-10k --no-profile-pass 2>&1 >/dev/null | grep -o "sha=[0-9a-f]*")
  b=$(NBP_BENCH_SHA=1 NBP_FIT_F64=1 python $R/bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-10k --no-profile-pass 2>&1 >/dev/null | grep -o "sha=[0-9a-f]*")
  echo "config $c: bracketed $a | all-double $b | $([ "$a" == "$b" ] && [ -n "$a" ] && echo identical || echo DIFFERENT)"
done
