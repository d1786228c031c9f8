#!/bin/bash
# The record of the single-precision bracketing of the bandwidth searches (profiles/rNN_fit_bracketing.txt): bit-equal bandwidths
# and timings with / without (NBP_FIT_F64=1), a soak over extreme clouds, every evaluation against the host's and the bound, what a
# pair costs in the candidate loop forms, small launches, the configurations either way.   Usage (GPU box): tools/exp/fit_bracketing_record.sh > out.txt
R=${GRAFT_REPO_ROOT:-$PWD}
  echo "== tools/exp/lcv_f32_check.py: all-double (NBP_FIT_F64=1) against the shipped search, per launch of fits (evals are per KDE: the sum over its coordinates)"
  python $R/tools/exp/lcv_f32_check.py 2>/dev/null | grep -v "fits=    1 "
  echo; echo "== tools/exp/fit_soak.py 60: random counts, slot sizes, manifolds, scales 1e-6 .. 1e6, offsets to 1e6, duplicates, heavy tails; all-double = bracketed = default (speculative) search, bit for bit"
  python $R/tools/exp/fit_soak.py 60 2>/dev/null | tail -n 13
  echo; echo "== tools/exp/lone_fit_latency.py: small launches (default = speculative search where every workgroup is resident)"
  python $R/tools/exp/lone_fit_latency.py 2>/dev/null
  for t in lcv_f32_values lcv_f32_proto; do
    [ -x $R/tools/exp/$t ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -Wno-unused-value -w $R/tools/exp/$t.hip -o $R/tools/exp/$t 2>/dev/null
  done
  echo; echo "== tools/exp/lcv_f32_values.hip: single-precision evaluations against the host's, P = 1, 2, 4 rows, nine point counts, two coordinate kinds, five bandwidths"
  $R/tools/exp/lcv_f32_values | awk '{ n++; if ($0 ~ /ok $/) ok++; for (k = 1; k < NF; k++) if ($k == "diff") { d = $(k + 1) + 0; if (d < 0) d = -d; if (d > m) m = d } } END { printf "%d evaluations, %d inside the bound, largest |f32 - f64| %.2e\n", n, ok, m }'
  $R/tools/exp/lcv_f32_values | grep "P=1 circ=0" | grep "h= 0.30"
  echo; echo "== tools/exp/lcv_f32_proto.hip: the pair loop alone, 8192 fits x 16 evaluations"
  $R/tools/exp/lcv_f32_proto 8192 16
  echo; echo "== the configurations, ms per solve as shipped | with NBP_FIT_F64=1 (same process order, same box)"
  for c in 2 3 4 5; do
    a=$(python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-10k --no-profile-pass 2>/dev/null | python -c "import json,sys; print('%.2f' % json.load(sys.stdin)['ms_per_step'])")
    b=$(NBP_FIT_F64=1 python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-10k --no-profile-pass 2>/dev/null | python -c "import json,sys; print('%.2f' % json.load(sys.stdin)['ms_per_step'])")
    echo "config $c: $a | $b"
  done
  echo; echo "== tools/exp/bracketing_whole_solve_sha.sh: every posterior of a full solve, bracketed fits against all-double"
  bash $R/tools/exp/bracketing_whole_solve_sha.sh 2>/dev/null
