#!/bin/bash
# Whole-solve posterior sha of BASELINE configurations under the library's geometry switches: every alternative kernel path claims
# the shipped path's particles bit for bit -- here checked on full solves.   Usage (GPU box): tools/exp/whole_solve_sha_switches.sh
R=${GRAFT_REPO_ROOT:-$PWD}
sha() { env "$@" NBP_BENCH_SHA=1 python $R/bench.py --config $C --steps 2 --warmup 1 --no-cpu-baseline --no-10k --no-profile-pass 2>&1 >/dev/null | grep -o "sha=[0-9a-f]*"; }
for C in ${CONFIGS:-2 3 4 5}; do
  base=$(sha NBP_X=1)
  echo "config $C: shipped $base"
  for sw in NBP_FIT_F64=1 NBP_NO_SPECULATIVE_FITS=1 NBP_SPEC_DEPTH3=0 NBP_NO_XS_PRODUCTS=1 NBP_PROPOSAL_WAVE_MIN=1000000 NBP_NO_LAZY_BANDWIDTH=1 NBP_FUSED_MIN=256 NBP_PRODUCT_ALL_LEVELS_HL=2 NBP_PRODUCT_NCH=2; do
    s=$(sha $sw)
    echo "   $sw: $s $([ "$s" == "$base" ] && echo identical || echo DIFFERENT)"
  done
done
