#!/bin/bash
# The same differential check over particle counts (the kernels' geometries -- rows of a fit, waves of a workgroup, helper lanes --
# follow N): a 400-variable Euclid(2) chain and a 300-pose circular chain at several N under the geometry switches.
R=${GRAFT_REPO_ROOT:-$PWD}
sha() { env "$@" NBP_BENCH_SHA=1 python $R/bench.py --config $C --nvars $NV --particles $N --steps 1 --warmup 0 --no-cpu-baseline --no-10k --no-profile-pass 2>&1 >/dev/null | grep -o "sha=[0-9a-f]*\|Error.*\|error.*" | head -1; }
for C in 2 3; do
  NV=$([ $C == 2 ] && echo 400 || echo 300)
  for N in 64 100 128 192 256 257 300 320 500; do
    base=$(sha NBP_X=1)
    line="config $C N=$N: shipped $base |"
    for sw in NBP_FIT_F64=1 NBP_NO_SPECULATIVE_FITS=1 NBP_NO_XS_PRODUCTS=1 NBP_PROPOSAL_WAVE_MIN=1000000 NBP_FUSED_MIN=64 NBP_PRODUCT_ALL_LEVELS_HL=2 NBP_PRODUCT_NCH=1 NBP_PRODUCT_HL2_MIN=100000; do
      s=$(sha $sw)
      line="$line $([ "$s" == "$base" ] && [ -n "$s" ] && echo ok || echo "$sw:DIFFERENT($s)")"
    done
    echo "$line"
  done
done
