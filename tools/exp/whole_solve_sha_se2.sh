#!/bin/bash
# the differential check on SE(2) (config 4's shape, a small lattice) and on the Euclid(3) mixture chain (config 5's shape)
R=${GRAFT_REPO_ROOT:-$PWD}
sha() { env "$@" NBP_BENCH_SHA=1 python $R/bench.py --config $C --nvars $NV --particles $N --steps 1 --warmup 0 --no-cpu-baseline --no-10k --no-profile-pass 2>&1 >/dev/null | grep -o "sha=[0-9a-f]*\|Error.*" | head -1; }
for C in 4 5; do
  NV=$([ $C == 4 ] && echo 8 || echo 300)
  for N in 100 200 257 300; do
    base=$(sha NBP_X=1)
    line="config $C (size $NV) N=$N: shipped $base |"
    for sw in NBP_FIT_F64=1 NBP_NO_SPECULATIVE_FITS=1 NBP_NO_XS_PRODUCTS=1 NBP_PROPOSAL_WAVE_MIN=1000000 NBP_FUSED_MIN=64 NBP_PRODUCT_ALL_LEVELS_HL=2 NBP_PRODUCT_NCH=1 NBP_PRODUCT_HL2_MIN=100000; do
      s=$(sha $sw)
      line="$line $([ "$s" == "$base" ] && [ -n "$s" ] && echo ok || echo "$sw:DIFFERENT($s)")"
    done
    echo "$line"
  done
done
