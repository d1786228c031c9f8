#!/bin/bash
# lcv_ko.sh LIB...: the chip-filling fit (N = 200, 8192 fits) under knock-out builds of the pair loop (-DNBP_LCV_KO=bits)
mkdir -p gpurun_out
for L in "$@"; do
  export NBP_LIB_OVERRIDE=$PWD/$L
  echo "== $(basename $L .so): $(python tools/lcv_bench.py 200 8192 2>/dev/null | tail -n 1)"
done | tee gpurun_out/lcv_ko.txt
