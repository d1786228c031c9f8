#!/bin/bash
# cost of compiling the device library with -ffp-contract=off (one rounding per written operation everywhere; the
# multiply-adds that matter are explicit fma() calls): bench lines of configs 2 / 4 / 5 with both builds, same box
# usage (GPU box): tools/exp/contract_off_cost.sh  -> gpurun_out/r06/contract_off_cost.txt
out=gpurun_out/r06/contract_off_cost.txt
mkdir -p gpurun_out/r06
: > $out
for cfg in 2 4 5; do
  for lib in default nocontract; do
    if [ $lib = nocontract ]; then export NBP_LIB_OVERRIDE=$PWD/tools/exp/libnbp_nocontract.so; else unset NBP_LIB_OVERRIDE; fi
    line=$(timeout 900 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-10k 2>/dev/null | tail -1)
    echo "config $cfg lib $lib: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "kernels", {k: round(v, 2) for k, v in ((d.get("roofline") or {}).get("kernel_ms_per_step") or {}).items()})' 2>&1 | tail -1)" >> $out
  done
done
cat $out
