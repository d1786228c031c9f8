#!/bin/bash
# A/B of two builds of libnbp on one box: bench lines of the given configs with the shipped library and with an alternative one
# usage (GPU box): tools/exp/ab_bench.sh <alt .so> <out file> [configs: default "2 3 4 5"]   (NBP_AB_10K=1: config 2 with the 10 000-variable leg)
alt=$1; out=$2; shift 2; cfgs=${@:-2 3 4 5}
mkdir -p $(dirname $out); : > $out
for cfg in $cfgs; do
  for lib in shipped alt shipped alt; do
    if [ $lib = alt ]; then export NBP_LIB_OVERRIDE=$PWD/$alt; else unset NBP_LIB_OVERRIDE; fi
    extra="--no-10k"; [ "$cfg" = 2 ] && [ -n "$NBP_AB_10K" ] && extra=""
    line=$(timeout 900 python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline $extra 2>/dev/null | tail -1)
    echo "config $cfg $lib ($([ $lib = alt ] && echo $alt || echo csrc/libnbp.so)): $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k=(d.get("roofline") or {}).get("kernel_ms_per_step") or {}; ns=d.get("north_star_10k") or {}; print("ms_per_step %.3f" % d["ms_per_step"], {a: round(v, 2) for a, v in k.items() if v > 0}, ("10k: %.2f ms" % ns["ms_per_step"]) if ns else "")' 2>&1 | tail -1)" >> $out
  done
done
unset NBP_LIB_OVERRIDE
cat $out
