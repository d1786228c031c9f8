#!/bin/bash
# lcv_ab.sh LIB...: the bandwidth fit alone (tools/lcv_bench.py, N = 200 / 256 / 100) and a config-2 solve under each library
# (paths relative to the repository root; "default" = csrc/libnbp.so).  Output: gpurun_out/lcv_ab_<name>.txt
mkdir -p gpurun_out
for L in "$@"; do
  name=$(basename "$L" .so)
  if [ "$L" = default ]; then unset NBP_LIB_OVERRIDE; else export NBP_LIB_OVERRIDE=$PWD/$L; fi
  {
    for geo in "200 8192" "200 2048" "200 64" "200 1" "256 8192" "100 8192" "128 8192"; do python tools/lcv_bench.py $geo; done
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-10k 2>/dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.readline()); k=d["roofline"]["kernel_ms_per_step"]
print("config 2: %.2f ms/step  proposal %.2f prep %.2f product %.2f  every-fit %.2f  valu %.3f" % (d["ms_per_step"],k["nbp_proposal_kernel"],k["nbp_prep_kernel"],k["nbp_product_kernel"],d.get("ms_per_step_every_fit",0),d["roofline_valu"]["frac"]))'
  } > gpurun_out/lcv_ab_$name.txt 2>&1
  echo "== $name"; cat gpurun_out/lcv_ab_$name.txt
done
