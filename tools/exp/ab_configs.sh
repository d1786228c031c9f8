#!/bin/bash
# ab_configs.sh TAG: solve time and kernel split of configs 4 and 5 (full size) and 2
TAG=$1
for c in 2 4 5; do
  python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-10k > gpurun_out/abc_${TAG}_$c.json 2> gpurun_out/abc_${TAG}_$c.err
  python - <<PY
import json
d=json.load(open("gpurun_out/abc_${TAG}_$c.json"))
k=d["roofline"]["kernel_ms_per_step"]
print("$TAG config $c: %.2f ms/step  proposal %.2f prep %.2f product %.2f  graph_init %.3f s" % (d["ms_per_step"],k["nbp_proposal_kernel"],k["nbp_prep_kernel"],k["nbp_product_kernel"],d["host_setup"]["graph_init_s"]))
PY
done
