#!/bin/bash
# ab_bench.sh TAG [ENV=VAL ...]: config-2 solve time and kernel split under the given environment
TAG=$1; shift
env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-10k > gpurun_out/ab_$TAG.json 2> gpurun_out/ab_$TAG.err
python - <<PY
import json
d=json.load(open("gpurun_out/ab_$TAG.json"))
k=d["roofline"]["kernel_ms_per_step"]
print("$TAG: %.2f ms/step  proposal %.2f prep %.2f product %.2f  every-fit %.2f" % (d["ms_per_step"],k["nbp_proposal_kernel"],k["nbp_prep_kernel"],k["nbp_product_kernel"],d.get("ms_per_step_every_fit",0)))
PY
