#!/bin/bash
# copy_profiles.sh TAG: what tools/collect_profiles.sh TAG left under gpurun_out/TAG -> profiles/TAG_* (the tracked, judged copies)
TAG=${1:-r05}
cd "$(dirname "$0")/.." || exit 1
for f in gpurun_out/$TAG/*.json gpurun_out/$TAG/*.csv gpurun_out/$TAG/*.txt; do
  b=$(basename $f)
  case $b in pmc_traffic.txt|pmc_traffic_fused.txt) continue;; esac
  [ -s $f ] && cp $f profiles/${TAG}_$b
done
cp gpurun_out/$TAG/pmc_default/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
cp gpurun_out/$TAG/pmc_fused/pmc_traffic.json profiles/${TAG}_pmc_traffic_fused.json
python - <<PY
import json
def block(txt, js, head):
    a = [l for l in open(txt).read().strip().split("\n") if "amdgpu.ids" not in l]
    sha = json.load(open(js))["kernel_sources_sha"]
    return head + a[0].split(" : ")[1] + "   [kernel sources %s]\n" % sha + "\n".join(a[1:]) + "\n"
out = block("gpurun_out/$TAG/pmc_traffic.txt", "profiles/${TAG}_pmc_traffic.json", "default launch form: ") + "\n" + \
      block("gpurun_out/$TAG/pmc_traffic_fused.txt", "profiles/${TAG}_pmc_traffic_fused.json", "fused rounds (NBP_FUSED_MIN=256): ")
open("profiles/${TAG}_pmc_traffic_summary.txt", "w").write(out)
print(out.split("\n")[0])
PY
