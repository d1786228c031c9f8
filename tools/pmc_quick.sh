#!/bin/bash
# HBM traffic of the config-2 solve in three PMC passes (FETCH_SIZE, WRITE_SIZE, L2 hit / miss; no tracing).
# Usage (on the GPU box): tools/pmc_quick.sh <tag> [ENV=VALUE ...]     -> gpurun_out/<tag>/pmc_traffic.json
TAG=$1; shift
[ -n "$TAG" ] || { echo "usage: tools/pmc_quick.sh <tag> [ENV=VALUE ...]"; exit 2; }
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-10k --no-profile-pass"
for p in FETCH_SIZE WRITE_SIZE; do env "$@" rocprofv3 --pmc $p -d $O/pmc/$p -- $B > $O/bench_$p.json 2> $O/pmc_$p.err; done
env "$@" rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc/TCC -- $B > /dev/null 2> $O/pmc_TCC.err
ALG=$(env "$@" python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-10k | python -c "import json,sys; print(json.load(sys.stdin)['roofline']['alg_bytes_per_step'])")
python $R/tools/pmc_traffic.py $O/pmc 5 200 $O/pmc_traffic.json $ALG > /dev/null 2> $O/pmc_traffic.err
rm -rf $O/pmc
python - <<EOF
import json, sys
sys.path.insert(0, "$R")
import bench
d = json.load(open("$O/pmc_traffic.json"))
d["kernel_sources_sha"] = bench.kernel_sources_sha()
json.dump(d, open("$O/pmc_traffic.json", "w"), indent=1)
print("$TAG", "$@", ": %.1f MB per solve = %.2f x algorithmic (%.1f MB)" % (d["hbm_bytes_per_step"] / 1e6, d.get("traffic_over_algorithmic", 0), d.get("algorithmic_bytes_per_step", 0) / 1e6))
for k, v in sorted(d["hbm_bytes_per_step_by_kernel"].items(), key=lambda kv: -kv[1]):
    print("   %-34s %7.1f MB  (%d launches, %.2f MB each)" % (k, v / 1e6, d["kernels"][k]["launches"], d["kernels"][k]["hbm_bytes_per_launch"] / 1e6))
EOF
