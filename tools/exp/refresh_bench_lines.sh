#!/bin/bash
# the bench lines of the round again (no traces, no counters): after host-side changes that leave the kernel sources -- and so the
# PMC stamp -- as they were.  usage (GPU box): tools/exp/refresh_bench_lines.sh r06
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err
python $R/bench.py --fused-min 256 --no-cpu-baseline --no-10k > $O/bench_fused.json 2> $O/bench_fused.err
for c in 3 4 5; do python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_config$c.json 2> $O/bench_config$c.err; done
python - <<PY
import json
for f in ("bench_plain", "bench_fused", "bench_config3", "bench_config4", "bench_config5"):
    try:
        d = json.loads(open("$O/" + f + ".json").read().strip().splitlines()[-1])
        r = d.get("roofline", {})
        print(f, d["value"], d["ms_per_step"], "traffic", r.get("traffic"), "stale", r.get("traffic_source_is_stale"))
    except Exception as e:
        print(f, "FAILED", e)
PY
