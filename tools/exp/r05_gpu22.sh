R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
cd /tmp
python $R/bench.py > $R/gpurun_out/r05/bench_plain.json 2> $R/gpurun_out/r05/bench_plain.err
python -c "
import json; d=json.load(open('$R/gpurun_out/r05/bench_plain.json')); print(d['ms_per_step'], d['roofline']['traffic_over_algorithmic'], d['roofline']['traffic_source_is_stale'], d['north_star_10k']['ms_per_step'], d['vs_cpu_baseline'])"
