R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
bash $R/tools/exp/whole_solve_sha_se2.sh > $R/gpurun_out/r05/whole_solve_sha_se2.txt 2>&1
cd $R
tools/pmc_quick.sh r05/pmc_default NBP_X=1 > $R/gpurun_out/r05/pmc_traffic.txt 2>&1
tools/pmc_quick.sh r05/pmc_fused NBP_FUSED_MIN=256 > $R/gpurun_out/r05/pmc_traffic_fused.txt 2>&1
cd /tmp
python $R/bench.py > $R/gpurun_out/r05/bench_plain.json 2> $R/gpurun_out/r05/bench_plain.err
head -2 $R/gpurun_out/r05/pmc_traffic.txt | tail -1; cat $R/gpurun_out/r05/whole_solve_sha_se2.txt
python -c "
import json; d=json.load(open('$R/gpurun_out/r05/bench_plain.json')); print(d['ms_per_step'], d['roofline']['traffic_over_algorithmic'], d['roofline']['traffic_source_is_stale'], d['north_star_10k']['ms_per_step'])"
