mkdir -p gpurun_out/r05
cd /tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py > $R/gpurun_out/r05/bench_plain.json 2> $R/gpurun_out/r05/bench_plain.err
for c in 3 4 5; do python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r05/bench_config$c.json 2> $R/gpurun_out/r05/bench_config$c.err; done
python - <<PY
import json
d=json.load(open("$R/gpurun_out/r05/bench_plain.json"))
print(d["ms_per_step"], d["roofline"]["traffic_over_algorithmic"], d["roofline"]["within_2x_of_a_ceiling"], {k: round(v, 3) for k, v in d["roofline_valu"].items() if k.startswith("frac") and k != "frac_note"}, d["north_star_10k"]["ms_per_step"], {k: round(v, 3) for k, v in d["north_star_10k"]["roofline_valu"].items() if k.startswith("frac")})
for c in (3,4,5):
    e=json.load(open("$R/gpurun_out/r05/bench_config%d.json" % c)); print(c, e["ms_per_step"], {k: round(v, 3) for k, v in e["roofline_valu"].items() if k.startswith("frac") and k != "frac_note"})
PY
