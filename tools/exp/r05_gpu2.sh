# round 5: Optim's arithmetic in the Euclid(3) searches -- what it buys in the stage-wise comparison and what it costs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in "" tools/libnbp_nm0.so; do
  echo "=== library: ${lib:-default (NBP_NM_OPTIM_E3=1)}"
  NBP_LIB_OVERRIDE=$lib timeout 1500 python -m pytest tests/test_gpu_stagewise_parity.py -m gpu -x -q -s -k "config5 or config3_full or config4_full" --durations=8 2>&1 | grep -E "every stage|passed|failed|Error|assert|s call" | cut -c1-600
  for c in 5 3; do
  NBP_LIB_OVERRIDE=$lib python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-10k > gpurun_out/nm_$c.json 2> gpurun_out/nm_$c.err
  python - <<PY
import json
d=json.load(open("gpurun_out/nm_$c.json"))
k=d["roofline"]["kernel_ms_per_step"]
print("config $c: %.2f ms/step  proposal %.2f prep %.2f product %.2f  graph_init %.3f s" % (d["ms_per_step"],k["nbp_proposal_kernel"],k["nbp_prep_kernel"],k["nbp_product_kernel"],d["host_setup"]["graph_init_s"]))
PY
  done
done
