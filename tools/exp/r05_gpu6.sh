# one Philox block per (sample, pass, density) handed over through LDS in the latency geometries too (NBP_X_UUL_LAT), at launch bounds 512 / 256
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
for lib in default uul512 uul256; do
  if [ $lib != default ]; then export NBP_LIB_OVERRIDE=$PWD/tools/exp/libnbp_$lib.so; else unset NBP_LIB_OVERRIDE; fi
  for spec in "1 2" "1 3" "1 5" "1 8" "3 3" "6 4" "12 3" "20 3" "66 2" "66 3"; do set -- $spec
    python tools/exp/prod_batch.py $1 $2 2>/dev/null | sed "s/^/$lib  /"
  done
  for c in 2 3; do
    python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-10k 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib config $c', j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['posterior_max_mean_err'])"
  done
done > $O/uul.txt 2>&1
cat $O/uul.txt
