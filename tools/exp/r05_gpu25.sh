R=${GRAFT_REPO_ROOT:-$PWD}
for c in 2 3 4 5; do
  nv=$([ $c == 4 ] && echo 6 || echo 300)
  a=$(NBP_BENCH_SHA=1 python $R/bench.py --config $c --nvars $nv --steps 1 --warmup 0 --no-cpu-baseline --no-10k --no-profile-pass 2>&1 >/dev/null | grep -o "sha=[0-9a-f]*")
  b=$(NBP_BENCH_SHA=1 python $R/bench.py --config $c --nvars $nv --steps 1 --warmup 0 --no-cpu-baseline --no-10k --no-profile-pass --python-host 2>&1 >/dev/null | grep -o "sha=[0-9a-f]*")
  echo "config $c (size $nv): native host $a | python host $b | $([ "$a" == "$b" ] && [ -n "$a" ] && echo identical || echo DIFFERENT)"
done
