cd /tmp && export TMPDIR=/tmp
for n in 256 300; do
  rm -rf /tmp/kt_$n; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$n -- python $GRAFT_REPO_ROOT/tools/lcv_bench.py $n 4096 > /dev/null 2>&1
  f=$(find /tmp/kt_$n -name "*kernel_trace.csv" | head -1)
  python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f")) if "bandwidth" in r["Kernel_Name"]]
r=rows[-1]
print("$n", {k:r[k] for k in r if any(t in k for t in ("LDS","Scratch","VGPR","SGPR","Workgroup","Grid"))})
PY
done
