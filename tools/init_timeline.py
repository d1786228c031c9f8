"""The launches of graph initialisation (everything before the first nbp_reseed_kernel of bench.py) summarised by kernel
and grid size.  Usage: init_timeline.py <rocprofv3 --kernel-trace output dir> [--list N]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd import kernel_rows, short_name  # noqa: E402

rows = kernel_rows(sys.argv[1])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = next(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("nbp_reseed_kernel"))
init = [r for r in rows[:first] if short_name(r["Kernel_Name"]).startswith("nbp_")]
t0, t1 = int(init[0]["Start_Timestamp"]), int(init[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in init)
print(f"graph initialisation: {len(init)} launches, {(t1 - t0) / 1e6:.1f} ms wall on the device, {busy / 1e6:.1f} ms inside kernels")
by = collections.defaultdict(lambda: [0, 0.0])
for r in init:
    k = short_name(r["Kernel_Name"]).replace("nbp_", "").replace("_kernel", "")
    by[k][0] += 1
    by[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} n {n:6d}  total {us / 1e3:9.2f} ms  avg {us / n:8.1f} us")
if "--list" in sys.argv:
    n = int(sys.argv[sys.argv.index("--list") + 1])
    prev = t0
    for r in init[1000:1000 + n]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"   +gap {(s - prev) / 1e3:7.1f}  {short_name(r['Kernel_Name'])[:40]:40s} wgs {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']) * max(1, int(r['Grid_Size_Y'])):5d}  {(e - s) / 1e3:8.1f} us")
        prev = e
