"""per-kernel totals of a rocprofv3 --kernel-trace run (rocpd database) as the CSV `--stats` used to write:
Name,Calls,TotalDurationNs(us),AverageNs(us),Percentage.  Usage: kernel_stats.py <dir or .db> > out.csv"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd import kernel_rows

rows = kernel_rows(sys.argv[1])
tot = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    t = tot[r["Kernel_Name"]]
    t[0] += 1
    t[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
s = sum(t[1] for t in tot.values()) or 1.0
print("Name,Calls,TotalDurationUs,AverageUs,Percentage")
for k, (n, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"\"{k}\",{n},{d:.3f},{d / n:.3f},{100 * d / s:.3f}")
