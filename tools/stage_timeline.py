"""The launches of ONE solve (the last step of bench.py's timed region) in order: kernel, workgroups, duration.
Usage: stage_timeline.py <rocprofv3 --kernel-trace output dir> [--sum]   (--sum: per (kernel, grid) totals only)"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd import kernel_rows, short_name  # noqa: E402


def main(path, summary):
    rows = kernel_rows(path)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("nbp_reseed_kernel")]
    region = rows[starts[-1]:] if starts else rows
    t0 = int(region[0]["Start_Timestamp"])
    tot = collections.OrderedDict()
    prev_end = t0
    for r in region:
        name = short_name(r["Kernel_Name"]).replace("nbp_", "").replace("_kernel", "")
        grid = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * max(1, int(r["Grid_Size_Y"]))
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if not summary:
            print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:6.1f}  {name:24s} wgs {grid:6d} x {int(r['Workgroup_Size_X']):4d}  {(e - s) / 1e3:8.1f} us")
        prev_end = e
        k = (name, grid)
        tot[k] = (tot.get(k, (0, 0))[0] + 1, tot.get(k, (0, 0))[1] + (e - s) / 1e3)
    print(f"one solve: {(prev_end - t0) / 1e3:.1f} us, {len(region)} launches")
    if summary:
        for (name, grid), (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
            print(f"{name:24s} wgs {grid:6d}  n {n:3d}  total {us:9.1f} us  avg {us / n:8.1f}")


if __name__ == "__main__":
    main(sys.argv[1], "--sum" in sys.argv)
