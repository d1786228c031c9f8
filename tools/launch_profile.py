"""Per-launch durations of one step from a rocprofv3 --kernel-trace CSV, sorted by grid size.
Usage: launch_profile.py <kernel_trace.csv> [launches_per_step=60]"""
import csv
import sys


def main(path, lps=60):
    rows = list(csv.DictReader(open(path)))
    for kn in ("nbp_proposal_kernel", "nbp_prep_kernel", "nbp_product_kernel"):
        rs = [r for r in rows if r["Kernel_Name"].startswith(kn)][-lps:]
        out = []
        for r in rs:
            g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) * max(1, int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))
            out.append((g, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        out.sort()
        print(kn, "total %.1f ms over %d launches" % (sum(d for _, d in out) / 1e3, len(out)))
        print("   blocks:us  " + " ".join("%d:%.0f" % x for x in out[::2]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
