"""Summarise a rocprofv3 --kernel-trace CSV for the TIMED region of bench.py: the last
`steps` steps, delimited by the nbp_reseed_kernel launch that starts every step (earlier dispatches belong to
graph initialisation and warm-up).  Usage: summarize_trace.py <kernel_trace.csv> <steps> [launches_per_step]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(path, steps, lps=60):
    from rocpd import kernel_rows, short_name
    rows = kernel_rows(path)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # every step of bench.py starts with one nbp_reseed_kernel launch: the timed region begins at the `steps`-th from last
    rs_ = [int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"].startswith("nbp_reseed_kernel")]
    t_begin = rs_[-steps] if len(rs_) >= steps else 0
    region = [r for r in rows if int(r["Start_Timestamp"]) >= t_begin]
    if region:
        span = (max(int(r["End_Timestamp"]) for r in region) - t_begin) / 1e6
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in region) / 1e6
        print(f"timed region (from the {steps}-th last reseed launch): {span / steps:.2f} ms per step wall, {busy / steps:.2f} ms per step inside kernels, {len(region) / steps:.0f} launches per step")
    rows = region
    by = collections.defaultdict(list)
    for r in rows:
        name = short_name(r["Kernel_Name"])
        if name.startswith("nbp_product_kernel"):
            name = "nbp_product_kernel(x16|l8|m4|t2)"  # one launch per stage, three geometries
        if name.startswith("nbp_proposal_kernel"):
            name = "nbp_proposal_kernel(generic|lin2|lin3)"  # one launch per stage
        if name.startswith("nbp_"):
            by[name].append(r)
    print(f"{'kernel':34s} {'launches':>8s} {'avg_us':>10s} {'total_ms':>9s} | avg_us by grid size (blocks): <=8, <=64, <=300, >300")
    out = {}
    for name, rs in sorted(by.items()):
        tail = rs
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tail]
        g = [int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) for r in tail]
        buckets = collections.defaultdict(list)
        for dur, grid in zip(d, g):
            b = 0 if grid <= 8 else 1 if grid <= 64 else 2 if grid <= 300 else 3
            buckets[b].append(dur)
        bs = " ".join(f"{(sum(buckets[b]) / len(buckets[b])) if buckets[b] else 0:9.1f}(n={len(buckets[b])}, {sum(buckets[b]) / 1e3 / steps:6.1f}ms/step)" for b in range(4))
        print(f"{name:34s} {len(tail):8d} {sum(d) / len(d):10.1f} {sum(d) / 1e3:9.2f} | {bs}")
        out[name] = (len(tail), sum(d) / len(d))
    return out


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 60)
