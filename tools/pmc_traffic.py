"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum,TCC_MISS_sum; one pass each,
no tracing) of `bench.py --steps K`: per-kernel averages over the TIMED region (the last K x 60
dispatches of each kernel), the nbp_copy_kernel calibration of the counters' units (a copy launch
moves exactly blocks x slot_stride x 8 bytes each way in the same 8 B/lane access pattern the other
kernels use), and the corrected HBM bytes per launch.
Usage: pmc_traffic.py <dir with FETCH_SIZE/ WRITE_SIZE/ TCC/> <steps> <N> [out.json] [algorithmic bytes per step]"""
import collections
import csv
import glob
import gzip
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd import short_name  # noqa: E402


def load(d):
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from rocpd import counter_rows, short_name
    return counter_rows(d)


def per_kernel(rows, counter, steps, lps=60):
    by = collections.defaultdict(list)
    rows = [r for r in rows if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # the timed region starts at the `steps`-th last nbp_reseed_kernel dispatch (one per step)
    rs_ = [int(r["Dispatch_Id"]) for r in rows if r["Kernel_Name"].startswith("nbp_reseed_kernel")]
    d0 = rs_[-steps] if len(rs_) >= steps else 0
    for r in rows:
        if int(r["Dispatch_Id"]) < d0 and not r["Kernel_Name"].startswith("nbp_copy_kernel"):
            continue
        name = short_name(r["Kernel_Name"])
        if name.startswith("nbp_"):
            by[name].append((float(r["Counter_Value"]), int(r["Grid_Size"]) // int(r["Workgroup_Size"])))
    out = {}
    for k, v in by.items():
        out[k] = v
    return out


def main(root, steps, N, outp=None, alg=None):
    S = 3 * N + 8
    res = {"steps": steps, "N": N, "note": "values are per launch, timed region only"}
    cal = {}
    data = {}
    for cname, sub in (("FETCH_SIZE", "FETCH_SIZE"), ("WRITE_SIZE", "WRITE_SIZE"), ("TCC_HIT_sum", "TCC"), ("TCC_MISS_sum", "TCC")):
        data[cname] = per_kernel(load(f"{root}/{sub}"), cname, steps)
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        cp = data[cname].get("nbp_copy_kernel", [])
        big = [(v, b) for v, b in cp if b >= 32]  # launches large enough that descriptors/instructions do not matter
        if big:
            known = sum(b * S * 8 for _, b in big)
            raw = sum(v for v, _ in big)
            cal[cname] = known / raw  # bytes per counter unit
    res["calibration_bytes_per_unit"] = cal
    kern = {}
    for k in sorted(set(data["FETCH_SIZE"]) | set(data["WRITE_SIZE"])):
        e = {}
        for cname in ("FETCH_SIZE", "WRITE_SIZE"):
            v = data[cname].get(k, [])
            if v:
                e[cname + "_raw_avg"] = sum(x for x, _ in v) / len(v)
                e[cname + "_bytes_avg"] = e[cname + "_raw_avg"] * cal.get(cname, 1024.0)
                e["launches"] = len(v)
        h = data["TCC_HIT_sum"].get(k, [])
        m = data["TCC_MISS_sum"].get(k, [])
        if h and m:
            hs, ms = sum(x for x, _ in h), sum(x for x, _ in m)
            e["l2_hit_rate"] = hs / max(hs + ms, 1.0)
        if "FETCH_SIZE_bytes_avg" in e and "WRITE_SIZE_bytes_avg" in e:
            e["hbm_bytes_per_launch"] = e["FETCH_SIZE_bytes_avg"] + e["WRITE_SIZE_bytes_avg"]
        kern[k] = e
    res["kernels"] = kern
    # the whole solve: every launch of the timed region, per step (the copy kernel's launches outside the timed region
    # were only admitted for the calibration: count steps' worth of them by their share of reseed-delimited dispatches)
    per_step, total = {}, 0.0
    for k, e in kern.items():
        if "hbm_bytes_per_launch" not in e:
            continue
        n = e["launches"]
        if k == "nbp_copy_kernel":
            n = min(n, steps * max(1, round(e["launches"] / max(steps + 1, 1))))
        per_step[k] = e["hbm_bytes_per_launch"] * n / steps
        total += per_step[k]
    res["hbm_bytes_per_step_by_kernel"] = per_step
    res["hbm_bytes_per_step"] = total
    if alg:
        res["algorithmic_bytes_per_step"] = alg
        res["traffic_over_algorithmic"] = total / alg
    s = json.dumps(res, indent=1)
    print(s)
    if outp:
        open(outp, "w").write(s + "\n")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None,
         float(sys.argv[5]) if len(sys.argv) > 5 else None)
