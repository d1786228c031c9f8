"""Per-kernel sums of SQ counters from rocprofv3 --pmc passes (rocpd databases under <dir>/*): which pipe a kernel
keeps busy.  Usage: pmc_sq.py <dir> [kernel-name-prefix]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd import counter_rows, short_name


def main(root, prefix="nbp_"):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(lambda: collections.defaultdict(int))
    for sub in sorted(os.listdir(root)):
        d = os.path.join(root, sub)
        if not os.path.isdir(d):
            continue
        for r in counter_rows(d):
            k = short_name(r["Kernel_Name"])
            if k.startswith(prefix):
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                n[k][r["Counter_Name"]] += 1
    for k in sorted(tot):
        print(k)
        for c in sorted(tot[k]):
            print(f"  {c:28s} {tot[k][c]:16.0f}  ({n[k][c]} dispatches)")
        t = tot[k]
        if t.get("SQ_WAVE_CYCLES"):
            w = t["SQ_WAVE_CYCLES"]
            for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c in t:
                    print(f"  {c + ' / WAVE_CYCLES':40s} {t[c] / w:.3f}")
        if t.get("SQ_BUSY_CYCLES") and t.get("SQ_ACTIVE_INST_VALU"):
            print(f"  note: ACTIVE_INST_* and WAVE_CYCLES count quad-cycles summed over waves")


if __name__ == "__main__":
    main(*sys.argv[1:])
