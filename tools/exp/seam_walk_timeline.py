#!/usr/bin/env python3
"""the device's side of a QUEUED walk through the clique seam (examples/solve_by_clique_calls.c <..> -2): from a rocprofv3 kernel trace,
the last walk's launches -- busy time, idle gaps (and where the large ones are), per-kernel totals -- beside the whole-tree program's.
usage: seam_walk_timeline.py <trace dir>"""
import collections, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rocpd import kernel_rows, short_name
rows = kernel_rows(sys.argv[1])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short_name(r["Kernel_Name"]).replace("nbp_", "").replace("_kernel", ""),
       int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * max(1, int(r["Grid_Size_Y"]))) for r in rows]
# walks are separated by long idle stretches (the host reads the posteriors, compares); take the last stretch of dense launches
cuts = [0] + [i for i in range(1, len(ev)) if ev[i][0] - ev[i - 1][1] > 3_000_000]  # > 3 ms idle
segs = [ev[a:b] for a, b in zip(cuts, cuts[1:] + [len(ev)]) if b - a > 200]
def describe(seg, name):
    t0, t1 = seg[0][0], seg[-1][1]
    busy = sum(e - s for s, e, _, _ in seg)
    gaps = [(seg[i][0] - seg[i - 1][1], i) for i in range(1, len(seg))]
    big = sorted([g for g in gaps if g[0] > 20_000], reverse=True)
    print(f"{name}: {len(seg)} launches, {(t1 - t0) / 1e6:.2f} ms from first launch to last end, busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms "
          f"({len(big)} gaps over 20 us: {sum(g[0] for g in big) / 1e6:.2f} ms; largest {[round(g[0] / 1e3) for g in big[:8]]} us)")
    tot = collections.Counter()
    for s, e, n, g in seg:
        tot[n.split('<')[0].split('_')[0] if not n.startswith('prep') else 'prep'] += e - s
    print("   by kernel family (ms):", {k: round(v / 1e6, 2) for k, v in tot.most_common(8)})
    byname = collections.defaultdict(lambda: [0, 0])
    for s, e, n, g in seg:
        byname[n][0] += 1
        byname[n][1] += e - s
    for n, (c, t) in sorted(byname.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"      {n[:60]:60s} {c:6d} launches {t / 1e6:8.2f} ms  avg {t / c / 1e3:7.1f} us")
for i, seg in enumerate(segs[-3:]):
    describe(seg, f"stretch {len(segs) - 3 + i + 1} of {len(segs)}")
