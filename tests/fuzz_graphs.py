"""Differential fuzz of WHOLE SOLVES (graph initialisation, up pass, down pass) against the oracle, bit for bit: random sparse
graphs on all five manifolds -- a spanning tree of relative factors with loop closures, extra priors, multihypo sightings,
mixtures, nullhypo, EuclidDistance ranges, partial priors, marginalized variables -- with measurement noise from 1e-2 to 1, odometry
steps from 1 to 1000 (priors that put a graph at 1e4: the badly scaled inputs the op fuzz found the tie order with), N = 64 / 100,
gibbsIters 1 .. 4, joint messages on every third graph.  The HIP backend under the native host's schedule, the oracle under the
Python mirror's; every variable's points and bandwidths compared with np.array_equal.
usage (GPU box): fuzz_graphs.py [seeds=40] [first seed=0] [seam | perturb | sharded]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif
from oracle.oracle_backend import OracleBackend


THREADS = int(os.environ.get("FUZZ_SEAM_THREADS", "0"))  # seam mode: the batched leg as concurrent single calls from this many host threads
SIZES = tuple(int(x) for x in os.environ.get("FUZZ_GRAPH_SIZES", "4,36").split(","))  # variables per graph: from, to (FUZZ_GRAPH_SIZES=120,400: levels that fill launches)


def random_graph(seed):
    r = np.random.default_rng(seed)
    kind = int(r.integers(0, 5))
    step = float(r.choice([1.0, 1.0, 30.0, 1000.0]))
    sig = float(r.choice([0.01, 0.1, 1.0]))
    far = float(r.choice([0.0, 0.0, 100.0, 1e4]))
    if kind == 0:
        vt = iif.ContinuousScalar
        rel = lambda: iif.LinearRelative(iif.Normal(step, sig))
        pri = lambda: iif.Prior(iif.Normal(far, 10 * sig))
    elif kind == 1:
        vt = iif.ContinuousEuclid(2)
        rel = lambda: iif.LinearRelative(iif.MvNormal([step, 0.1 * step], [sig, sig]))
        pri = lambda: iif.Prior(iif.MvNormal([far, -far], [10 * sig, 10 * sig]))
    elif kind == 2:
        vt = iif.ContinuousEuclid(3)
        rel = lambda: iif.LinearRelative(iif.MvNormal([step, 0.0, -0.5 * step], [sig, sig, sig]))
        pri = lambda: iif.Prior(iif.MvNormal([far, 0.0, -far], [10 * sig, 10 * sig, 10 * sig]))
    elif kind == 3:
        vt = iif.Circular
        rel = lambda: iif.CircularCircular(iif.Normal(0.3, min(sig, 0.3)))
        pri = lambda: iif.PriorCircular(iif.Normal(float(r.uniform(-3, 3)), 0.2))
    else:
        vt = iif.SpecialEuclidean2
        rel = lambda: iif.ManifoldFactor(iif.MvNormal([step, 0.1 * step, 0.2], [sig, sig, 0.1 * min(sig, 0.5)]))
        pri = lambda: iif.ManifoldPrior(np.array([far, -far, 0.3]), iif.MvNormal(np.zeros(3), [10 * sig, 10 * sig, 0.05]))
    n = int(r.integers(*SIZES))
    fg = iif.initfg(iif.SolverParams(N=int(r.choice([64, 100])), gibbsIters=int(r.integers(1, 5))))
    fg.solverParams.useMsgLikelihoods = seed % 3 == 1
    for i in range(n):
        iif.addVariable(fg, f"v{i}", vt)
    iif.addFactor(fg, ["v0"], pri())
    for i in range(1, n):
        j = int(r.integers(max(0, i - 6), i))
        nh = 0.1 if r.random() < 0.15 else 0.0
        if kind == 0 and r.random() < 0.15:
            iif.addFactor(fg, [f"v{j}", f"v{i}"], iif.Mixture(iif.LinearRelative, (iif.Normal(step, sig), iif.Normal(2 * step, 5 * sig)), [0.7, 0.3]))
        else:
            iif.addFactor(fg, [f"v{j}", f"v{i}"], rel(), nullhypo=nh)
    for _ in range(int(r.integers(0, n // 3 + 1))):
        a, b, c = (int(x) for x in r.choice(n, size=3, replace=False))
        u = r.random()
        if u < 0.35:
            iif.addFactor(fg, [f"v{a}", f"v{b}"], rel())
        elif u < 0.5:
            iif.addFactor(fg, [f"v{a}"], pri())
        elif u < 0.6 and kind in (1, 2):
            iif.addFactor(fg, [f"v{a}", f"v{b}"], iif.EuclidDistance(iif.Normal(abs(step) * abs(a - b) * 0.5 + 1.0, sig)))
        elif u < 0.7 and kind in (1, 2):
            iif.addFactor(fg, [f"v{a}"], iif.PartialPrior(vt, iif.Normal(far, 10 * sig), (int(r.integers(1, 3 if kind == 1 else 4)),)))
        elif kind in (0, 1, 3):
            iif.addFactor(fg, [f"v{a}", f"v{b}", f"v{c}"], rel(), multihypo=[1.0, 0.5, 0.5])
    if r.random() < 0.3:
        fg.getVariable(f"v{int(r.integers(0, n))}").ismargin = True
    return fg, dict(kind=kind, n=n, step=step, sig=sig, far=far, N=fg.solverParams.N, joint=fg.solverParams.useMsgLikelihoods)


def solve_pair(seed):
    (fa, info), (fb, _) = random_graph(seed), random_graph(seed)
    order = iif.nestedDissectionOrder(fa)
    try:
        iif.solveTree(fa, eliminationOrder=order, backend=lambda n, s, side_ints=0: OracleBackend(n, s, side_ints, threads=32), seed=seed)
    except ValueError as e:
        return info, None, f"not solved ({str(e)[:80]})"
    iif.solveTree(fb, eliminationOrder=order, backend=lambda n, s, side_ints=0: iif.HipBackend(n, s, side_ints=side_ints), seed=seed)
    differ = [v for v in fa.ls() if not (np.array_equal(fa.getVal(v), fb.getVal(v)) and np.array_equal(np.asarray(fa.getVariable(v).bw), np.asarray(fb.getVariable(v).bw)))]
    worst = max((float(np.abs(fa.getVal(v) - fb.getVal(v)).max()) for v in differ), default=0.0)
    finite = all(np.isfinite(fb.getVal(v)).all() for v in fb.ls())
    return info, (len(fa.ls()), differ, worst, finite), None


def seam_pair(seed):
    """the same random graph through the PER-CLIQUE entry points (tests/clique_csm.py: one nbp_clique_upsolve / _downsolve per
    clique, joint messages through nbp_clique_upsolve_joint; and the cliques of a level in one nbp_clique_solve_batch) against the
    whole-tree resident program, both on the device: the same bytes (an SE(2) belief crosses the host boundary between clique
    calls as a rotation matrix: 1e-9 there)"""
    from clique_csm import solve_tree_by_clique_calls, solve_tree_by_clique_calls_joint, solve_tree_by_level_batches
    hip = lambda n, s, side_ints=0: iif.HipBackend(n, s, side_ints=side_ints)
    (fa, info), (fb, _) = random_graph(seed), random_graph(seed)
    for f in (fa, fb):
        iif.initAll(f, backend=hip, seed=seed)
        f.solverParams.graphinit = False
    order = iif.nestedDissectionOrder(fa)
    tree = iif.buildTreeReset(fa, order)
    try:
        iif.solveTree(fa, tree=iif.buildTreeReset(fa, order), backend=hip, seed=seed + 7)
    except ValueError as e:
        return info, None, f"not solved ({str(e)[:80]})"
    be = hip(fb.solverParams.N, 2048)
    import clique_csm
    real_batch = clique_csm.clique_solve_batch
    try:
        if THREADS:
            # the cliques of a level as CONCURRENT single calls on the one context, from a pool of host threads (ctypes releases
            # the GIL): the library merges them (csrc/nbp_host.cpp clique_solve) -- in place of the batched call
            from concurrent.futures import ThreadPoolExecutor
            from iif_amd.native_host import clique_solve

            def threaded(backend, calls):
                with ThreadPoolExecutor(THREADS) as ex:
                    return list(ex.map(lambda c: clique_solve(backend, *c[0], **c[1]), calls))
            clique_csm.clique_solve_batch = threaded
        if info["joint"]:
            post, _ = solve_tree_by_clique_calls_joint(fb, tree, be, seed + 7)
            post2, _ = solve_tree_by_clique_calls_joint(fb, tree, be, seed + 7, batched=True)
        else:
            post, _ = solve_tree_by_clique_calls(fb, tree, be, seed + 7)
            post2, _ = solve_tree_by_level_batches(fb, tree, be, seed + 7)
    finally:
        clique_csm.clique_solve_batch = real_batch
        be.close()
    differ, worst = [], 0.0
    for v in fa.ls():
        for q in (post, post2):
            a, b = fa.getVal(v), q[v].pts
            if info["kind"] != 4:  # the same bytes
                d = np.abs(a - b).max() if a.shape == b.shape else np.inf
                db = np.abs(np.asarray(fa.getVariable(v).bw) - np.asarray(q[v].bw)).max()
                if d > 0 or db > 0:
                    differ.append(v); worst = max(worst, float(d), float(db))
            else:
                # SE(2): between clique calls a belief is the reference's host form, (t, R) -- theta -> (cos, sin) -> atan2 is not
                # a bitwise round trip, an ulp in a heading moves a 3-D search by 1e-4 and a Gibbs label with it: other draws of
                # the same posterior.  Held as a distribution: the means of x, y within 0.75 of the spread
                for k in range(2):
                    sd = 0.5 * (a[:, k].std() + b[:, k].std()) + 1e-9
                    d = abs(a[:, k].mean() - b[:, k].mean()) / sd
                    if d > 0.75:
                        differ.append(v); worst = max(worst, float(d))
    return info, (len(fa.ls()), sorted(set(differ)), worst, True), None


def perturb_pair(seed):
    """SE(2) graphs only: the whole-tree program twice on the device, the second time with the heading of every particle of ONE
    variable moved by one ulp before the solve -- how far apart two solves are that differ by what a belief's trip through its
    host form (t, R) does to it.  The yardstick for the SE(2) lines of the clique-seam fuzz: same criterion."""
    hip = lambda n, s, side_ints=0: iif.HipBackend(n, s, side_ints=side_ints)
    (fa, info), (fb, _) = random_graph(seed), random_graph(seed)
    if info["kind"] != 4:
        return info, None, "not an SE(2) graph"
    for f in (fa, fb):
        iif.initAll(f, backend=hip, seed=seed)
        f.solverParams.graphinit = False
    v = fb.getVariable("v1")
    th = np.arctan2(v.val[:, 3], v.val[:, 2])
    th2 = np.nextafter(th, np.inf)
    v.val[:, 2], v.val[:, 3], v.val[:, 4], v.val[:, 5] = np.cos(th2), np.sin(th2), -np.sin(th2), np.cos(th2)
    order = iif.nestedDissectionOrder(fa)
    for f in (fa, fb):
        iif.solveTree(f, tree=iif.buildTreeReset(f, order), backend=hip, seed=seed + 7)
    differ, worst = [], 0.0
    for v in fa.ls():
        a, b = fa.getVal(v), fb.getVal(v)
        for k in range(2):
            sd = 0.5 * (a[:, k].std() + b[:, k].std()) + 1e-9
            d = abs(a[:, k].mean() - b[:, k].mean()) / sd
            if d > 0.75:
                differ.append(v); worst = max(worst, float(d))
    return info, (len(fa.ls()), sorted(set(differ)), worst, True), None


def sharded_pair(seed):
    """the sharded solve of row (e): the cliques of the graph's tree partitioned over 2 .. 4 ranks (dist_solver.partition_cliques),
    every rank's program compiled with its ghosts and exchange points, the ranks emulated on ONE GPU (a context each, separator
    slots carried between the arenas at the exchange points as the RCCL exchange carries them) -- against the one-rank program:
    every frontal variable on its owner, the same bytes"""
    import collections
    from iif_amd.dist_solver import partition_cliques
    hip = lambda n, s, side_ints=0: iif.HipBackend(n, s, side_ints=side_ints)
    fg, info = random_graph(seed)
    iif.initAll(fg, backend=hip, seed=seed)
    tree = iif.buildTreeReset(fg, iif.nestedDissectionOrder(fg))
    world = 2 + seed % 3
    if len(tree.cliques) < world:
        return info, None, "fewer cliques than ranks"
    N, RAW = fg.solverParams.N, abi.EUCLID3

    def load(be, tp):
        for v in fg.ls():
            var = fg.getVariable(v)
            be.slot_write(tp.main[v], var.varType.manifold, var.val, var.bw)
        iif.solver.write_densities(fg, be)

    try:
        tp = iif.TreeProgram(fg, tree, seed=seed + 3)
    except ValueError as e:
        return info, None, f"not compiled ({str(e)[:80]})"
    be = hip(N, tp.n_slots)
    load(be, tp)
    prog = be.program(tp.stages, lazy_bandwidth=True)
    prog.run(); be.synchronize()
    ref = {v: be.slot_read(tp.main[v], RAW) for v in fg.ls()}
    prog.close(); be.close()
    owner = partition_cliques(tree, world)
    if len(set(owner.values())) < world:
        return info, None, "the partition left a rank empty"
    tps = [iif.TreeProgram(fg, tree, seed=seed + 3, owner=owner, rank=r) for r in range(world)]
    bes = [hip(N, t.n_slots) for t in tps]
    progs = []
    try:
        for b, t in zip(bes, tps):
            load(b, t)
            progs.append(b.program(t.stages, lazy_bandwidth=True))
        mail = collections.defaultdict(collections.deque)
        pos, sent = [0] * world, [False] * world
        for _ in range(100000):
            progress = False
            for r, t in enumerate(tps):
                if pos[r] >= len(t.segments):
                    continue
                seg = t.segments[pos[r]]
                if seg[0] == "run":
                    if seg[2] > seg[1]:
                        progs[r].run(seg[1], seg[2])
                    pos[r] += 1; progress = True
                    continue
                if not sent[r]:
                    bes[r].synchronize()
                    for peer, slot in seg[1]:
                        mail[(r, peer)].append(bes[r].slot_read(slot, RAW))
                    sent[r] = True; progress = True
                need = collections.Counter(peer for peer, _ in seg[2])
                if all(len(mail[(q, r)]) >= n for q, n in need.items()):
                    for peer, slot in seg[2]:
                        pts, bw = mail[(peer, r)].popleft()
                        bes[r].slot_write(slot, RAW, pts, bw)
                    pos[r] += 1; sent[r] = False; progress = True
            if all(pos[r] >= len(t.segments) for r, t in enumerate(tps)):
                break
            if not progress:
                return info, (len(fg.ls()), ["<the emulated ranks deadlocked>"], float("inf"), True), None
        differ, worst = [], 0.0
        for c, r in owner.items():
            for v in tree.cliques[c].frontalIDs:
                pts, bw = bes[r].slot_read(tps[r].main[v], RAW)
                if not (np.array_equal(pts, ref[v][0]) and np.array_equal(np.asarray(bw), np.asarray(ref[v][1]))):
                    differ.append(v); worst = max(worst, float(np.abs(pts - ref[v][0]).max()))
        info = dict(info, world=world)
        return info, (len(fg.ls()), differ, worst, True), None
    finally:
        for pr in progs:
            pr.close()
        for b in bes:
            b.close()


def main():
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    seam = len(sys.argv) > 3 and sys.argv[3] == "seam"  # third argument "seam": clique calls against the whole-tree program
    ok = bad = skipped = 0
    for seed in range(first, first + nseeds):
        mode = sys.argv[3] if len(sys.argv) > 3 else ""
        info, res, why = {"perturb": perturb_pair, "seam": seam_pair, "sharded": sharded_pair}.get(mode, solve_pair)(seed)
        tag = f"graph {seed} (manifold kind {info['kind']}, {info['n']} variables, N {info['N']}, step {info['step']:g}, noise {info['sig']:g}, prior at {info['far']:g}{', joint messages' if info['joint'] else ''})"
        if res is None:
            skipped += 1
            print(f"{tag}: {why}", flush=True)
            continue
        nv, differ, worst, finite = res
        if differ:
            bad += 1
            print(f"{tag}: {len(differ)} of {nv} variables DIFFER (by up to {worst:.3e}){'' if finite else ', non-finite values on the device'}: {differ[:6]}", flush=True)
        else:
            ok += 1
            print(f"{tag}: {nv} of {nv} variables " + ("agree (SE(2): as distributions)" if seam and info["kind"] == 4 else "bit-identical"), flush=True)
    what = {"seam": "walks by clique calls (single and batched) deliver the whole-tree program's posteriors",
            "perturb": "SE(2) solves within the criterion of their one-ulp twin",
            "sharded": "solves sharded over 2 .. 4 emulated ranks deliver the one-rank program's bytes"}.get(sys.argv[3] if len(sys.argv) > 3 else "", "whole solves bit-identical to the oracle's")
    print(f"fuzz_graphs: {ok} of {ok + bad} {what} ({bad} differ, {skipped} not solved)")


if __name__ == "__main__":
    main()
