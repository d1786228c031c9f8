"""-m gpu: a whole-tree solve compiled by the native host (include/nbp_host.h) gives bitwise the same
posteriors as the program the Python mirror compiles (same descriptors, same kernels)."""
import numpy as np
import pytest

from parity_utils import iif

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["chain", "lattice", "doors"])
def test_native_compiled_solve_equals_python_compiled_solve(hip_backend, name):
    from iif_amd import native_host
    fg = {"chain": lambda: iif.generateChainEuclid(40, vardims=2, priorEvery=10, N=100),
          "lattice": lambda: iif.generateSE2Lattice(rows=3, cols=5, N=100, closeEvery=2),
          "doors": lambda: iif.generateCircularDoors(nposes=40, N=100, sightEvery=5)}[name]()
    iif.initAll(fg, backend=hip_backend, seed=0)
    order = iif.nestedDissectionOrder(fg)

    def solve(native):
        if native:
            g = native_host.NativeGraph.from_fg(fg)
            assert g.order_nested_dissection() == order
            nt = g.build_tree(order)
            be = hip_backend(100, nt.plan_slots(False))
            main, prog = nt.main, nt.compile(be, 77)
            nmsg = nt.stats()["messages"]
        else:
            tp = iif.TreeProgram(fg, iif.buildTreeReset(fg, order), seed=77)
            be = hip_backend(100, tp.n_slots)
            main, prog, nmsg = tp.main, be.program(tp.stages), tp.n_messages
        for v in fg.ls():
            var = fg.getVariable(v)
            be.slot_write(main[v], var.varType.manifold, var.val, var.bw)
        prog.run()
        prog.reseed(5)
        prog.run()
        be.synchronize()
        out = {v: be.slot_read(main[v], fg.getVariable(v).varType.manifold) for v in fg.ls()}
        prog.close()
        be.close()
        return out, nmsg

    (a, ma), (b, mb) = solve(True), solve(False)
    assert ma == mb
    for v in fg.ls():
        np.testing.assert_array_equal(a[v][0], b[v][0])
        np.testing.assert_array_equal(a[v][1], b[v][1])


def test_pure_c_example_solves_a_chain(tmp_path):
    """examples/solve_chain.c: the whole path through include/nbp.h + include/nbp_host.h from plain C
    (what a non-Python host such as the Julia shim sees)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "incrementalinference.jl_amd", "csrc")
    exe = str(tmp_path / "solve_chain")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "solve_chain.c"), "-o", exe,
                           "-L", lib, "-lnbp", f"-Wl,-rpath,{lib}", "-lm"])
    out = subprocess.run([exe, "120", "100"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "120 variables" in out.stdout and "worst posterior mean error" in out.stdout


def test_pure_c_clique_calls_equal_whole_tree_program(tmp_path):
    """examples/solve_by_clique_calls.c: a plain-C host walks the tree and calls nbp_clique_upsolve / nbp_clique_downsolve
    once per clique; the posteriors are byte-identical to the whole-tree resident program (exit status 0)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "incrementalinference.jl_amd", "csrc")
    exe = str(tmp_path / "clique_calls")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-fopenmp", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "solve_by_clique_calls.c"),
                           "-o", exe, "-L", lib, "-lnbp", f"-Wl,-rpath,{lib}", "-lm"])
    out = subprocess.run([exe, "12", "128"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "12 of 12 posteriors byte-identical" in out.stdout, out.stdout
    # the cliques of a tree level solved by four concurrent callers, one context each: the same bytes
    out = subprocess.run([exe, "60", "100", "10", "4"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "60 of 60 posteriors byte-identical" in out.stdout and "4 concurrent caller(s)" in out.stdout, out.stdout
    # the same callers -- 4, 16 and 48 of them -- on ONE context: the library merges the single-clique calls that arrive while a
    # batch is on the device into the next one (one lane, two lanes side by side, four; with and without the moment a leader
    # gives the others to come back; batches cut to five requests): one call and one result per clique, the same bytes
    for callers, env_extra in (("4", {}), ("16", {}), ("48", {"NBP_COMBINE_LANES": "4"}), ("16", {"NBP_COMBINE_LANES": "1", "NBP_COMBINE_GATHER_US": "0"}),
                               ("16", {"NBP_COMBINE_MAX": "5"}), ("16", {"NBP_SHARED_SLOTS": "24"})):  # (24 slots: a batch is cut to the two or three cliques that fit)
        env = dict(os.environ, NBP_SHARED_CTX="1", NBP_PLAN_CACHE_STATS="1", **env_extra)
        out = subprocess.run([exe, "200", "100", "20", callers], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "200 of 200 posteriors byte-identical" in out.stdout and "callers on ONE context" in out.stdout, (env_extra, out.stdout)
        if int(callers) >= 16:  # (it did merge: the widest batch held several calls -- with sixteen callers behind a 0.5 ms batch, always)
            assert "single-clique calls merged" in out.stderr, out.stderr[-500:]
    # the cliques of a tree level in one nbp_clique_solve_batch call
    out = subprocess.run([exe, "60", "100", "10", "0"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "60 of 60 posteriors byte-identical" in out.stdout and "one batched call per tree level" in out.stdout, out.stdout
    # the same, queued: every belief resident on the device (handles), a batch per tree level submitted without waiting
    # (nbp_clique_submit_batch), one wait at the end of both passes -- the same bytes again
    out = subprocess.run([exe, "60", "100", "10", "-1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "60 of 60 posteriors byte-identical" in out.stdout and "QUEUED" in out.stdout, out.stdout
    # ... and with the requests of every level KEPT across walks (the second walk re-submits what the first one built)
    out = subprocess.run([exe, "60", "100", "10", "-2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "60 of 60 posteriors byte-identical" in out.stdout and "KEPT across walks" in out.stdout, out.stdout
    # The library's PLAN CACHE (round 6): a level whose program has not changed is the cached program with new seeds, and from
    # its third run a hipGraph launch.  Six walks with a seed of their own each (NBP_WALK_SEEDS) and the whole-tree program's
    # seed again on the last two: the cached programs, re-seeded five times, must deliver its bytes -- with the cache, without
    # it, and with room for two programs only (every level evicted and rebuilt all the time).
    for env_extra in ({"NBP_PLAN_CACHE_STATS": "1"}, {"NBP_PLAN_CACHE": "0"}, {"NBP_PLAN_CACHE_ENTRIES": "2", "NBP_PLAN_CACHE_STATS": "1"}):
        for mode in ("-1", "-2"):
            env = dict(os.environ, NBP_WALKS="6", NBP_WALK_SEEDS="1", **env_extra)
            out = subprocess.run([exe, "200", "100", "20", mode], capture_output=True, text=True, timeout=600, env=env)
            assert out.returncode == 0, out.stdout + out.stderr
            assert "200 of 200 posteriors byte-identical" in out.stdout, (env_extra, mode, out.stdout)
            if env_extra.get("NBP_PLAN_CACHE_STATS") and "NBP_PLAN_CACHE_ENTRIES" not in env_extra:
                import re
                m = re.search(r"plan cache: (\d+) hits, (\d+) misses", out.stderr)
                assert m and int(m.group(1)) >= 4 * int(m.group(2)) > 0, out.stderr[-500:]  # five of six walks are hits on every level


def test_native_graph_init_equals_python_init_all(hip_backend):
    from iif_amd import native_host
    def fresh():
        return iif.generateCircularDoors(nposes=30, N=100, sightEvery=5)
    fa = fresh()
    iif.initAll(fa, backend=hip_backend, seed=31)
    fb = fresh()
    g = native_host.NativeGraph.from_fg(fb)
    need, planned = g.init_plan(31)
    be = hip_backend(100, need)
    for i, v in enumerate(fb.ls()):
        var = fb.getVariable(v)
        be.slot_write(i, var.varType.manifold, var.val, var.bw)
    prog = g.init_compile(be)
    prog.run()
    be.synchronize()
    assert set(planned) == set(fb.ls())
    for i, v in enumerate(fb.ls()):
        pts, bw = be.slot_read(i, fb.getVariable(v).varType.manifold)
        np.testing.assert_array_equal(pts, fa.getVal(v))
        np.testing.assert_array_equal(bw, fa.getVariable(v).bw)
    prog.close()
    be.close()
