"""not gpu: the generators of the differential fuzz (tests/fuzz_proposals.py, fuzz_products.py, fuzz_degenerate.py, fuzz_graphs.py),
run with the ORACLE on both sides: every descriptor they draw is one the checker accepts, every output is finite, and the
comparison machinery reports nothing between a thing and itself.  (The device's side of them is tests/test_gpu_fuzz_ops.py.)"""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name, oracle_both_sides=True):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tests", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not oracle_both_sides:
        return mod
    from oracle.oracle_backend import OracleBackend

    class OracleOnBothSides:
        solver = mod.iif.solver
        HipBackend = staticmethod(lambda N, n, side_ints=0: OracleBackend(N, n, side_ints, threads=8))

    mod.iif = OracleOnBothSides
    return mod


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_proposal_and_deconvolution_generators(seed):
    fz = load("fuzz_proposals")
    for B, which, simple in ((40, None, False), (24, fz.KINDS[4], False), (30, fz.KINDS[1], True)):
        n, bad = fz.run_launch(1000 * seed + B, 64, B, which, simple)
        assert n == B and not bad
    n, bad = fz.run_deconv_launch(seed, 64, 20)
    assert n == 20 and not bad


@pytest.mark.parametrize("seed", [0, 1])
def test_product_generator(seed):
    fz = load("fuzz_products")
    fz.DUP = bool(seed)
    for B, man in ((30, None), (20, fz.MANS[4])):
        n, bad = fz.run_launch(7000 * seed + B, 64, B, man)
        assert n == B and not bad


def test_degenerate_generator():
    fd = load("fuzz_degenerate")
    for man in (1, 2, 3, 4, 5):
        for count, name in ((64, "identical"), (3, "three values"), (17, "two far clusters"), (64, "huge offset"), (2, "tiny spread")):
            bad, finite = fd.run_case(man, 64, count, name, 5)
            assert finite and not bad, (man, count, name)


def test_graph_generator_draws_every_manifold_and_solvable_graphs():
    fg_mod = load("fuzz_graphs", oracle_both_sides=False)  # (the graphs are only drawn here: their solves need the device)
    kinds = set()
    for seed in range(24):
        fg, info = fg_mod.random_graph(seed)
        kinds.add(info["kind"])
        assert len(fg.ls()) == info["n"] and len(fg.lsf()) >= info["n"]
    assert kinds == {0, 1, 2, 3, 4}
