"""Integer hypothesis recipe (exact-match item): the cases of test/testExplicitMultihypo.jl:8-240
against the oracle's restatement of _prepareHypoRecipe! (ExplicitDiscreteMarginalizations.jl:142-289).
Given the same mhidx vector, certainidx / activehypo / allelements must be identical."""
import ctypes as C

import numpy as np
import pytest

from oracle.oracle_backend import lib

MAXV = 6


def recipe(mh, nvars, sfidx, nullhypo, mhidx):
    L = lib()
    N = len(mhidx)
    mhidx = np.asarray(mhidx, dtype=np.int32)
    p = np.zeros(MAXV) if mh is None else np.asarray(list(mh) + [0.0] * (MAXV - len(mh)), dtype=float)
    cert = (C.c_int32 * MAXV)()
    nc = C.c_int32()
    hypo = (C.c_int32 * (MAXV + 1))()
    nact = (C.c_int32 * (MAXV + 1))()
    act = (C.c_int32 * ((MAXV + 1) * MAXV))()
    nel = (C.c_int32 * (MAXV + 1))()
    el = (C.c_int32 * ((MAXV + 1) * N))()
    ng = L.orc_hypo_recipe(0 if mh is None else 1, p.ctypes.data_as(C.POINTER(C.c_double)), nvars, sfidx, nullhypo,
                           mhidx.ctypes.data_as(C.POINTER(C.c_int32)), N, cert, C.byref(nc), hypo, nact, act, nel, el)
    return {
        "certainidx": list(cert)[: nc.value],
        "activehypo": [(hypo[g], list(act)[g * MAXV: g * MAXV + nact[g]]) for g in range(ng)],
        "allelements": [list(el)[g * N: g * N + nel[g]] for g in range(ng)],
    }


def findall(mhidx, k):
    return [i + 1 for i, v in enumerate(mhidx) if v == k]


@pytest.mark.parametrize("sfidx", [1, 2])
def test_only_nullhypothesis(sfidx):
    # testExplicitMultihypo.jl:8-58: _prepareHypoRecipe!(nothing, 20, sfidx, 2, ones(Bool,2), 0.5)
    rng = np.random.default_rng(sfidx)
    mhidx = rng.integers(0, 2, size=20)
    r = recipe(None, 2, sfidx, 0.5, mhidx)
    assert r["certainidx"] == [1, 2]
    assert r["activehypo"] == [(0, [sfidx]), (1, [1, 2]), (2, [])]
    assert r["allelements"][0] == findall(mhidx, 0)
    assert r["allelements"][1] == findall(mhidx, 1)
    assert r["allelements"][2] == []
    assert len(r["allelements"][0]) + len(r["allelements"][1]) == 20


@pytest.mark.parametrize("sfidx", [1, 2])
def test_without_multihypothesis(sfidx):
    # testExplicitMultihypo.jl:63-110
    r = recipe(None, 2, sfidx, 0.0, np.ones(20, dtype=int))
    assert r["certainidx"] == [1, 2]
    assert r["allelements"] == [[], list(range(1, 21)), []]
    assert r["activehypo"] == [(0, [sfidx]), (1, [1, 2]), (2, [])]


def test_bimodal_certain_variable():
    # testExplicitMultihypo.jl:114-148: Categorical([0, .5, .5]), sfidx = 1
    rng = np.random.default_rng(3)
    mhidx = rng.integers(2, 4, size=40)
    r = recipe([0.0, 0.5, 0.5], 3, 1, 0.0, mhidx)
    assert r["certainidx"] == [1]
    assert [h for h, _ in r["activehypo"]] == [1, 2, 3]
    assert r["activehypo"][1][1] == [1, 2] and r["activehypo"][2][1] == [1, 3]
    assert r["allelements"][0] == []
    assert r["allelements"][1] == findall(mhidx, 2)
    assert r["allelements"][2] == findall(mhidx, 3)
    assert len(r["allelements"][1]) + len(r["allelements"][2]) == 40


@pytest.mark.parametrize("sfidx,expect", [(2, [(0, [2]), (1, [1, 2]), (2, [1, 2]), (3, [2, 3])]),
                                          (3, [(0, [3]), (1, [1, 3]), (2, [2, 3]), (3, [1, 3])])])
def test_bimodal_fractional_variable(sfidx, expect):
    # testExplicitMultihypo.jl:152-240
    rng = np.random.default_rng(10 + sfidx)
    mhidx = rng.choice([0, 2, 3], size=40)
    r = recipe([0.0, 0.5, 0.5], 3, sfidx, 0.0, mhidx)
    assert r["certainidx"] == [1]
    assert r["activehypo"] == expect
    assert r["allelements"][0] == findall(mhidx, 0)
    assert r["allelements"][1] == []
    assert r["allelements"][2] == findall(mhidx, 2)
    assert r["allelements"][3] == findall(mhidx, 3)
    assert sum(len(e) for e in r["allelements"]) == 40


def test_door_sighting_pattern():
    # test/testMultiHypo3Door.jl:57 pattern scaled to 4 doors: [x, l0..l3], multihypo=[1,.25,.25,.25,.25]
    mh = [0.0, 0.25, 0.25, 0.25, 0.25]
    mhidx = np.array([2, 3, 4, 5] * 5)
    r = recipe(mh, 5, 1, 0.0, mhidx)
    assert r["certainidx"] == [1]
    assert r["activehypo"][1:] == [(2, [1, 2]), (3, [1, 3]), (4, [1, 4]), (5, [1, 5])]
    for k in range(2, 6):
        assert r["allelements"][k - 1] == findall(mhidx, k)
    mhidx = np.array([0, 2, 3, 4, 5] * 4)
    r = recipe(mh, 5, 3, 0.0, mhidx)
    assert r["activehypo"] == [(0, [3]), (1, [1, 3]), (2, [2, 3, 4, 5]), (3, [1, 3]), (4, [2, 3, 4, 5]), (5, [2, 3, 4, 5])]
    assert r["allelements"][3] == findall(mhidx, 3)
