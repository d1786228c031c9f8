"""Every stage of a tree program against the oracle on the oracle's state (the machinery of tests/test_gpu_stagewise_parity.py) at
particle counts BASELINE does not use (the kernels' geometries follow N) and with solver parameters it does not use.
Usage: stagewise_any_n.py <shape 2|3|4|5> <N> [size] [param=value ...]   (SolverParams fields: productNiter, inflateCycles, gibbsIters, ...)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_stagewise_parity as T
from parity_utils import iif
from oracle.oracle_backend import OracleBackend
shape, N = sys.argv[1], int(sys.argv[2])
sets = [a.split("=") for a in sys.argv[3:] if "=" in a]
pos = [a for a in sys.argv[3:] if "=" not in a]
size = int(pos[0]) if pos else {"2": 300, "3": 300, "4": 6, "5": 300}[shape]
make = {"2": lambda: iif.generateChainEuclid(size, vardims=2, priorEvery=100, N=N),
        "3": lambda: iif.generateCircularDoors(nposes=size, N=N, sightEvery=25),
        "4": lambda: iif.generateSE2Lattice(rows=size, cols=100, N=N, closeEvery=5),
        "5": lambda: iif.generateMixtureChain(nvars=size, N=N, priorEvery=100)}[shape]
name = f"config{shape}_full_size_probe_N{N}"


def make_with_params():
    fg = make()
    for k, v in sets:
        cur = getattr(fg.solverParams, k)
        setattr(fg.solverParams, k, type(cur)(float(v)) if not isinstance(cur, bool) else v not in ("0", "False", "false"))
    return fg


T.FULL[name] = make_with_params
tag = " ".join(f"{k}={v}" for k, v in sets)
try:
    T.test_every_stage_of_the_tree_program_on_the_oracles_state(lambda n, s, side_ints=0: OracleBackend(n, s, side_ints, threads=48),
                                                                 lambda n, s, side_ints=0: iif.HipBackend(n, s, side_ints=side_ints), name)
    print(f"shape {shape} N={N} size {size} {tag}: every stage agrees")
except Exception as e:  # noqa: BLE001
    print(f"shape {shape} N={N} size {size} {tag}: FAILED: {str(e)[:400]}")
