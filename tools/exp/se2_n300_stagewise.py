"""config 4's shape (SE(2) lattice, 8 x 100) at N = 300: every stage of the tree program against the oracle on the oracle's state
(the machinery of tests/test_gpu_stagewise_parity.py) -- where does the device first part from it?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import test_gpu_stagewise_parity as T
from parity_utils import iif
from oracle.oracle_backend import OracleBackend
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T.FULL["config4_full_size_probe"] = lambda: iif.generateSE2Lattice(rows=rows, cols=100, N=N, closeEvery=5)
try:
    T.test_every_stage_of_the_tree_program_on_the_oracles_state(lambda n, s, side_ints=0: OracleBackend(n, s, side_ints, threads=32),
                                                                 lambda n, s, side_ints=0: iif.HipBackend(n, s, side_ints=side_ints), "config4_full_size_probe")
    print("every stage agrees")
except Exception as e:  # noqa: BLE001
    import traceback
    traceback.print_exc()
    print("FAILED:", str(e)[:3000])
