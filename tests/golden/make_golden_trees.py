"""Generate the tree-level golden fixtures tests/golden/tree_*.npz from the CPU oracle (fixed Philox keys): the
beliefs after graph initialisation and the posteriors after one solveTree of one small graph per BASELINE.json
configuration.  Like make_golden.py they pin the restatement and the HIP path against drift; they are not reference
outputs (the reference cannot be executed here).   Run:  python tests/golden/make_golden_trees.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_tree_cases import TREE_CASES, beliefs_of, run_init, run_solve  # noqa: E402
from oracle.oracle_backend import OracleBackend  # noqa: E402


def main():
    be = lambda N, n, side_ints=0: OracleBackend(N, n, side_ints, threads=8)
    for name in TREE_CASES:
        fg = run_init(name, be)
        out = beliefs_of(fg, "init")
        out.update(beliefs_of(run_solve(name, be, out), "post"))
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(name, len(fg.ls()), "variables", os.path.getsize(os.path.join(HERE, f"{name}.npz")) // 1024, "KB")


if __name__ == "__main__":
    main()
