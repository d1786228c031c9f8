"""Generate the golden fixtures of tests/golden/*.npz.

Source of truth: the CPU oracle (oracle/nbp_oracle.c) with fixed Philox keys.  The reference ships
no golden vectors and cannot be executed here, so these fixtures pin the
restatement (and the HIP kernels) against silent drift; they are NOT reference outputs.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_cases import CASES, run_case  # noqa: E402
from oracle.oracle_backend import OracleBackend  # noqa: E402


def main():
    for name, case in CASES.items():
        out = run_case(case, lambda N, n, side_ints=0: OracleBackend(N, n, side_ints))
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
