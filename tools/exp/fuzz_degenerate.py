"""runner of tests/fuzz_degenerate.py (the generator and the comparison live with the tests: they drive the oracle, which only tests may)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import fuzz_degenerate
fuzz_degenerate.main()
