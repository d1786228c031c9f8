"""CPU-side checks of the C ABI: the library loads, exports every symbol include/nbp.h declares,
the ctypes mirror has the same struct sizes, and the product path fails loudly without a GPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

from parity_utils import abi, iif

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    return abi.load_library()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "nbp.h")).read()
    declared = set(re.findall(r"\b(nbp_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(abi.EXPORTS), declared ^ set(abi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layout_matches_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "nbp.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(nbp_proposal_desc),sizeof(nbp_product_desc),sizeof(nbp_copy_desc),sizeof(nbp_diag),'
                   'offsetof(nbp_proposal_desc,comp),offsetof(nbp_proposal_desc,seed));return 0;}')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(abi.ProposalDesc), C.sizeof(abi.ProductDesc), C.sizeof(abi.CopyDesc), C.sizeof(abi.Diag),
            abi.ProposalDesc.comp.offset, abi.ProposalDesc.seed.offset]
    assert got == want


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(iif.NbpError, match="no HIP device"):
        iif.HipBackend(100, 4)
    fg = iif.generateGraph_Kaess()
    with pytest.raises(iif.NbpError):
        iif.solveTree(fg)


def test_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, "incrementalinference.jl_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "liboracle" not in txt and "oracle_backend" not in txt and "import oracle" not in txt, f
