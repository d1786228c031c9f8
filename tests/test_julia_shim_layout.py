"""CPU: the structs of the Julia shim (ext/IIFNbpExt.jl) against the C headers.  Julia is not installed here, so the
file is parsed: every `struct Nbp... end` block gives (field name, Julia type) pairs, a C-ABI layout is computed
from them (natural alignment, NTuple = inline array, Ptr = 8 bytes) and compared with sizeof / offsetof / field order
of the corresponding C struct as gcc sees it.  A shim that drifts from the header fails here instead of corrupting
descriptor arrays at run time."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "ext", "IIFNbpExt.jl")

C_NAME = {"NbpProposalDesc": "nbp_proposal_desc", "NbpProductDesc": "nbp_product_desc", "NbpCopyDesc": "nbp_copy_desc",
          "NbpDiag": "nbp_diag", "NbpSolverParams": "nbp_solver_params", "NbpFactorSpec": "nbp_factor_spec",
          "NbpTreeBelief": "nbp_tree_belief", "NbpCliqueDesc": "nbp_clique_desc", "NbpCliqueRequest": "nbp_clique_request"}
PRIM = {"Int32": 4, "UInt32": 4, "Int64": 8, "UInt64": 8, "Float64": 8, "UInt8": 1, "Int8": 1}


def parse_shim():
    src = open(SHIM).read()
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"^const (NBP_[A-Z_]+) = (\d+)\s*$", src, re.M)}
    structs = {}
    for m in re.finditer(r"^struct (Nbp\w+)\n(.*?)^end", src, re.M | re.S):
        fields = []
        for line in m.group(2).splitlines():
            line = line.split("#")[0].strip()
            if not line:
                continue
            fm = re.fullmatch(r"(\w+)::(.+)", line)
            assert fm, f"{m.group(1)}: one `name::Type` per line, got {line!r}"
            fields.append((fm.group(1), fm.group(2).strip()))
        structs[m.group(1)] = fields
    return consts, structs


def size_align(jtype, consts):
    if jtype in PRIM:
        return PRIM[jtype], PRIM[jtype]
    if jtype.startswith("Ptr{"):
        return 8, 8
    m = re.fullmatch(r"NTuple\{(.+),\s*(\w+)\}", jtype)
    assert m, f"unsupported field type {jtype}"
    n = eval(m.group(1), {}, consts)
    s, a = size_align(m.group(2), consts)
    return n * s, a


def julia_layout(fields, consts):
    off, amax, out = 0, 1, []
    for name, jt in fields:
        s, a = size_align(jt, consts)
        off = (off + a - 1) // a * a
        out.append((name, off, s))
        off += s
        amax = max(amax, a)
    return out, (off + amax - 1) // amax * amax


def c_layout(cname, names, tmp_path):
    probe = ['#include <stdio.h>', '#include <stddef.h>', '#include "nbp_host.h"', 'int main(){', f'  {cname} x; (void)x;',
             f'  printf("%zu\\n", sizeof({cname}));']
    for n in names:
        probe.append(f'  printf("{n} %zu %zu\\n", offsetof({cname}, {n}), sizeof(x.{n}));')
    probe.append("  return 0;}")
    src = tmp_path / f"{cname}.c"
    src.write_text("\n".join(probe))
    exe = tmp_path / cname
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    lines = subprocess.check_output([str(exe)]).decode().split("\n")
    return [(p[0], int(p[1]), int(p[2])) for p in (l.split() for l in lines[1:] if l)], int(lines[0])


def c_field_names(cname):
    hdr = open(os.path.join(ROOT, "include", "nbp.h")).read() + open(os.path.join(ROOT, "include", "nbp_host.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, re.S).group(1)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):  # `int32_t a, b;` and `const int32_t *a, *b;`
            names.append(re.search(r"(\w+)\s*(?:\[[^\]]*\])*\s*$", part.strip()).group(1))
    return names


def test_every_boundary_struct_is_mirrored():
    consts, structs = parse_shim()
    assert set(C_NAME) == set(structs), set(C_NAME) ^ set(structs)
    for k, v in (("NBP_MAXV", 6), ("NBP_MAXF", 128), ("NBP_MAXC", 4), ("NBP_COMP_STRIDE", 13)):
        assert consts[k] == v
        assert re.search(r"#define %s %d\b" % (k, v), open(os.path.join(ROOT, "include", "nbp.h")).read())


def test_julia_structs_match_the_headers(tmp_path):
    consts, structs = parse_shim()
    for jname, cname in C_NAME.items():
        fields = structs[jname]
        assert [n for n, _ in fields] == c_field_names(cname), (jname, "field names / order differ from the header")
        jl, jsize = julia_layout(fields, consts)
        cl, csize = c_layout(cname, [n for n, _ in fields], tmp_path)
        assert jl == cl, (jname, [(a, b) for a, b in zip(jl, cl) if a != b])
        assert jsize == csize, (jname, jsize, csize)
    # the figures DESIGN.md / INTEGRATION.md quote
    assert julia_layout(structs["NbpProposalDesc"], consts)[1] == 584


def test_shim_ccalls_name_exported_symbols():
    from parity_utils import abi
    from iif_amd import native_host
    called = set(re.findall(r"ccall\(\(:(\w+), libnbp\)", open(SHIM).read()))
    assert called and called <= set(abi.EXPORTS) | set(native_host.HOST_EXPORTS), called - set(abi.EXPORTS) - set(native_host.HOST_EXPORTS)
    for need in ("nbp_clique_upsolve", "nbp_clique_downsolve", "nbp_conv", "nbp_manifold_product", "nbp_kde_bandwidth", "nbp_ctx_create"):
        assert need in called


# ---- every ccall's (return type, argument types) against the prototype in the headers -------------------------------
# C type -> the set of Julia ccall argument types that pass it correctly (by value 4/8-byte scalars, pointers of any
# pointee the header names: Ptr{T} / Ref{T} / Ptr{Cvoid}; a Julia Vector passed to Ptr{T} is its data pointer)
_JL_STRUCT = {v: k for k, v in C_NAME.items()}


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def c_prototypes():
    hdr = open(os.path.join(ROOT, "include", "nbp.h")).read() + open(os.path.join(ROOT, "include", "nbp_host.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*([\w ]+?[\w\*])\s*\b(nbp_\w+)\(([^;{]*?)\)\s*;", hdr, re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3)
        if ret.startswith("typedef") or ret.startswith("#"):
            continue
        al = [] if args.strip() in ("", "void") else _split_top(" ".join(args.split()))
        types = []
        for a in al:
            a = re.sub(r"\b(\w+)\s*(\[[^\]]*\])?$", "", a.strip()).strip() if not a.strip().endswith("*") else a.strip()  # drop the parameter name
            types.append(a.replace("const ", "").replace(" const", "").replace(" ", ""))
        protos[name] = (ret.replace("const ", "").replace(" ", ""), types)
    return protos


def _ok(ctype, jtype):
    scal = {"int32_t": {"Int32", "Cint"}, "nbp_status": {"Int32", "Cint"}, "int64_t": {"Int64"}, "uint64_t": {"UInt64"},
            "double": {"Float64", "Cdouble"}}
    if ctype in scal:
        return jtype in scal[ctype]
    if ctype.endswith("**"):  # array of pointers
        base = ctype[:-2]
        return jtype in ("Ptr{Ptr{%s}}" % _jl_scalar(base), "Ptr{Ptr{Cvoid}}", "Ref{Ptr{Cvoid}}")
    if ctype.endswith("*"):
        base = ctype[:-1]
        if base in ("void", "nbp_ctx", "nbp_program", "nbp_comm", "nbp_graph", "nbp_tree", "nbp_clique_ticket", "nbp_read_token"):  # opaque handles
            return jtype in ("Ptr{Cvoid}",)
        if base == "char":
            return jtype in ("Cstring", "Ptr{UInt8}", "Ptr{Cchar}")
        j = _jl_scalar(base)
        return jtype in ("Ptr{%s}" % j, "Ref{%s}" % j)
    return False


def _jl_scalar(base):
    return {"int32_t": "Int32", "int64_t": "Int64", "uint64_t": "UInt64", "uint8_t": "UInt8", "double": "Float64"}.get(base, _JL_STRUCT.get(base, base))


def shim_ccalls():
    src = open(SHIM).read()
    out = []
    for m in re.finditer(r"ccall\(\(:(\w+), libnbp\),\s*([\w\{\}]+),\s*\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        args = src[m.end():i - 1]
        out.append((m.group(1), m.group(2), [a for a in _split_top(" ".join(args.split())) if a]))
    return out


def test_shim_ccall_signatures_match_the_prototypes():
    protos = c_prototypes()
    calls = shim_ccalls()
    assert len(calls) >= 8
    for name, jret, jargs in calls:
        assert name in protos, f"{name}: no prototype in include/"
        cret, cargs = protos[name]
        if cret.endswith("*"):
            assert jret in ("Ptr{Cvoid}", "Cstring", "Ptr{UInt8}"), (name, cret, jret)
        else:
            assert _ok(cret, jret), (name, "return type", cret, jret)
        assert len(jargs) == len(cargs), (name, "argument count", cargs, jargs)
        for k, (ct, jt) in enumerate(zip(cargs, jargs)):
            assert _ok(ct, jt), (name, f"argument {k}", ct, jt)


def test_shim_forwards_iters_and_reports_only_touched_labels():
    src = open(SHIM).read()
    up = src[src.index("function upGibbsCliqueDensity"):src.index("function solveCliqDownFrontalProducts!")]
    assert re.search(r"runclique\(:up,.*iters\)", up), "upGibbsCliqueDensity must forward its `iters` (SolveTree.jl:171,216-227)"
    assert "for l in touched" in up and "directFrtlMsgIDs" in up and "directPriorMsgIDs" in up
    assert "p.beliefs[i].n_pts" in src  # the count written back by libnbp, not N


# The reference methods the shim specialises, by the types of their positional arguments (the data of a dispatch check, with
# the file:line of each definition; an untyped argument is Any).  A shim method with exactly these types would OVERWRITE the
# reference's method (an error while an extension precompiles on Julia >= 1.10) and its `invoke` fall-back would reach itself.
REFERENCE_SIGNATURES = {
    "upGibbsCliqueDensity": ["AbstractDFG", "TreeClique", "Symbol", "Any", "Int", "Bool", "Int", "Any"],   # services/SolveTree.jl:164-173
    "solveCliqDownFrontalProducts!": ["AbstractDFG", "TreeClique", "SolverParams", "Any"],                 # CliqStateMachineUtils.jl:479-486
    "addLikelihoodsDifferentialCHILD!": ["AbstractDFG", "Vector{Symbol}", "AbstractDFG"],                  # services/TreeMessageUtils.jl:279-286
    "approxConvBelief": ["AbstractDFG", "DFGFactor", "Symbol", "AbstractVector"],                                    # services/ApproxConv.jl:4-10
}
NARROWER = {("GraphsDFG", "AbstractDFG")}  # LocalDFG = GraphsDFG <: AbstractDFG (DistributedFactorGraphs)


def _positional_types(sig):
    out = []
    for a in _split_top(sig.split(";")[0]):
        a = a.split("=")[0].strip()
        out.append(a.split("::", 1)[1].strip() if "::" in a else "Any")
    return out


def test_shim_overrides_are_more_specific_than_the_reference_methods():
    src = open(SHIM).read()
    for name, ref in REFERENCE_SIGNATURES.items():
        heads = re.findall(r"^function " + re.escape(name) + r"\((.*?)\)\n", src, flags=re.S | re.M)
        assert heads, name
        for head in heads:
            mine = _positional_types(head)
            assert len(mine) >= len(ref) - 1, (name, mine)
            narrower = False
            for m, r in zip(mine, ref):
                if m == r:
                    continue
                assert (m, r) in NARROWER or m.startswith(r + "{"), (name, "argument type neither equal nor narrower", m, r)
                narrower = True
            assert narrower, f"{name}: the shim method has the reference's own signature -- it would overwrite it"
        # the fall-back reaches the reference's method by its own signature
        for m in re.finditer(r"invoke\(" + re.escape(name) + r",\s*Tuple\{", src):
            k, depth = m.end(), 1
            while depth:  # the matching brace of Tuple{
                depth += {"{": 1, "}": -1}.get(src[k], 0)
                k += 1
            types = _split_top(src[m.end():k - 1])
            assert types == ref[:len(types)] and types[0] == "AbstractDFG", (name, types)
