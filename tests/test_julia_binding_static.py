"""CPU-only static checks of the Julia binding jutul.jl_amd/julia/JutulHIP.jl (there is no `julia` in the image, so the file has
never been parsed by Julia; these tests are what stands between it and a typo):

(a) every `@jh :name (argtypes...) args...` and every raw `ccall((:name, libjutul_hip), Int32, (argtypes...), args...)` against the
    C prototype in include/jutul_hip.h: the symbol exists, arity, integer / float widths, pointer-ness, handles, and the number
    of argument expressions equals the length of the type tuple;
(b) every name of the `import Jutul: ...` list and every `Jutul.name` the binding uses is defined by the reference, and every
    method the binding adds to a Jutul function has a call shape (positional arity, keywords) that the reference's own generic
    method of that seam has -- against tests/golden/jutul_api.json (signature DATA extracted from /root/reference by
    tests/golden/make_jutul_api.py; refreshed and compared when /root/reference is present);
(c) block structure: function / struct / if / for / let / do / begin / try ... end balance, per top-level definition.
"""
import json
import os
import re

import pytest

from _julia_static import (block_balance, find_methods, match_bracket, parse_arglist, split_top, strip_comments,
                           top_level_definitions)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JL = os.path.join(ROOT, "jutul.jl_amd", "julia", "JutulHIP.jl")
HDR = os.path.join(ROOT, "include", "jutul_hip.h")
API = os.path.join(ROOT, "tests", "golden", "jutul_api.json")


# ---------------------------------------------------------------------------------------------------------------- (a) the C ABI
def c_prototypes():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    handles = set(re.findall(r"typedef\s+struct\s+\w+\s*\*\s*(jh_\w+)\s*;", src))
    protos = {}
    for m in re.finditer(r"\bint32_t\s+(jh_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        name, params = m.group(1), " ".join(m.group(2).split())
        if "(*" in m.group(0) and name.endswith("_callback"):
            continue
        kinds = []
        if params not in ("void", ""):
            for p in split_top(params, ","):
                kinds.append(c_kind(p, handles))
        protos[name] = kinds
    return protos, handles


def c_kind(p, handles):
    """C parameter -> (category, pointee): categories i32 i64 f64 handle ptr cstring callback"""
    p = p.strip()
    if "(*" in p:
        return ("callback", None)
    stars = p.count("*")
    base = re.sub(r"\bconst\b", "", p.replace("*", " ")).split()
    tname = " ".join(base[:-1]) if len(base) > 1 else base[0]      # drop the parameter name
    if tname in handles:
        return ("handle", None) if stars == 0 else ("ptr", "handle")
    scal = {"int32_t": "i32", "int64_t": "i64", "double": "f64", "uint8_t": "u8", "unsigned char": "u8", "char": "char",
            "void": "void", "uint64_t": "u64", "float": "f32", "size_t": "u64", "jh_halo_callback": "callback",
            "jh_newton_report": "struct"}
    assert tname in scal, f"unknown C type in jutul_hip.h: {p!r}"
    k = scal[tname]
    if stars == 0:
        return (k, None)
    if k == "char" and "const" in p:
        return ("cstring", None)
    return ("ptr", k)


JL_SCALAR = {"Int32": "i32", "Int64": "i64", "Float64": "f64", "Cint": "i32", "Clonglong": "i64", "Cdouble": "f64"}
JL_POINTEE = {"Int32": "i32", "Int64": "i64", "Float64": "f64", "UInt8": "u8", "Cvoid": "void", "Handle": "handle", "Cchar": "char"}


def jl_kind(t):
    t = t.strip()
    if t == "Handle":
        return ("handle", None)
    if t == "Cstring":
        return ("cstring", None)
    if t in JL_SCALAR:
        return (JL_SCALAR[t], None)
    m = re.fullmatch(r"(Ptr|Ref)\{(\w+)\}", t)
    assert m, f"unreadable Julia ccall type {t!r}"
    return ("ptr", JL_POINTEE[m.group(2)])


def compatible(jk, ck):
    if jk == ck:
        return True
    if jk[0] == "ptr" and ck[0] == "ptr":
        # untyped byte buffers: char* / uint8_t* / void* take Ptr{UInt8} or Ptr{Cvoid}; any array may be handed to a void*
        if ck[1] in ("char", "u8", "void") and jk[1] in ("u8", "void", "char"):
            return True
        if ck[1] == "void":
            return True
    if jk in (("ptr", "u8"), ("ptr", "char")) and ck == ("cstring", None):   # a `const char *` byte buffer (ids, IPC handles)
        return True
    if jk == ("ptr", "void") and ck[0] == "callback":
        return True
    return False


def split_macro_args(s):
    """whitespace-separated macro arguments, brackets and strings respected"""
    out, depth, cur, i = [], 0, [], 0
    while i < len(s):
        c = s[i]
        if c == '"':
            j = i + 1
            while j < len(s) and s[j] != '"':
                j += 2 if s[j] == "\\" else 1
            cur.append(s[i:j + 1]); i = j + 1; continue
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
        if depth == 0 and c in " \t":
            if cur:
                out.append("".join(cur)); cur = []
        else:
            cur.append(c)
        i += 1
    if cur:
        out.append("".join(cur))
    return out


def call_sites():
    """[(line, name, [julia types], n_args)] of every @jh and raw ccall into libjutul_hip"""
    src = strip_comments(open(JL).read())
    sites = []
    for m in re.finditer(r"@jh[ \t]+:(\w+)[ \t]+\(", src):
        line = src.count("\n", 0, m.start()) + 1
        op = m.end() - 1
        cl = match_bracket(src, op)
        types = [t for t in split_top(src[op + 1:cl], ",") if t]
        eol = src.find("\n", cl)
        args = split_macro_args(src[cl + 1:eol].strip())
        sites.append((line, m.group(1), types, len(args)))
    for m in re.finditer(r"ccall\(\(:(\w+),[ \t]*libjutul_hip\),[ \t]*(\w+),[ \t]*\(", src):
        line = src.count("\n", 0, m.start()) + 1
        assert m.group(2) == "Int32", f"line {line}: every entry point returns int32_t"
        op = m.end() - 1
        cl = match_bracket(src, op)
        types = [t for t in split_top(src[op + 1:cl], ",") if t]
        call_open = src.find("(", m.start())
        call_close = match_bracket(src, call_open)
        args = [a for a in split_top(src[cl + 1:call_close], ",") if a]
        sites.append((line, m.group(1), types, len(args)))
    return sites


def test_every_ccall_matches_its_c_prototype():
    protos, _ = c_prototypes()
    sites = call_sites()
    assert len(sites) >= 60, "the call-site scanner lost track of the binding"
    assert len(protos) >= 110
    faults = []
    for line, name, types, nargs in sites:
        if name not in protos:
            faults.append(f"JutulHIP.jl:{line}: {name} is not declared in include/jutul_hip.h"); continue
        want = protos[name]
        if len(types) != len(want):
            faults.append(f"JutulHIP.jl:{line}: {name} takes {len(want)} arguments, the type tuple has {len(types)}"); continue
        if nargs != len(types):
            faults.append(f"JutulHIP.jl:{line}: {name}: {nargs} argument expressions for a tuple of {len(types)} types")
        for i, (t, ck) in enumerate(zip(types, want)):
            if not compatible(jl_kind(t), ck):
                faults.append(f"JutulHIP.jl:{line}: {name} argument {i + 1}: Julia {t} vs C {ck}")
    assert not faults, "\n".join(faults)


def test_the_scanner_catches_planted_abi_faults():
    """the checker itself: a wrong width, a missing pointer and a short argument list must all be reported"""
    protos, _ = c_prototypes()
    assert protos["jh_context_set_option"] == [("handle", None), ("cstring", None), ("i64", None)]
    assert protos["jh_tpfa_sizes"][-1] == ("ptr", "i32") and protos["jh_context_create"][1] == ("ptr", "handle")
    assert not compatible(jl_kind("Int32"), ("i64", None))
    assert not compatible(jl_kind("Int64"), ("ptr", "i64"))
    assert not compatible(jl_kind("Ptr{Int32}"), ("ptr", "i64"))
    assert not compatible(jl_kind("Handle"), ("ptr", "handle"))
    assert compatible(jl_kind("Ref{Handle}"), ("ptr", "handle"))
    assert compatible(jl_kind("Ptr{UInt8}"), ("ptr", "char"))
    assert split_macro_args('s.law Int32(isnothing(par) ? 0 : length(par)) (a ? b : c) "x y"') == \
        ["s.law", "Int32(isnothing(par) ? 0 : length(par))", "(a ? b : c)", '"x y"']


def test_every_entry_point_the_seams_need_is_bound():
    """the per-Newton seams of SURVEY §8b call these; a binding that drops one has lost a seam"""
    used = {name for _, name, _, _ in call_sites()}
    need = {"jh_context_create", "jh_tpfa_create_weighted", "jh_law_create", "jh_law_set_data", "jh_csr_create", "jh_vec_create",
            "jh_law_set_state", "jh_law_set_state0", "jh_law_set_sources", "jh_assemble", "jh_convergence", "jh_ilu0_create",
            "jh_ilu0_factor", "jh_krylov_create", "jh_bicgstab", "jh_gmres", "jh_vec_negate_into", "jh_increment_norm",
            "jh_update_primary", "jh_law_change_report", "jh_law_update_state0", "jh_law_reset_state", "jh_law_get_variable",
            "jh_comm_init", "jh_halo_create", "jh_halo_exchange_state", "jh_unit_diagonalize", "jh_allreduce", "jh_scale_system"}
    assert need <= used, sorted(need - used)


# ------------------------------------------------------------------------------------------------- (b) Jutul's side of the seams
def api():
    return json.load(open(API))


def test_fixture_is_current_when_the_reference_is_here():
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("/root/reference is not on this box: the committed fixture is what is checked")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_jutul_api", os.path.join(ROOT, "tests", "golden", "make_jutul_api.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fresh = json.loads(json.dumps(mod.scan(), sort_keys=True))
    assert fresh == api(), "tests/golden/jutul_api.json is stale: python tests/golden/make_jutul_api.py"


def test_every_jutul_name_the_binding_uses_exists_in_the_reference():
    a = api()
    src = strip_comments(open(JL).read())
    m = re.search(r"import Jutul:(.*?)\n\n", src, flags=re.S)
    imported = [x.strip() for x in m.group(1).replace("\n", " ").split(",") if x.strip()]
    qualified = sorted(set(re.findall(r"\bJutul\.([A-Za-z_]\w*!?)", src)))
    assert len(imported) >= 35 and len(qualified) >= 10
    assert sorted(imported) == sorted(a["_meta"]["imported"]) and qualified == a["_meta"]["qualified"], \
        "the binding's use of Jutul names changed: python tests/golden/make_jutul_api.py"
    for n in imported + qualified:
        v = a["names"][n]
        assert v["methods"] or v["types"] or v.get("external"), f"{n} is not defined anywhere in the reference"


# the binding's own methods of Jutul functions -> the reference method each one specialises (file:line of the generic method whose
# callers will reach the binding's method instead)
SEAMS = {
    "setup_equation_storage": "src/conservation/conservation.jl:137",
    "setup_linearized_system!": "src/models.jl:654",
    "align_equations_to_linearized_system!": "src/models.jl:676",
    "update_linearized_system!": "src/models.jl:774",
    "declare_pattern": "src/conservation/conservation.jl:486",
    "align_to_jacobian!": "src/conservation/conservation.jl:143",
    "update_equation!": "src/conservation/conservation.jl:572",
    "get_diagonal_entries": "src/equations.jl:631",
    "update_linearized_system_equation!": "src/equations.jl:516",
    "reset_variables!": "src/models.jl:1079",
    "reset_previous_state!": "src/models.jl:1075",
    "get_output_state": "src/models.jl:1048",
    "convergence_criterion": "src/equations.jl:619",
    "update_preconditioner!": "src/linsolve/precond/ilu.jl:37",
    "operator_nrows": "src/linsolve/precond/ilu.jl:112",
    "linear_solve!": "src/linsolve/krylov.jl:71",
    "update_primary_variables!": "src/models.jl:928",
    "update_after_step!": "src/models.jl:983",
    "reset_state_to_previous_state!": "src/models.jl:1068",
    "post_update_linearized_system!": "src/models.jl:770",
    "matrix_layout": None, "float_type": None, "index_type": None, "synchronize": None, "transfer": None, "apply!": None,
}


def binding_methods():
    src = open(JL).read()
    a = api()
    jutul_functions = {n for n, v in a["names"].items() if v["methods"]}
    out = []
    for name, line, sig in find_methods(src):
        if name in jutul_functions and not name[0].isupper():
            out.append((name, line, sig))
    return out


def test_overloads_have_the_call_shape_of_the_reference_method():
    a = api()["names"]
    methods = binding_methods()
    assert len(methods) >= 22, [m[0] for m in methods]
    faults = []
    for name, line, sig in methods:
        assert name in SEAMS, f"JutulHIP.jl:{line}: {name} extends a Jutul function this test has no seam for -- add it to SEAMS"
        ref = a[name]["methods"]
        at = SEAMS[name]
        cands = [r for r in ref if at is None or r["at"] == at]
        if not cands:
            faults.append(f"JutulHIP.jl:{line}: {name}: the reference has no method at {at} (fixture stale?)"); continue
        ok = False
        why = []
        for r in cands:
            lo, hi = r["min"], r["max"]
            # positional: the binding's method must accept every arity the generic callers use (the reference method's own range)
            if sig["max"] is not None and (hi is None or sig["max"] < hi) and not (hi is None and sig["max"] >= lo):
                why.append(f"accepts at most {sig['max']} positional arguments, {r['at']} takes up to {hi}"); continue
            if sig["min"] > lo:
                why.append(f"needs {sig['min']} positional arguments, {r['at']} is called with {lo}"); continue
            # keywords: every keyword the reference method names must be accepted (named or through a splat)
            miss = [k for k in r["keywords"] if k not in sig["keywords"] and not sig["kwsplat"]]
            if miss:
                why.append(f"does not accept the keywords {miss} of {r['at']}"); continue
            ok = True
            break
        if not ok:
            faults.append(f"JutulHIP.jl:{line}: {name}: " + "; ".join(why))
    assert not faults, "\n".join(faults)


def test_invoke_targets_exist():
    """`invoke(f, Tuple{...}, args...)` falls back to the reference's generic method: that method must exist with that arity"""
    a = api()["names"]
    src = strip_comments(open(JL).read())
    n = 0
    for m in re.finditer(r"invoke\((\w+!?),[ \t]*Tuple\{", src):
        name = m.group(1)
        op = src.find("{", m.start())
        cl = match_bracket(src, op)
        arity = len([t for t in split_top(src[op + 1:cl], ",") if t])
        assert any(r["min"] <= arity and (r["max"] is None or arity <= r["max"]) for r in a[name]["methods"]), (name, arity)
        n += 1
    assert n >= 4


# -------------------------------------------------------------------------------------------------------- (c) block structure
def test_blocks_balance_over_the_whole_file():
    faults = block_balance(open(JL).read())
    assert not faults, faults


def test_blocks_balance_per_top_level_definition():
    src = open(JL).read()
    lines = strip_comments(src).split("\n")
    defs = top_level_definitions(src)
    assert len(defs) >= 40
    for first, last, head in defs:
        chunk = "\n".join(lines[first - 1:last])
        faults = block_balance(chunk)
        assert not faults, f"JutulHIP.jl:{first}-{last} ({head[:60]}): {faults}"


def test_block_checker_catches_planted_faults():
    good = "function f(x)\n    for i in 1:3\n        x[end] += i\n    end\n    y = [i for i in 1:3 if i > 1]\n    return x\nend\n"
    assert block_balance(good) == []
    assert block_balance(good.replace("    end\n", "", 1)) != []
    assert block_balance(good + "end\n") != []
    assert block_balance("struct A\n  x::Int\nend\nmutable struct B\n  y\nend\nabstract type C end\n") == []
    assert block_balance("function g()\n  s = \"end # not code\"  # end\n  :end\nend") == []
    assert block_balance("f(x) = (1,\n") != []
    sig = parse_arglist("a, b::T = 1, c...; k = 2, kw...")
    assert (sig["min"], sig["max"], sig["keywords"], sig["kwsplat"]) == (1, None, ["k"], True)


# ------------------------------------------------------------------------------------------------ (b') dispatch ambiguities
# Two methods are ambiguous for some call when each is strictly more specific than the other in some argument and no argument
# position has disjoint types.  The lattice below is what the test knows about the types that occur in the signatures of the
# functions the binding extends: `SUPER[T]` = T's supertypes among the annotated types (reference: the struct / const lines
# recorded in the fixture, e.g. core_types.jl:875 for CompositeModel); "" is Any.  A type name the table does not know fails the
# test, so a new overload or a reference update forces a decision instead of passing silently.
SUPER = {
    # the binding's own types
    "HIPModel": ["SimulationModel", "JutulModel"], "HIPContext": ["GPUJutulContext", "JutulContext"],
    "HIPConservationLawStorage": [], "HIPEquationMember": [], "SourceRows": ["AbstractArray", "AbstractMatrix"], "HIPLinearizedSystem": ["JutulLinearSystem"], "HIPPreconditioner": ["JutulPreconditioner"],
    "HIPDistributedExecutor": ["JutulExecutor"], "SourceAccumulator": ["AbstractArray", "AbstractMatrix"],
    # reference types
    "SimulationModel": ["JutulModel"], "MultiModel": ["JutulModel"], "CompositeModel": ["SimulationModel", "JutulModel"],
    "JutulModel": [], "JutulContext": [], "GPUJutulContext": ["JutulContext"], "SingleCUDAContext": ["GPUJutulContext", "JutulContext"],
    "ParallelCSRContext": ["JutulContext"], "DefaultContext": ["JutulContext"], "CPUJutulContext": ["JutulContext"],
    "ConservationLaw": ["JutulEquation"], "ScalarTestEquation": ["JutulEquation"], "JutulEquation": [], "Pair": [],
    "ConservationLawTPFAStorage": [], "ConservationLawFiniteVolumeStorage": [], "CompactAutoDiffCache": ["JutulAutoDiffCache"],
    "JutulAutoDiffCache": [], "GenericAutoDiffCache": ["JutulAutoDiffCache"],
    "AbstractArray": [], "AbstractMatrix": ["AbstractArray"], "AbstractVector": ["AbstractArray"], "Missing": [], "Nothing": [],
    "Cells": ["JutulEntity"], "Faces": ["JutulEntity"], "JutulEntity": [], "Any": [],
    "LSystem": ["JutulLinearSystem"], "MultiLinearizedSystem": ["LSystem", "JutulLinearSystem"], "JutulLinearSystem": [],
    "LinearizedSystem": ["LSystem", "JutulLinearSystem"], "LinearizedBlock": ["LSystem", "JutulLinearSystem"],
    "GenericKrylov": [], "LUSolver": [], "JutulPreconditioner": [], "ILUZeroPreconditioner": ["JutulPreconditioner"],
    "FactorStore": [], "AMGPreconditioner": ["JutulPreconditioner"], "DiagonalPreconditioner": ["JutulPreconditioner"],
    "LUPreconditioner": ["JutulPreconditioner"], "TrivialPreconditioner": ["JutulPreconditioner"],
    "GroupWisePreconditioner": ["JutulPreconditioner"], "BoomerAMGPreconditioner": ["JutulPreconditioner"],
    "AMGCLPreconditioner": ["JutulPreconditioner"], "JacobiPreconditioner": ["JutulPreconditioner"],
    "SPAI0Preconditioner": ["JutulPreconditioner"], "StaticSparsityMatrixCSR": ["AbstractArray", "AbstractMatrix"],
    "JutulExecutor": [], "PArrayExecutor": ["JutulExecutor"], "DefaultExecutor": ["JutulExecutor"],
    "CartesianMesh": [], "CoarseMesh": [], "FastAssemblyData": [], "UnstructuredMesh": [], "TwoPointPotentialFlowHardCoded": [],
    "SparseMatrixCSC": ["AbstractArray", "AbstractMatrix"], "AbstractDict": [], "NamedTuple": [], "Real": [],
    "AbstractFloat": ["Real"], "Integer": ["Real"],
    "JutulSimulator": [], "PArraySimulator": ["JutulSimulator"], "Simulator": ["JutulSimulator"],
}
# types that are neither sub- nor supertype of each other but do intersect (a value can be both)
OVERLAP = {frozenset(("HIPModel", "CompositeModel"))}
# ambiguities accepted with a reason: (function, reference method) -> why it cannot be reached / is the reference's own shape
ACCEPTED_AMBIGUITY = {
    ("update_linearized_system_equation!", "src/equations.jl:530"):
        "nz::Missing method vs (law::ConservationLaw, s::Storage): the reference's own conservation.jl:298 has the identical shape "
        "against equations.jl:530; the binding's update_linearized_system! never passes `missing` (it passes `nothing`)",
}


def base_type(t):
    t = t.strip()
    if t == "":
        return "Any"
    t = re.sub(r"^<:", "", t)
    m = re.match(r"[\w.]+", t)
    return m.group(0).split(".")[-1]


def relation(b, r):
    """'eq' | 'b' (binding more specific) | 'r' (reference more specific) | 'disjoint' | 'overlap'"""
    for x in (b, r):
        assert x in SUPER, f"type {x!r} is not in the test's lattice (SUPER): classify it"
    if b == r:
        return "eq"
    if r == "Any" or r in SUPER[b]:
        return "b"
    if b == "Any" or b in SUPER[r]:
        return "r"
    return "overlap" if frozenset((b, r)) in OVERLAP else "disjoint"


def test_no_dispatch_ambiguity_with_the_reference_methods():
    a = api()["names"]
    faults, checked = [], 0
    for name, line, sig in binding_methods():
        bt = [base_type(t) for t in sig["types"]]
        for r in a[name]["methods"]:
            rt = [base_type(t) for t in r["types"]]
            # common arities (varargs absorb the tail as Any)
            lo = max(sig["min"], r["min"])
            hi = min(sig["max"] if sig["max"] is not None else 99, r["max"] if r["max"] is not None else 99)
            if lo > hi:
                continue
            for n in range(lo, hi + 1):
                checked += 1
                rel = [relation(bt[i] if i < len(bt) and not sig["positional"][i].endswith("...") else "Any",
                                rt[i] if i < len(rt) else "Any") for i in range(n)]
                if "disjoint" in rel:
                    continue
                if "b" in rel or "overlap" in rel:
                    if ("r" in rel or "overlap" in rel) and (name, r["at"]) not in ACCEPTED_AMBIGUITY:
                        faults.append(f"JutulHIP.jl:{line}: {name}{tuple(bt[:n])} is ambiguous with {r['at']} {tuple(rt[:n])}: {rel}")
                elif "r" not in rel and all(x == "eq" for x in rel):
                    faults.append(f"JutulHIP.jl:{line}: {name}{tuple(bt[:n])} OVERWRITES the reference method {r['at']} (same signature)")
    assert checked > 100
    assert not faults, "\n".join(faults)


def test_the_ambiguity_checker_catches_a_planted_case():
    assert relation("HIPModel", "JutulModel") == "b" and relation("Any", "Missing") == "r"
    assert relation("ConservationLaw", "Pair") == "disjoint" and relation("HIPModel", "CompositeModel") == "overlap"
    assert relation("HIPConservationLawStorage", "CompactAutoDiffCache") == "disjoint"
    assert base_type("ConservationLaw{<:Any, <:TwoPointPotentialFlowHardCoded, <:Any, <:Any}") == "ConservationLaw"
    assert base_type("Jutul.StaticCSR.StaticSparsityMatrixCSR") == "StaticSparsityMatrixCSR" and base_type("") == "Any"


# ------------------------------------------------------------------------------------------- (b'') type parameters and fields
def test_parametric_reference_types_are_used_with_their_parameter_count():
    """`SimulationModel{<:Any, <:Any, <:Any, HIPContext}` must have as many parameters as core_types.jl:241 declares, etc."""
    a = api()["names"]
    src = strip_comments(open(JL).read())
    n = 0
    for name, v in a.items():
        decl = [t for t in v["types"] if "nparams" in t]
        if not decl:
            continue
        for m in re.finditer(r"\b" + re.escape(name) + r"\{", src):
            cl = match_bracket(src, m.end() - 1)
            k = len([t for t in split_top(src[m.end():cl], ",") if t])
            assert k == decl[0]["nparams"], f"{name}{{...}} used with {k} parameters, {decl[0]['at']} declares {decl[0]['nparams']}"
            n += 1
    assert n >= 2


def test_supertypes_of_the_binding_structs_are_abstract_types_of_the_reference():
    a = api()["names"]
    src = strip_comments(open(JL).read())
    subs = re.findall(r"(?m)^(?:mutable[ \t]+)?(?:struct|abstract type)[ \t]+(\w+)[ \t]*<:[ \t]*([\w.]+)", src)
    assert len(subs) >= 7
    own_abstract = set(re.findall(r"abstract type[ \t]+(\w+)", src))
    for name, sup in subs:
        sup = sup.split(".")[-1]
        if sup in own_abstract or sup in ("AbstractMatrix",):
            continue
        kinds = [t["kind"] for t in a[sup]["types"]]
        assert "abstract type" in kinds, f"{name} <: {sup}: {sup} is {kinds or 'not a type'} in the reference (only abstract types can be subtyped)"


def test_fields_read_from_reference_structs_exist():
    f = api()["struct_fields"]
    src = strip_comments(open(JL).read())
    used = {"GenericKrylov": set(re.findall(r"\bkrylov\.(\w+)", src)), "IterativeSolverConfig": set(re.findall(r"\bcfg\.(\w+)", src)),
            "SimulationModel": set(re.findall(r"\bmodel\.(\w+)", src))}
    assert {"config", "preconditioner", "storage", "scaling", "solver"} <= used["GenericKrylov"]
    assert {"max_iterations", "min_iterations", "true_residual", "precond_side"} <= used["IterativeSolverConfig"]
    assert {"context", "domain"} <= used["SimulationModel"]
    for struct, names in used.items():
        assert names <= set(f[struct]), f"{struct} has no field(s) {sorted(names - set(f[struct]))} (reference fields: {f[struct]})"


# ------------------------------------------------------------------------------------------------ (d) every name resolves
# A typo in a function or type name is a run-time UndefVarError in Julia; nothing above would see `lenght(x)`.  Every identifier
# that is CALLED and every type used in an annotation or a parametric type must be: defined in the file, imported from Jutul, a
# function-valued argument of the enclosing function, or on the lists below -- each entry checked by hand against Julia 1.10's Base
# / Core (the lists are closed on purpose: a new name forces a look).
BASE_FUNCTIONS = {
    "all", "append!", "ccall", "collect", "copy", "cumsum", "eachindex", "empty!", "enumerate", "error", "esc", "finalizer", "findfirst",
    "first", "get", "get!", "haskey", "invoke", "isempty", "isfinite", "isnan", "isnothing", "keys", "length", "map", "max", "min",
    "minimum", "new", "pairs", "pointer", "push!", "reduce", "similar", "size", "sizeof", "something", "sqrt", "sum", "unsafe_string",
    "values", "zeros", "setindex!", "getindex",
}
BASE_TYPES = {
    "Float64", "Int32", "Int64", "Int", "UInt8", "Bool", "String", "Symbol", "Cstring", "Cvoid", "Ptr", "Ref", "Vector", "Matrix", "Array",
    "Dict", "Tuple", "Union", "Nothing", "Any", "Integer", "AbstractMatrix", "AbstractArray", "IndexCartesian", "Type",
}
FUNCTION_ARGUMENTS = {"bcast", "allgather"}     # collectives handed in by the host (setup_distributed!, set_halo!)
KEYWORDS = {"if", "for", "while", "function", "elseif", "return", "macro", "let", "do", "where", "in", "isa", "begin"}


def _code_only(src):
    s = strip_comments(src)
    s = re.sub(r'"""(?:\\.|[^\\])*?"""', lambda m: re.sub(r"[^\n]", " ", m.group(0)), s, flags=re.S)
    return re.sub(r'"(?:\\.|[^"\\\n])*"', lambda m: " " * len(m.group(0)), s)


def _binding_namespace(src, s):
    defined = {n for n, _, _ in find_methods(src)}
    defined |= set(re.findall(r"(?m)^(?:mutable[ \t]+)?struct[ \t]+(\w+)", s)) | set(re.findall(r"(?m)^abstract type[ \t]+(\w+)", s))
    defined |= set(re.findall(r"(?m)^const[ \t]+(\w+)", s)) | set(re.findall(r"(?m)^macro[ \t]+(\w+)", s))
    m = re.search(r"import Jutul:(.*?)\n\n", src, flags=re.S)
    imported = {x.strip() for x in m.group(1).replace("\n", " ").split(",") if x.strip()}
    return defined, imported


def test_every_called_function_and_every_type_name_resolves():
    src = strip_comments(open(JL).read())
    s = _code_only(open(JL).read())
    defined, imported = _binding_namespace(src, s)
    called = set(re.findall(r"(?<![\w.:@])([A-Za-z_]\w*!?)\(", s)) - KEYWORDS
    assert len(called) > 90
    unknown = sorted(called - defined - imported - BASE_FUNCTIONS - BASE_TYPES - FUNCTION_ARGUMENTS)
    assert not unknown, f"called but defined nowhere (typo, or a Base function to add to the list after checking it): {unknown}"
    # types: `::T`, `<:T`, `T{...}`, `isa T`
    types = set(re.findall(r"(?:::|<:|\bisa[ \t]+)[ \t]*([A-Z]\w*)(?![\w.])", s)) | set(re.findall(r"(?<![\w.])([A-Z]\w*)\{", s))   # (qualified Jutul.X: test (b))
    assert len(types) > 25
    unknown_t = sorted(types - defined - imported - BASE_TYPES)
    assert not unknown_t, f"type names that resolve nowhere: {unknown_t}"
    # fields of the binding's own structs: `x.field` where x is a conventional variable name of that struct type
    fields = {}
    for m in re.finditer(r"(?m)^(?:mutable[ \t]+)?struct[ \t]+(\w+)[^\n]*\n(.*?)^end", s, flags=re.S):
        names = []
        for ln in m.group(2).split("\n"):
            f = re.match(r"^[ \t]+([a-z_]\w*)(?:::[^\n]*)?[ \t]*$", ln)
            if f:
                names.append(f.group(1))
        fields[m.group(1)] = set(names)
    conv = {"s": "HIPConservationLawStorage", "m": "HIPEquationMember", "ctx": "HIPContext", "ws": "HIPKrylov"}
    for var, struct in conv.items():
        used = set(re.findall(r"(?<![\w.])" + var + r"\.([a-z_]\w*)", s))
        if var == "m":      # (`m` is also a regex-free loop variable nowhere; members only)
            used = set(re.findall(r"(?<![\w.])m\.([a-z_]\w*)", s))
        assert used, (var, struct)
        assert used <= fields[struct], f"{struct} has no field(s) {sorted(used - fields[struct])}"


def test_the_name_checker_catches_a_planted_typo():
    s = _code_only("function f(x)\n    n = lenght(x)   # typo\n    return zeros(n)\nend\n")
    called = set(re.findall(r"(?<![\w.:@])([A-Za-z_]\w*!?)\(", s)) - KEYWORDS
    assert "lenght" in called - BASE_FUNCTIONS and "zeros" in BASE_FUNCTIONS and "f" in called


BASE_VALUES = {"nothing", "true", "false", "C_NULL", "NaN", "missing", "undef", "vcat", "libjutul_hip", "ENV", "Base", "Jutul", "GC",
               "LinearAlgebra"}
MORE_KEYWORDS = {"end", "else", "try", "catch", "finally", "struct", "mutable", "const", "using", "import", "module", "abstract", "type",
                 "local", "global", "export", "new", "quote", "break", "continue"}


def undefined_locals(raw):
    """{function header: names read in the body that are neither parameters, nor assigned anywhere in the body (any `x =`, tuple
    targets, loop / lambda / do-block variables -- an over-approximation on purpose), nor module-level names, imports or Base}"""
    src, s = strip_comments(raw), _code_only(raw)
    defined, imported = _binding_namespace(src, s)
    lines = s.split("\n")
    out = {}
    for first, last, head in top_level_definitions(raw):
        if not head.startswith("function"):
            continue
        body = "\n".join(lines[first - 1:last])
        op = body.index("(")
        cl = match_bracket(body, op)
        params = set()
        for part in split_top(body[op + 1:cl].replace(";", ","), ","):
            nm = re.split(r"::", split_top(part, "=")[0].strip().rstrip("."))[0].strip()
            if nm:
                params.add(nm)
        rest = body[cl + 1:]
        assigned = set(re.findall(r"(?<![\w.])([a-z_]\w*)[ \t]*(?:[-+*/]?=)(?!=)", rest))
        for m in re.finditer(r"(?<![\w.])((?:[a-z_]\w*[ \t]*,[ \t]*)+[a-z_]\w*)[ \t]*=(?!=)", rest):
            assigned |= {x.strip() for x in m.group(1).split(",")}
        for m in re.finditer(r"\(([a-z_][\w ,()]*)\)[ \t]*(?:=(?!=)|->|in\b)", rest):
            assigned |= set(re.findall(r"[a-z_]\w*", m.group(1)))
        assigned |= set(re.findall(r"\bfor[ \t]+([a-z_]\w*)[ \t]+in\b", rest)) | set(re.findall(r"\bfor[ \t]+([a-z_]\w*)[ \t]*=", rest))
        assigned |= set(re.findall(r",[ \t]*([a-z_]\w*)[ \t]+in\b", rest)) | set(re.findall(r"(?<![\w.])([a-z_]\w*)[ \t]*->", rest))
        assigned |= set(re.findall(r"\bdo[ \t]+([a-z_]\w*)", rest))
        used = set(re.findall(r"(?<![\w.:@$])([a-z_]\w*!?)(?![\w!]*\()", rest))
        unk = used - params - assigned - defined - imported - BASE_VALUES - KEYWORDS - MORE_KEYWORDS - BASE_FUNCTIONS - FUNCTION_ARGUMENTS
        if unk:
            out[head[:80]] = sorted(unk)
    return out


def test_no_function_reads_a_name_that_is_defined_nowhere():
    raw = open(JL).read()
    assert undefined_locals(raw) == {}
    planted = raw.replace("    n = iters[]\n", "    n = itres[]\n", 1)
    assert planted != raw
    bad = undefined_locals(planted)
    assert any("itres" in v for v in bad.values()), bad


# ------------------------------------------------------------------------------------------- (e) call sites of Jutul functions
def jutul_call_sites():
    """[(line, name, n_positional, [keyword names], has_splat)] of every CALL (not definition) of an imported or `Jutul.`-qualified
    function in the binding"""
    raw = open(JL).read()
    src = strip_comments(raw)
    s = _code_only(raw)
    a = api()["names"]
    funcs = {n for n, v in a.items() if v["methods"] and not n[0].isupper()}
    def_starts = set()
    for m in re.finditer(r"(?m)^[ \t]*(?:function[ \t]+)?((?:[A-Za-z_]\w*\.)*[A-Za-z_]\w*!?)[ \t]*\(", s):
        cl = match_bracket(s, m.end() - 1)
        if m.group(0).lstrip().startswith("function") or re.match(r"[ \t]*(?:::[^\n=]*?)?(?:where[^\n=]*?)?=(?!=)", s[cl + 1:cl + 200]):
            def_starts.add(m.start(1))
    out = []
    for m in re.finditer(r"(?<![\w!])((?:Jutul\.)?)([A-Za-z_]\w*!?)\(", s):
        name = m.group(2)
        if name not in funcs or m.start(1) in def_starts or m.start(2) in def_starts:
            continue
        if s[max(0, m.start() - 1):m.start()] == ".":     # some_object.name(...)
            continue
        if re.search(r"invoke\($", s[max(0, m.start() - 7):m.start()]):
            continue
        op = m.end() - 1
        cl = match_bracket(src, op)                         # (same offsets: _code_only keeps lengths)
        inner = src[op + 1:cl]
        semi = split_top(inner, ";")
        pos = [p for p in split_top(semi[0], ",") if p]
        kws = [p for part in semi[1:] for p in split_top(part, ",") if p]
        npos, kwn, splat = 0, [], False
        for p in pos:
            if re.match(r"^[a-z_]\w*[ \t]*=(?!=)", p):
                kwn.append(p.split("=")[0].strip())
            elif p.endswith("..."):
                splat = True
            else:
                npos += 1
        for k in kws:
            if k.endswith("..."):
                splat = True
            else:
                kwn.append(k.split("=")[0].strip())
        out.append((s.count("\n", 0, m.start()) + 1, name, npos, kwn, splat))
    return out


def test_every_call_of_a_jutul_function_fits_a_method():
    """a call `f(a, b; k = v)` of an imported / qualified Jutul function must fit SOME method of f -- the reference's or one the
    binding adds: positional count inside the method's range, every keyword named by the method or swallowed by its splat"""
    a = api()["names"]
    own = {}
    for name, _line, sig in find_methods(open(JL).read()):
        own.setdefault(name, []).append(dict(min=sig["min"], max=sig["max"], keywords=sig["keywords"], kwsplat=sig["kwsplat"]))
    sites = jutul_call_sites()
    assert len(sites) >= 25, sites
    faults = []
    for line, name, npos, kwn, splat in sites:
        cands = a[name]["methods"] + own.get(name, [])
        ok = False
        for r in cands:
            if npos < r["min"] and not splat:
                continue
            if r["max"] is not None and npos > r["max"]:
                continue
            if any(k not in r["keywords"] and not r["kwsplat"] for k in kwn):
                continue
            ok = True
            break
        if not ok:
            faults.append(f"JutulHIP.jl:{line}: {name} called with {npos} positional arguments and keywords {kwn}: no method takes that "
                          f"(shapes: {[(r['min'], r['max'], r['keywords'], r['kwsplat']) for r in cands][:6]})")
    assert not faults, "\n".join(faults)


def test_the_call_site_checker_sees_the_calls_it_should():
    sites = {(n, p, tuple(k)) for _, n, p, k, _ in jutul_call_sites()}
    assert ("linear_solve_return", 3, ("prepare",)) in sites and ("update_preconditioner!", 5, ()) in sites
    assert ("variable_change_report", 3, ()) in sites and ("update_after_step!", 5, ()) in sites
    a = api()["names"]
    # a planted extra argument would fit no method of the reference
    assert not any(r["min"] <= 4 and (r["max"] is None or 4 <= r["max"]) for r in a["variable_change_report"]["methods"])
