"""Static contract test of the reference-side binding julia/JCDFHip.jl (Julia is not in the image, so the glue never
runs here): every `ccall((:name, libjcdf), Ret, (Args...), actual...)` and the two timing structs are parsed out of the
Julia text and checked against include/jcdf.h and the ctypes prototypes the GPU tests call through — names, arity,
return type, argument types, number of actual arguments — and the dispatch of run_gpu_fock_build!
(DensityFitting.jl:78-90: dense when df_force_dense, contraction_mode "denseGPU", or adaptively below 800 basis
functions on one rank) is present with both labels.  Also: include/jcdf.h <-> _lib.PROTOTYPES, prototype by prototype."""
import ctypes as C
import os
import re

import juliachem_jl_amd  # noqa: F401  (import shim)
from juliachem_jl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = open(os.path.join(ROOT, "julia", "JCDFHip.jl")).read()
HEADER = open(os.path.join(ROOT, "include", "jcdf.h")).read()

# Julia ccall type -> (kind, detail): what the C side must declare at that position
JULIA_TYPES = {
    "Int32": ("i32", None), "Int64": ("i64", None), "Cdouble": ("f64", None), "Float64": ("f64", None),
    "Cstring": ("ptr", "char"), "Ptr{Cvoid}": ("ptr", "void"), "Ptr{Float64}": ("ptr", "double"),
    "Ptr{Int64}": ("ptr", "int64_t"), "Ptr{Int32}": ("ptr", "int32_t"), "Ref{Ptr{Cvoid}}": ("ptr", "handle*"),
    "Ref{JCDFTimings}": ("ptr", "jcdf_timings"), "Ptr{JCDFTimings}": ("ptr", "jcdf_timings"),
    "Ref{JCDFGroupTimings}": ("ptr", "jcdf_group_timings"),
}


def _split_top(s):
    """split at top-level commas (parentheses, brackets and braces nest)"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _strip_comments(src):
    return "\n".join(line.split("#")[0] if "#" in line and '"' not in line.split("#")[0][-1:] else line for line in src.splitlines())


def julia_ccalls():
    """[(name, ret, [arg types], n_actual_args)] for every ccall of the glue"""
    src = re.sub(r"#[^\n]*", "", GLUE)                       # (no '#' inside string literals of a ccall in this file)
    calls = []
    for m in re.finditer(r"ccall\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        parts = _split_top(src[m.end():i - 1])
        sym = re.fullmatch(r"\(:(\w+),\s*libjcdf\)", parts[0])
        assert sym, parts[0]
        args = parts[2].strip()
        assert args.startswith("(") and args.endswith(")"), args
        types = _split_top(args[1:-1])
        calls.append((sym.group(1), parts[1], types, len(parts) - 3))
    return calls


def c_prototypes():
    """{name: (ret, [arg C types])} parsed from include/jcdf.h"""
    txt = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?[a-zA-Z_0-9]+\s*\**)\s*(jcdf_[a-z_0-9A-Z]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.M | re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        alist = [] if args in ("void", "") else [a.strip() for a in args.split(",")]
        protos[name] = (ret, alist)
    return protos


def c_kind(decl):
    """C parameter / return declaration -> (kind, detail)"""
    d = decl.replace("const ", "").strip()
    if "*" in d:
        base = d.split("*")[0].strip()
        stars = d.count("*")
        if base in ("jcdf_handle", "jcdf_group") and stars == 2:
            return ("ptr", "handle*")
        if base in ("jcdf_handle", "jcdf_group", "void"):
            return ("ptr", "void")
        return ("ptr", base)
    base = d.split()[0]
    return {"int32_t": ("i32", None), "int64_t": ("i64", None), "double": ("f64", None)}[base]


def ctypes_kind(t):
    if t is C.c_int32:
        return ("i32", None)
    if t is C.c_int64:
        return ("i64", None)
    if t is C.c_double:
        return ("f64", None)
    if t is C.c_char_p:
        return ("ptr", "char")
    if t is C.c_void_p:
        return ("ptr", "void")
    if isinstance(t, type) and issubclass(t, C._Pointer):
        inner = t._type_
        name = {C.c_void_p: "handle*", C.c_int64: "int64_t", C.c_int32: "int32_t", C.c_double: "double"}.get(inner)
        return ("ptr", name or inner.__name__)
    raise AssertionError(t)


def _compatible(a, b):
    """same scalar kind; pointers match when one side is untyped (void) or both name the same pointee"""
    if a[0] != b[0]:
        return False
    if a[0] != "ptr":
        return True
    return "void" in (a[1], b[1]) or a[1] == b[1]


def test_header_and_ctypes_prototypes_agree():
    protos = c_prototypes()
    assert set(protos) == set(_lib.PROTOTYPES), (sorted(set(protos) ^ set(_lib.PROTOTYPES)))
    for name, (ret, args) in protos.items():
        res, argtypes = _lib.PROTOTYPES[name]
        assert len(args) == len(argtypes), (name, args, argtypes)
        assert _compatible(c_kind(ret), ctypes_kind(res)), (name, ret, res)
        for k, (a, t) in enumerate(zip(args, argtypes)):
            assert _compatible(c_kind(a), ctypes_kind(t)), (name, k, a, t)


def test_every_ccall_of_the_glue_matches_the_header():
    protos = c_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 20
    for name, ret, types, n_actual in calls:
        assert name in protos, "ccall of %s: not declared in include/jcdf.h" % name
        cret, cargs = protos[name]
        assert len(types) == len(cargs), (name, types, cargs)
        assert n_actual == len(types), "ccall of %s passes %d arguments for %d declared types" % (name, n_actual, len(types))
        assert ret in JULIA_TYPES, (name, ret)
        assert _compatible(JULIA_TYPES[ret], c_kind(cret)), (name, ret, cret)
        for k, (jt, ca) in enumerate(zip(types, cargs)):
            assert jt in JULIA_TYPES, (name, k, jt)
            jk, ck = JULIA_TYPES[jt], c_kind(ca)
            assert jk[0] == ck[0], (name, k, jt, ca)
            if jk[0] == "ptr":                          # a typed Julia pointer must point at what the header declares
                assert _compatible(jk, ck), (name, k, jt, ca)
                if jk[1] not in ("void",) and ck[1] not in ("void",):
                    assert jk[1] == ck[1], (name, k, jt, ca)


def test_the_glue_calls_what_a_run_needs():
    names = {c[0] for c in julia_ccalls()}
    for need in ("jcdf_create", "jcdf_destroy", "jcdf_last_error", "jcdf_configure", "jcdf_set_metric", "jcdf_set_core_hamiltonian",
                 "jcdf_push_three_center", "jcdf_set_exchange_screening", "jcdf_fock_build",
                 "jcdf_group_create", "jcdf_group_destroy", "jcdf_group_last_error", "jcdf_group_configure", "jcdf_group_set_metric",
                 "jcdf_group_set_core_hamiltonian", "jcdf_group_push_three_center", "jcdf_group_set_exchange_screening",
                 "jcdf_group_fock_build", "jcdf_group_transport"):
        assert need in names, need


def _julia_struct(name):
    m = re.search(r"struct %s\b[^\n]*\n(.*?)\nend" % name, GLUE, flags=re.S)
    assert m, name
    return [tuple(x.strip() for x in line.split("#")[0].split("::")) for line in m.group(1).splitlines() if "::" in line]


def _c_struct(name):
    txt = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    m = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), txt, flags=re.S)
    assert m, name
    return [tuple(reversed(f.strip().split())) for f in m.group(1).split(";") if f.strip()]


def test_timing_structs_have_the_same_layout_on_all_three_sides():
    for jname, cname, ct in (("JCDFTimings", "jcdf_timings", _lib.jcdf_timings),
                             ("JCDFGroupTimings", "jcdf_group_timings", _lib.jcdf_group_timings)):
        jl, cc = _julia_struct(jname), _c_struct(cname)
        assert [f for f, _ in jl] == [f for f, _ in cc] == [f for f, _ in ct._fields_], (jname, jl, cc)
        assert all(t == "Cdouble" for _, t in jl) and all(t == "double" for _, t in cc)
        assert all(t is C.c_double for _, t in ct._fields_)
        assert C.sizeof(ct) == 8 * len(jl)


def test_dense_and_adaptive_dispatch_of_the_reference_is_in_the_glue():
    """run_gpu_fock_build! (DensityFitting.jl:78-90): unscreened path when df_force_dense, contraction_mode "denseGPU", or
    df_use_adaptive with fewer than 800 basis functions on a single rank — through the Julia boundary the headline
    configuration (C20H42, N = 510) must not be Schwarz-screened."""
    m = re.search(r"function use_dense_map\(.*?\n(.*?)\nend", GLUE, flags=re.S)
    assert m
    body = m.group(1)
    for cond in ("scf_options.df_force_dense", 'scf_options.contraction_mode == "denseGPU"', "scf_options.df_use_adaptive",
                 "scf_data.μ < 800", "rank == 0", "n_ranks == 1"):
        assert cond in body, cond
    setup = GLUE[GLUE.index("function setup!"):GLUE.index("function df_rhf_fock_build_HIP!")]
    assert "use_dense_map(scf_data, scf_options, rank, n_ranks)" in setup
    # dense branch: the unscreened matrices of the reference, NULL pq lists with P = N^2, dense three-centre integrals reshaped
    dense_branch = setup[setup.index("if dense\n"):setup.index("else", setup.index("if dense\n"))]
    assert "setup_unscreened_screening_matricies(basis_sets, scf_data)" in dense_branch
    assert "P = N * N" in dense_branch and "C_NULL" in dense_branch
    assert "get_screening_metadata!" not in dense_branch
    assert re.search(r"calculate_three_center_integrals\([^)]*scf_data, 0, 1, false, false\)", setup.replace("\n", " "))
    assert '"dense hip"' in setup and '"screened hip"' in setup
    assert 'dense ? "dense hip" : "screened hip"' in setup
    # one host Fock matrix per rank and no host-side sum over devices any more
    op = GLUE[GLUE.index("function df_rhf_fock_build_HIP!"):]
    assert "axpy!" not in re.sub(r"#[^\n]*", "", op)
    assert "jcdf_group_fock_build" in op


def test_python_mirror_takes_the_same_dispatch_decision():
    import inspect
    from juliachem_jl_amd import df
    src = inspect.getsource(df.run_gpu_fock_build)
    for cond in ("df_force_dense", '"denseGPU"', "df_use_adaptive", "< 800", "n_ranks == 1"):
        assert cond in src, cond
