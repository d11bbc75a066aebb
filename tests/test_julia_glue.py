"""`-m "not gpu"`: static check of the Julia side of the boundary (arnoldimethod.jl_amd/julia/KrylovSchurHIP.jl) against
include/kschur.h -- no Julia runtime exists in the build image, so the glue cannot be executed; what CAN be verified
without one is verified here:

  * every `ccall((:name, LIB), ret, (argtypes...), ...)` names a function the header declares, with the same arity, the
    same return type and an argument of the same machine class (32/64-bit integer, double, pointer) in every position;
  * the Julia mirrors of the by-value structs (KsParams, KsHistory, KsExpandStats) have the header's fields in the
    header's order with the header's widths;
  * no ccall passes a Julia object as `Any` (the GC-unsafe form ADVICE r1 flagged);
  * the method set of the array-type seam that SURVEY.md section 8b enumerates is present;
  * (same machinery) the ctypes prototype table of _lib.py agrees with the header argument by argument.
"""
import os
import re

from __graft_entry__ import ROOT, import_package

pkg = import_package()
JL = os.path.join(ROOT, "arnoldimethod.jl_amd", "julia", "KrylovSchurHIP.jl")
HDR = os.path.join(ROOT, "include", "kschur.h")


# ------------------------------------------------------------------ header side
def _c_class(t: str) -> str:
    t = t.strip()
    if "*" in t or re.search(r"\bks_\w+_fn\b", t):
        return "ptr"
    t = re.sub(r"\bconst\b", "", t).strip()
    base = t.split()[0] if t else t
    return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "double": "f64", "void": "void"}[base]


def header_prototypes():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*(const\s+char\s*\*|int)\s+(ks_[A-Za-z0-9_]+)\s*\(([^;]*?)\)\s*;", txt, flags=re.M | re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        if args in ("void", ""):
            cls = []
        else:
            cls = []
            for a in args.split(","):
                a = a.strip()
                # drop the parameter name (last identifier) unless the declarator is a bare type
                mm = re.match(r"^(.*?[\s\*])([A-Za-z_]\w*)$", a)
                cls.append(_c_class(mm.group(1) if mm else a))
        protos[name] = ("ptr" if "char" in ret else "i32", cls)
    return protos


def header_structs():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\1\s*;", txt, flags=re.S):
        fields = []
        for f in m.group(2).split(";"):
            f = f.strip()
            if f:
                ty, nm = f.rsplit(None, 1)
                fields.append((nm, _c_class(ty)))
        out[m.group(1)] = fields
    return out


# ------------------------------------------------------------------ Julia side
_JL_CLASS = {"Cint": "i32", "Int32": "i32", "Int64": "i64", "UInt64": "u64", "Cdouble": "f64", "Float64": "f64", "Cstring": "ptr"}


def _jl_class(t: str) -> str:
    t = t.strip()
    if t.startswith("Ptr{") or t.startswith("Ref{"):
        return "ptr"
    return _JL_CLASS[t]  # KeyError == a type this checker does not know == fail loudly


def _split_top(s: str):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "{(":
            depth += 1
        elif ch in "})":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [p.strip() for p in parts if p.strip()]


def julia_ccalls():
    src = open(JL).read()
    src = "\n".join(ln.split("#=")[0] if "#=" in ln and "=#" not in ln else ln for ln in src.splitlines())
    src = re.sub(r"#=.*?=#", "", src)
    src = "\n".join(ln for ln in src.splitlines() if not ln.lstrip().startswith("#"))
    calls = []
    for m in re.finditer(r"ccall\(\(:(\w+),\s*LIB\),\s*(\w+),\s*\(", src):
        i = m.end()
        depth, j = 1, i
        while depth:
            ch = src[j]
            depth += ch in "({"
            depth -= ch in ")}"
            j += 1
        argt = src[i : j - 1]
        # the actual arguments follow up to the closing paren of ccall
        k, depth = j, 1
        while depth:
            ch = src[k]
            depth += ch in "([{"
            depth -= ch in ")]}"
            k += 1
        actual = _split_top(src[j : k - 1].lstrip(", \n"))
        calls.append((m.group(1), m.group(2), _split_top(argt), actual))
    return calls


def julia_structs():
    src = open(JL).read()
    out = {}
    for m in re.finditer(r"^struct (Ks\w+)\n(.*?)^end", src, flags=re.M | re.S):
        fields = []
        for f in re.split(r"[;\n]", m.group(2)):
            f = f.strip()
            if f and "::" in f:
                nm, ty = f.split("::")
                fields.append((nm.strip(), _jl_class(ty)))
        out[m.group(1)] = fields
    return out


# ------------------------------------------------------------------ tests
def test_every_ccall_matches_the_header():
    protos = header_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 25, "the glue lost its ccalls?"
    used = set()
    for name, ret, argt, actual in calls:
        assert name in protos, f"ccall to {name}, which include/kschur.h does not declare"
        used.add(name)
        want_ret, want_args = protos[name]
        assert _jl_class(ret) == want_ret, f"{name}: return type {ret}"
        assert len(argt) == len(want_args), f"{name}: {len(argt)} argument types, header has {len(want_args)}"
        assert len(actual) == len(want_args), f"{name}: {len(actual)} actual arguments for {len(want_args)} parameters ({actual})"
        for pos, (jt, cc) in enumerate(zip(argt, want_args)):
            assert jt != "Any", f"{name}: argument {pos + 1} passed as `Any`"
            assert _jl_class(jt) == cc, f"{name}: argument {pos + 1} is {jt} in Julia, class {cc} in the header"
    # the seam's verbs and the drivers are all bound
    for need in ("ks_ctx_create", "ks_ctx_destroy", "ks_operator_csr", "ks_operator_dense", "ks_operator_host_callback", "ks_operator_destroy",
                 "ks_operator_format", "ks_workspace_create", "ks_workspace_destroy", "ks_workspace_H", "ks_workspace_Q", "ks_cols_download",
                 "ks_col_fill_uniform", "ks_col_upload", "ks_col_copy", "ks_col_norm", "ks_col_div", "ks_apply", "ks_gemv_t",
                 "ks_gemv_n_sub", "ks_rotate", "ks_basis_times", "ks_iterate_arnoldi", "ks_partialschur", "ks_last_error_string"):
        assert need in used, f"{need} is not bound by the Julia glue"


def test_struct_mirrors_match_the_header():
    hs, js = header_structs(), julia_structs()
    for cname, jname in (("ks_params", "KsParams"), ("ks_history", "KsHistory"), ("ks_expand_stats", "KsExpandStats")):
        assert jname in js, f"{jname} missing in the Julia glue"
        assert js[jname] == hs[cname], f"{jname} != {cname}: {js[jname]} vs {hs[cname]}"


def test_array_type_seam_method_set_is_present():
    """the complete set of operations the reference applies to V / views of V."""
    src = open(JL).read()
    for pat, why in [
        (r"struct HipBasis\{T\} <: AbstractMatrix\{T\}", "the basis type"),
        (r"struct HipColumn\{T\} <: AbstractVector\{T\}", "view(V, :, j)"),
        (r"struct HipColumns\{T\} <: AbstractMatrix\{T\}", "view(V, :, a:b)"),
        (r"Base\.view\(V::HipBasis\{T\}, ::Colon, j::Integer\)", "view(V, :, j), expansion.jl:18,77"),
        (r"Base\.view\(V::HipBasis\{T\}, ::Colon, r::AbstractUnitRange", "view(V, :, 1:j), expansion.jl:34,76"),
        (r"Base\.size\(V::HipBasis\)", "size(V, 1), size(V, 2)"),
        (r"Base\.similar\(V::HipBasis\{T\}\)", "V_tmp = similar(V), ArnoldiMethod.jl:84"),
        (r"Random\.rand!\(v::HipColumn\)", "rand!(v), expansion.jl:15,21"),
        (r"Base\.copyto!\(v::HipColumn\{T\}, src::AbstractVector\)", "copyto!(v, v1), run.jl:126"),
        (r"LinearAlgebra\.norm\(v::HipColumn\)", "norm(v)"),
        (r"Base\.Broadcast\.materialize!\(dest::HipColumn\{T\}", "v ./= s"),
        (r"LinearAlgebra\.mul!\(y::HipColumn\{T\}, A::HipOperator\{T\}, x::HipColumn\{T\}\)", "mul!(w, A, v), expansion.jl:121"),
        (r"LinearAlgebra\.mul!\(h::AbstractVector, Va::Adjoint\{T,HipColumns\{T\}\}, v::HipColumn\{T\}\)", "mul!(h, Vprev', v)"),
        (r"Base\.:\*\(Va::Adjoint\{T,HipColumns\{T\}\}, v::HipColumn\{T\}\)", "Vprev' * v"),
        (r"LinearAlgebra\.mul!\(v::HipColumn\{T\}, Vp::HipColumns\{T\}, h::AbstractVector, α::Number, β::Number\)", "mul!(v, Vprev, h, -1, 1)"),
        (r"LinearAlgebra\.mul!\(dst::HipColumns\{T\}, src::HipColumns\{T\}, Qb::AbstractMatrix\)", "mul!(V_tmp, V, Q), run.jl:363,382"),
        (r"Base\.copyto!\(dst::HipColumns\{T\}, src::HipColumns\{T\}\)", "copyto!(V, V_tmp), run.jl:364,383"),
        (r"Base\.copyto!\(dst::HipColumn\{T\}, src::HipColumn\{T\}\)", "copyto!(V[:,k+1], V[:,maxdim+1]), run.jl:365"),
        (r"Base\.:\*\(Qv::HipColumns\{T\}, Y::AbstractMatrix\)", "P.Q * vecs, eigvals.jl:94"),
        (r"function ArnoldiMethod\.iterate_arnoldi!\(A::HipOperator\{T\}, arnoldi::ArnoldiWorkspace\{T,<:HipBasis\{T\}\}", "fused expansion"),
        (r"function ArnoldiMethod\.partialschur\(A::HipOperator\{T\}", "partialschur on a device operator"),
    ]:
        assert re.search(pat, src), f"missing method for {why}: /{pat}/"
    assert "unsafe_pointer_to_objref(user)[]" in src and "pointer_from_objref(box)" in src  # Ref{Any} box, not a tuple as `Any`


def test_ctypes_prototype_table_matches_header_argument_by_argument():
    import ctypes as C

    protos = header_prototypes()

    def cls(t):
        if t in (C.c_int, C.c_int32):
            return "i32"
        if t is C.c_int64:
            return "i64"
        if t is C.c_uint64:
            return "u64"
        if t is C.c_double:
            return "f64"
        return "ptr"  # c_void_p, POINTER(...), CFUNCTYPE(...)

    for name, args in pkg._lib.PROTOTYPES.items():
        want = protos[name][1]
        assert len(args) == len(want), f"{name}: ctypes table has {len(args)} arguments, header {len(want)}"
        for pos, (a, w) in enumerate(zip(args, want)):
            assert cls(a) == w, f"{name}: argument {pos + 1} is {a} in the ctypes table, class {w} in the header"
    # struct layouts
    hs = header_structs()
    for cname, ct in (("ks_params", pkg._lib.ks_params), ("ks_history", pkg._lib.ks_history), ("ks_expand_stats", pkg._lib.ks_expand_stats)):
        got = [(n, cls(t)) for n, t in ct._fields_]
        assert got == hs[cname], (cname, got, hs[cname])


# ------------------------------------------------------------------ enums: every code of the header has a name on both sides
def header_enums():
    """{first enumerator name: [(name, value), ...]} for every anonymous `enum { ... };` of the header."""
    txt = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"enum\s*\{(.*?)\}\s*;", txt, flags=re.S):
        items, nxt = [], 0
        for it in m.group(1).split(","):
            it = it.strip()
            if not it:
                continue
            if "=" in it:
                nm, v = (x.strip() for x in it.split("="))
                nxt = int(v, 0)
            else:
                nm = it
            items.append((nm, nxt))
            nxt += 1
        out[items[0][0]] = items
    return out


def _jl_tuple(name: str):
    src = open(JL).read()
    m = re.search(rf"^const\s+{name}\s*=\s*\((.*?)\)\s*$", src, flags=re.M)
    assert m, f"const {name} = (...) not found in the Julia glue"
    return [x.strip() for x in m.group(1).split(",") if x.strip()]


def test_layout_codes_have_a_name_on_every_side():
    """`operator_format` indexed a 6-tuple with 7 layout codes (BoundsError on config-3 matrices).  The
    Julia tuple, the Python table and the header's enum must have the same cardinality, codes dense from 0."""
    enums = header_enums()
    lay = enums["KS_LAYOUT_CSR"]
    assert [v for _, v in lay] == list(range(len(lay))), lay
    jl = _jl_tuple("LAYOUTS")
    assert len(jl) == len(lay), (jl, lay)
    assert not re.search(r"\(:csr, :csr_vi[^)]*\)\[", open(JL).read()), "a literal layout tuple is indexed somewhere: use LAYOUTS"
    py = pkg._lib.LAYOUTS
    assert sorted(k for k in py if k >= 0) == [v for _, v in lay], (py, lay)
    # the other small enums the glue mirrors as constants
    src = open(JL).read()
    for first, consts in (("KS_F64", ("KS_F64", "KS_C64")), ("KS_I32", ("KS_I32", "KS_I64")), ("KS_CSR", ("KS_CSR", "KS_CSC"))):
        items = dict(enums[first])
        assert len(items) == len(consts), (first, items)
        m = re.search(rf"^const\s+{', '.join(consts)}\s*=\s*(.*)$", src, flags=re.M)
        assert m, consts
        vals = [int(x) for x in re.findall(r"Cint\((\d+)\)", m.group(1))]
        assert vals == [items[c] for c in consts], (consts, vals, items)
    which = dict(enums["KS_LM"])
    jw = dict((k, int(v)) for k, v in re.findall(r":(\w+)\s*=>\s*Cint\((\d+)\)", re.search(r"^const WHICH = Dict\((.*)\)$", src, flags=re.M).group(1)))
    assert {f"KS_{k}": v for k, v in jw.items()} == which, (jw, which)
    assert {f"KS_{k}": v for k, v in pkg._lib.WHICH.items()} == which


def test_glue_vouches_for_the_factorisation_before_the_fused_expansion():
    """The implicit second pass is only taken on a factorisation the library trusts (include/kschur.h, PROVENANCE): the
    level-2 method must hand H over and call ks_workspace_assert_arnoldi before ks_iterate_arnoldi."""
    src = open(JL).read()
    body = src[src.index("function ArnoldiMethod.iterate_arnoldi!"):]
    body = body[: body.index("\nend\n")]
    a, b = body.find(":ks_workspace_assert_arnoldi"), body.find(":ks_iterate_arnoldi")
    assert 0 <= a < b, "assert_arnoldi must precede ks_iterate_arnoldi in iterate_arnoldi!"
    assert "three passes over V per step" not in src


def test_runtests_file_is_well_formed_and_only_uses_what_exists():
    """arnoldimethod.jl_amd/julia/runtests.jl (the reference's test sets on a device basis) cannot be executed here either: block
    structure balanced, every KrylovSchurHIP name it uses is exported by the module, every reference name it imports is one the
    reference's own tests import, and every test set cites the reference lines it replays; bench.py runs it when the box has Julia."""
    rt = open(os.path.join(ROOT, "arnoldimethod.jl_amd", "julia", "runtests.jl")).read()
    code = "\n".join(l.split("#")[0] for l in rt.splitlines())
    flat = code
    while re.search(r"\[[^\[\]]*\]", flat):     # `x[end]`, `[f(k) for k = 1:10]`: keywords inside brackets open / close nothing
        flat = re.sub(r"\[[^\[\]]*\]", "", flat)
    code, full = flat, code
    opens = len(re.findall(r"\b(begin|for|function|if|let|while|do|struct|module|try)\b", code))
    ends = len(re.findall(r"\bend\b", code))
    assert opens == ends, (opens, ends)
    jl = open(JL).read()
    exported = set(re.search(r"^export (.*)$", jl, flags=re.M).group(1).replace(" ", "").split(","))
    for name in ("HipContext", "HipOperator", "HipBasis", "hip_partialschur"):
        assert name in exported and re.search(r"\b%s\b" % name, rt), name
    used = set(re.findall(r"\b(Hip[A-Za-z]+|hip_[a-z_!]+)\b", full)) - {"hip_workspace"}
    assert used <= exported, used - exported
    imported = re.search(r"^using ArnoldiMethod: (.*)$", rt, flags=re.M).group(1).replace(" ", "").split(",")
    for name in imported:
        assert name in jl or name in ("partialeigen", "eigenvalues", "partialschur", "partialschur!"), name
    sets = re.findall(r"@testset \"([^\"]+)\"", rt)
    assert len(sets) >= 12
    cites = re.findall(r"# -+ (test/[a-z_]+\.jl:\d+-\d+|readme\.md:\d+-\d+)", rt)
    assert len(cites) >= 12, cites
    ref = "/root/reference"
    if os.path.isdir(ref):   # (only in the build container: the GPU box has no reference)
        for c in cites:
            f, rng = c.split(":")
            a, b = map(int, rng.split("-"))
            nl = len(open(os.path.join(ref, f)).read().splitlines())
            assert 1 <= a <= b <= nl, c
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert "runtests.jl" in bench and "julia_runtests" in bench
