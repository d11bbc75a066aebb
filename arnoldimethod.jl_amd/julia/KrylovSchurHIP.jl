# KrylovSchurHIP.jl -- the Julia side of the drop-in boundary.
#
# NOT EXECUTED in the build image (no Julia runtime exists there).  What IS checked without Julia: every
# `ccall((:name, LIB), ret, (argtypes...), ...)` below is parsed by tests/test_julia_glue.py and compared -- name,
# arity, every argument type, the KsParams / KsHistory / KsExpandStats field layouts -- against include/kschur.h, and
# the same entry points are exercised from Python/ctypes by the `-m gpu` tests.
#
# Three levels of integration with ArnoldiMethod.jl (v0.4.0), most faithful first:
#
#   (1) ARRAY-TYPE SEAM (src/ArnoldiMethod.jl:81-92, advertised use "custom array type for the basis", src/run.jl:142-143):
#       `HipBasis{T} <: AbstractMatrix{T}` lives in HBM.  `ArnoldiWorkspace(V::HipBasis, H)` is the reference's own
#       constructor, and the reference's own `partialschur!` / `_partialschur` (src/run.jl:152-392) run UNCHANGED:
#       every verb they apply to V or to a view of V is a method below that forwards to one C entry point
#           view(V, :, j) / view(V, :, a:b)              -> HipColumn / HipColumns        (no data movement)
#           rand!(v), copyto!(v, v1)                     -> ks_col_fill_uniform / ks_col_upload     expansion.jl:21, run.jl:121,126
#           norm(v)                                      -> ks_col_norm                             expansion.jl:24,41,48,81,88,96
#           v ./= s                                      -> ks_col_div                              expansion.jl:28,56,106
#           mul!(w, A, v)                                -> ks_apply                                expansion.jl:121
#           mul!(h, Vprev', v),  Vprev' * v              -> ks_gemv_t                               expansion.jl:37,46,84,93
#           mul!(v, Vprev, h, -1, 1)                     -> ks_gemv_n_sub                           expansion.jl:38,47,85,94
#           mul!(V_tmp[:, a:b], V[:, a:c], Q[a:c, a:b])  -> ks_rotate (in place; V_tmp is an alias)  run.jl:363,382
#           copyto!(V[:, a:b], V_tmp[:, a:b])            -> nothing (already in place)              run.jl:364,383
#           copyto!(V[:, k+1], V[:, maxdim+1])           -> ks_col_copy                             run.jl:365
#           P.Q * vecs                                   -> ks_basis_times                          eigvals.jl:94
#   (2) FUSED EXPANSION: `ArnoldiMethod.iterate_arnoldi!(A::HipOperator, arnoldi{<:HipBasis}, range)` replaces the
#       verb-by-verb loop of src/expansion.jl:116-133 by ONE call (ks_iterate_arnoldi: the whole range enqueued, DGKS
#       decisions on the device, TWO passes over V per step: the second DGKS projection is carried in a small triangular
#       factor).  Everything else of `_partialschur` still runs in Julia; the rotation verbs above fold the factor in.
#       The implicit second pass leans on the Arnoldi relation of the steps before `first(range)`; the reference only
#       ever calls iterate_arnoldi! on such a decomposition (src/run.jl:267,272), so the method vouches for it
#       (ks_workspace_assert_arnoldi) after handing the caller's H over -- without that the library would not trust a
#       basis the reference's own restart code rotated through the verbs and would run its explicit three-pass form.
#   (3) `ArnoldiMethod.partialschur(A::HipOperator; ...)`: builds the HipBasis workspace and calls the reference's
#       `partialschur!`; `hip_partialschur` is the whole solver as one C call (ks_partialschur), for comparison.
#
# Usage:
#     using ArnoldiMethod, SparseArrays
#     include("KrylovSchurHIP.jl"); using .KrylovSchurHIP
#     ctx = HipContext(0)
#     A   = HipOperator(ctx, sprand(10^6, 10^6, 5e-6) + I)       # SparseMatrixCSC{Float64,Int64} goes over as it is
#     decomp, history = partialschur(A; nev = 20, which = :SR, tol = 1e-10)   # ArnoldiMethod's own driver on HipBasis
#     λ, X = partialeigen(decomp)
module KrylovSchurHIP

using LinearAlgebra, SparseArrays, Random
import ArnoldiMethod
import ArnoldiMethod: ArnoldiWorkspace, PartialSchur

export HipContext, HipOperator, HipWorkspace, HipBasis, HipColumn, HipColumns, hip_partialschur, hip_partialschur!, hip_partialeigen, set_sstep!, relation_breaks

const LIB = get(ENV, "KSCHUR_LIB", joinpath(@__DIR__, "..", "libkschur_hip.so"))

const KS_F64, KS_C64 = Cint(0), Cint(1)
const KS_I32, KS_I64 = Cint(0), Cint(1)
const KS_CSR, KS_CSC = Cint(0), Cint(1)
const KS_ROW_MAJOR, KS_COL_MAJOR = Cint(0), Cint(1)
# KS_LAYOUT_* of include/kschur.h, in code order (tests/test_julia_glue.py compares the count with the header's enum)
const LAYOUTS = (:csr, :csr_vi, :csr_dvi, :sell, :sell_vi, :stencil, :csr_cb)
const WHICH = Dict(:LM => Cint(0), :LR => Cint(1), :SR => Cint(2), :LI => Cint(3), :SI => Cint(4))
const HipScalar = Union{Float64,ComplexF64}

dtype_code(::Type{Float64}) = KS_F64
dtype_code(::Type{ComplexF64}) = KS_C64

struct KsError <: Exception
    code::Cint
    msg::String
end

function check(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:ks_last_error_string, LIB), Cstring, ()))
    rc == 1 && throw(ArgumentError(msg))          # KS_ERR_ARGUMENT  <-> src/run.jl:111-116, ...
    rc == 2 && throw(DimensionMismatch(msg))      # KS_ERR_DIMENSION <-> checksquare, src/run.jl:110
    rc == 5 && throw(msg)                         # the reference throws a String (src/schurfact.jl:406)
    throw(KsError(rc, msg))
end

# ---------------------------------------------------------------------------------------------- context
# Children (operators, workspaces) hold a reference to their context and register themselves, so the context is
# destroyed only after every child handle: finalizer order is unspecified in Julia.
mutable struct HipContext
    h::Ptr{Cvoid}
    live::Int            # child handles still alive
    dead::Bool           # finalizer ran while children were alive: the last child destroys the context
    function HipContext(device::Integer = 0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ks_ctx_create, LIB), Cint, (Cint, Ref{Ptr{Cvoid}}), device, r))
        c = new(r[], 0, false)
        finalizer(_ctx_finalize, c)
        c
    end
end
function _ctx_destroy(c::HipContext)
    if c.h != C_NULL
        ccall((:ks_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), c.h)
        c.h = C_NULL
    end
    nothing
end
_ctx_finalize(c::HipContext) = c.live == 0 ? _ctx_destroy(c) : (c.dead = true; nothing)
_adopt(c::HipContext) = (c.live += 1; nothing)
function _release(c::HipContext)
    c.live -= 1
    (c.live == 0 && c.dead) && _ctx_destroy(c)
    nothing
end

# ---------------------------------------------------------------------------------------------- operator
# mul!(y, A, x) for A::SparseMatrixCSC (stdlib SparseArrays; call site src/expansion.jl:121):
# colptr / rowval / nzval are handed over as they are (CSC, 1-based, Int64 or Int32) and converted once into the
# device layout the library picks (ks_operator_format).
mutable struct HipOperator{T}
    h::Ptr{Cvoid}
    n::Int
    ctx::HipContext
    keep::Any            # what a callback operator must keep alive (Ref to the user's operator)
end
Base.eltype(::HipOperator{T}) where {T} = T
Base.eltype(::Type{HipOperator{T}}) where {T} = T
Base.size(A::HipOperator) = (A.n, A.n)
Base.size(A::HipOperator, i::Integer) = i <= 2 ? A.n : 1

function _finish_operator(::Type{T}, h::Ptr{Cvoid}, n::Int, ctx::HipContext, keep) where {T}
    op = HipOperator{T}(h, n, ctx, keep)
    _adopt(ctx)
    finalizer(op) do x
        if x.h != C_NULL
            ccall((:ks_operator_destroy, LIB), Cint, (Ptr{Cvoid},), x.h)
            x.h = C_NULL
            _release(x.ctx)
        end
    end
    op
end

index_code(::Type{Int64}) = KS_I64
index_code(::Type{Int32}) = KS_I32

function HipOperator(ctx::HipContext, A::SparseMatrixCSC{T,Ti}) where {T<:HipScalar,Ti<:Union{Int32,Int64}}
    n = LinearAlgebra.checksquare(A)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve A begin
        check(ccall((:ks_operator_csr, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ref{Ptr{Cvoid}}),
                    ctx.h, n, n, nnz(A), pointer(A.colptr), pointer(A.rowval), pointer(A.nzval),
                    KS_CSC, 1, index_code(Ti), dtype_code(T), r))
    end
    _finish_operator(T, r[], n, ctx, nothing)
end
# integer / Float32 / other element types are promoted like `vtype` does (src/run.jl:9-12)
HipOperator(ctx::HipContext, A::SparseMatrixCSC{Tv,Ti}) where {Tv<:Real,Ti} = HipOperator(ctx, SparseMatrixCSC{Float64,Int64}(A))
HipOperator(ctx::HipContext, A::SparseMatrixCSC{Tv,Ti}) where {Tv<:Complex,Ti} = HipOperator(ctx, SparseMatrixCSC{ComplexF64,Int64}(A))

# mul!(y, A, x) for a dense A::Matrix (column-major; the library keeps a row-major copy in HBM).
function HipOperator(ctx::HipContext, A::Matrix{T}) where {T<:HipScalar}
    n = LinearAlgebra.checksquare(A)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve A begin
        check(ccall((:ks_operator_dense, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Cint, Cint, Ref{Ptr{Cvoid}}),
                    ctx.h, n, pointer(A), stride(A, 2), KS_COL_MAJOR, dtype_code(T), r))
    end
    _finish_operator(T, r[], n, ctx, nothing)
end

# Shift-invert: mul!(y, A, x) with A = LinearMap((y, x) -> ldiv!(y, F, x)) around F = lu(M), M = A0 - sigma*I
# (docs/src/index.md:246-249).  The factorisation stays SuiteSparse's, on the host; its factors go to HBM once and every
# product is two sparse triangular solves ON THE DEVICE (ks_operator_lu) instead of a host ldiv! plus two PCIe copies.
# UMFPACK: (F.Rs .* M)[F.p, F.q] == F.L * F.U  ->  perm_in = p-1, scale = Rs, perm_out = q-1.  The library wants CSR
# factors: the CSC arrays of the TRANSPOSED factors are exactly that.
function HipOperator(ctx::HipContext, F::SparseArrays.UMFPACK.UmfpackLU{Tv}) where {Tv}
    T = Tv <: Complex ? ComplexF64 : Float64
    Lt = SparseMatrixCSC{T,Int64}(sparse(transpose(F.L)))   # CSC of L^T == CSR of L (columns ascending)
    Ut = SparseMatrixCSC{T,Int64}(sparse(transpose(F.U)))
    n = size(Lt, 1)
    lptr = Lt.colptr .- 1; lidx = Int32.(Lt.rowval .- 1)
    uptr = Ut.colptr .- 1; uidx = Int32.(Ut.rowval .- 1)
    pin = Int32.(F.p .- 1); pout = Int32.(F.q .- 1)
    rs = Vector{Float64}(F.Rs)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve Lt Ut lptr lidx uptr uidx pin pout rs begin
        check(ccall((:ks_operator_lu, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Cint, Ptr{Int64}, Ptr{Int32}, Ptr{Cvoid}, Ptr{Int64}, Ptr{Int32}, Ptr{Cvoid},
                     Ptr{Int32}, Ptr{Int32}, Ptr{Cdouble}, Ref{Ptr{Cvoid}}),
                    ctx.h, n, dtype_code(T), pointer(lptr), pointer(lidx), pointer(Lt.nzval), pointer(uptr), pointer(uidx),
                    pointer(Ut.nzval), pointer(pin), pointer(pout), pointer(rs), r))
    end
    _finish_operator(T, r[], n, ctx, nothing)
end

"Stored entries and dependency-chain lengths of the two triangular factors of a shift-invert operator."
function lu_info(A::HipOperator)
    v = [Ref{Int64}(0) for _ in 1:4]
    check(ccall((:ks_operator_lu_info, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Int64}, Ref{Int64}), A.h, v[1], v[2], v[3], v[4]))
    (nnz_l = v[1][], nnz_u = v[2][], levels_l = v[3][], levels_u = v[4][])
end

"Device layout the library chose for a stored matrix: (bytes streamed per non-zero, dictionary size, layout code)."
function operator_format(A::HipOperator)
    b = Ref{Cdouble}(0); d = Ref{Cint}(0); l = Ref{Cint}(0)
    check(ccall((:ks_operator_format, LIB), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ref{Cint}, Ref{Cint}), A.h, b, d, l))
    (bytes_per_nnz = b[], ndict = Int(d[]), layout = LAYOUTS[l[] + 1])
end

# Opaque operators (LinearMaps etc., docs/src/index.md:246-249): the library calls back with two host pointers per
# application; `user` is the address of a `Ref{Any}` that the HipOperator keeps alive (field `keep`) -- a mutable box,
# so the pointer is stable, unlike the address of an immutable tuple passed as `Any`.
struct _HostOp
    A::Any
    T::DataType
    n::Int
end
function _host_apply(user::Ptr{Cvoid}, x::Ptr{Cvoid}, y::Ptr{Cvoid})::Cint
    st = unsafe_pointer_to_objref(user)[]::_HostOp
    try
        _host_apply_typed(st.A, st.T, st.n, x, y)
        return Cint(0)
    catch
        return Cint(1)
    end
end
function _host_apply_typed(A, ::Type{T}, n::Int, x::Ptr{Cvoid}, y::Ptr{Cvoid}) where {T}
    mul!(unsafe_wrap(Array, Ptr{T}(y), n), A, unsafe_wrap(Array, Ptr{T}(x), n))
    nothing
end

function HipOperator(ctx::HipContext, A, ::Type{T}) where {T<:HipScalar}
    n = size(A, 1)
    box = Ref{Any}(_HostOp(A, T, n))
    r = Ref{Ptr{Cvoid}}(C_NULL)
    cb = @cfunction(_host_apply, Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}))
    GC.@preserve box begin
        check(ccall((:ks_operator_host_callback, LIB), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                    ctx.h, n, dtype_code(T), cb, pointer_from_objref(box), r))
    end
    _finish_operator(T, r[], n, ctx, box)   # `box` stays reachable through the operator for as long as the handle lives
end

# ---------------------------------------------------------------------------------------------- workspace handle
# The library's ArnoldiWorkspace: V in HBM; its own pinned host H and Q (used by the fused paths) wrapped without copies.
mutable struct HipWorkspace{T}
    h::Ptr{Cvoid}
    n::Int
    maxdim::Int
    H::Matrix{T}      # (maxdim+1) x maxdim, wraps library memory (do not resize)
    Q::Matrix{T}      # maxdim x maxdim
    ctx::HipContext
    # What the GLUE ITSELF knows about the factorisation (ADVICE r3): columns 1..relation_k+1 of V and columns
    # 1..relation_k of the caller's H form an Arnoldi / Krylov-Schur decomposition because this glue watched it being
    # built -- a start vector written into column 1, fused expansions, the reference's restart pair (src/run.jl:363-365).
    # Any other write to a column (upload, rand!, ./=, mul! into it, a Gram-Schmidt update) cuts it back.  Only then does
    # iterate_arnoldi! vouch for the decomposition (ks_workspace_assert_arnoldi); otherwise the library runs its
    # explicit three-pass expansion, which -- like the reference's iterate_arnoldi! -- reads no earlier column of H.
    # -1: nothing known.
    relation_k::Int
    pending_rot::Tuple{Int,Int}   # (m, k) of a rotation V[:, purge:k] <- V[:, purge:m] Q that still waits for its copyto!; (-1, -1): none
end

# column j (0-based) was written by something other than the expansion / restart pair
function _touched!(w::HipWorkspace, j::Integer)
    w.pending_rot = (-1, -1)
    w.relation_k = j == 0 ? 0 : min(w.relation_k, j - 1)   # (a lone first column is a 0-step decomposition)
    nothing
end

function HipWorkspace(ctx::HipContext, ::Type{T}, n::Integer, maxdim::Integer) where {T<:HipScalar}
    maxdim <= n || throw(ArgumentError("Krylov dimension should be less than matrix order."))  # src/ArnoldiMethod.jl:62-63
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ks_workspace_create, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Cint, Cint, Ref{Ptr{Cvoid}}),
                ctx.h, n, n, 0, maxdim, dtype_code(T), r))
    hp, ld = Ref{Ptr{Cvoid}}(C_NULL), Ref{Cint}(0)
    check(ccall((:ks_workspace_H, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Cint}), r[], hp, ld))
    H = unsafe_wrap(Array, Ptr{T}(hp[]), (maxdim + 1, maxdim))
    check(ccall((:ks_workspace_Q, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Cint}), r[], hp, ld))
    Q = unsafe_wrap(Array, Ptr{T}(hp[]), (maxdim, maxdim))
    w = HipWorkspace{T}(r[], n, maxdim, H, Q, ctx, -1, (-1, -1))
    _adopt(ctx)
    finalizer(w) do x
        if x.h != C_NULL
            ccall((:ks_workspace_destroy, LIB), Cint, (Ptr{Cvoid},), x.h)
            x.h = C_NULL
            _release(x.ctx)
        end
    end
    w
end

"""
    set_sstep!(w, s; pivot_min = NaN, gram_dev_max = NaN)

s-step (block) expansion for the fused `iterate_arnoldi!` / `partialschur` paths (include/kschur.h, ks_workspace_set_sstep):
`s >= 2` takes the steps of an expansion in blocks of up to `s` -- two passes over the basis per block instead of per step.
`0` switches it off; the library default is on (blocks of up to 20, `KS_SSTEP`).
"""
function set_sstep!(w::HipWorkspace, s::Integer; pivot_min::Float64 = NaN, gram_dev_max::Float64 = NaN)
    check(ccall((:ks_workspace_set_sstep, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cdouble), w.h, s, pivot_min, gram_dev_max))
    w
end

"""
    relation_breaks(w) -> (breaks, worst_leak)

Restarts of the library's drivers that cut through a 2 x 2 block of the real Schur form (imaginary-part targets on a real
operator: the members of a complex pair are not neighbours in the target's order, src/run.jl:298-339 / :363-365 then drop the
block's sub-diagonal entry and the Arnoldi relation of the kept columns is off by that much).  After the first one the
s-step expansion stays off for the run (include/kschur.h, ks_workspace_relation_info).  A caller that runs the restart
itself (`partialschur!` of the reference on a `HipBasis`) is covered as well: the expansion that follows `assert_arnoldi`
measures the relation of the last kept column before it runs in blocks (`relation_probes`), and a violation counts here.
"""
function relation_breaks(w::HipWorkspace)
    b = Ref{Cint}(0); l = Ref{Cdouble}(0.0)
    check(ccall((:ks_workspace_relation_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cdouble}), w.h, b, l))
    return (Int(b[]), l[])
end

"relation measurements taken on factorisations the caller vouched for (ks_workspace_relation_probes)"
function relation_probes(w::HipWorkspace)
    p = Ref{Cint}(0)
    check(ccall((:ks_workspace_relation_probes, LIB), Cint, (Ptr{Cvoid}, Ptr{Cint}), w.h, p))
    return Int(p[])
end

"""
    fused_rotations(w) -> (fused, chains_adopted, chains_dropped)

Restart rotations that ran inside the first sweep of the block expansion that followed them, and what became of the speculative
Newton chains (include/kschur.h, ks_workspace_fused_rotations).  Only the library's own restart drivers (`hip_partialschur`)
leave a rotation pending; a restart run by the reference's `partialschur!` on a `HipBasis` rotates through `mul!` at once.
"""
function fused_rotations(w::HipWorkspace)
    f = Ref{Cint}(0); a = Ref{Cint}(0); d = Ref{Cint}(0)
    check(ccall((:ks_workspace_fused_rotations, LIB), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}, Ptr{Cint}), w.h, f, a, d))
    return (Int(f[]), Int(a[]), Int(d[]))
end

"""
    deflated_blocks(w) -> (blocks, columns)

Blocks whose Newton chain was projected against locked Schur vectors of dominant eigenvalues step by step, and the number of
columns the last block batch deflated against (include/kschur.h, ks_workspace_deflated_blocks).
"""
function deflated_blocks(w::HipWorkspace)
    b = Ref{Cint}(0)
    c = Ref{Cint}(0)
    check(ccall((:ks_workspace_deflated_blocks, LIB), Cint, (Ptr{Cvoid}, Ptr{Cint}, Ptr{Cint}), w.h, b, c))
    return Int(b[]), Int(c[])
end

"""
    split_rotations(w) -> Int

Pending restart rotations that ran through the ordinary kernel in front of a first block reading its Newton chain from scratch
columns (ComplexF64, Float64 shapes without a fused kernel; include/kschur.h, ks_workspace_split_rotations).
"""
function split_rotations(w::HipWorkspace)
    c = Ref{Cint}(0)
    check(ccall((:ks_workspace_split_rotations, LIB), Cint, (Ptr{Cvoid}, Ptr{Cint}), w.h, c))
    return Int(c[])
end

"""
    sstep_partition(T, k0, count, smax) -> Vector{Int}

Block sizes the library uses for `count` steps of `iterate_arnoldi!` on top of `k0` columns (include/kschur.h, ks_sstep_partition).
"""
function sstep_partition(::Type{T}, k0::Integer, count::Integer, smax::Integer) where {T}
    out = Vector{Cint}(undef, 64); nb = Ref{Cint}(0)
    GC.@preserve out check(ccall((:ks_sstep_partition, LIB), Cint, (Cint, Cint, Cint, Cint, Ptr{Cint}, Cint, Ptr{Cint}), dtype_code(T), k0, count, smax, pointer(out), 64, nb))
    return Int.(out[1:min(Int(nb[]), 64)])
end

"Array(view(V, :, j0+1:j0+ncols)): host copy of device columns"
function columns(w::HipWorkspace{T}, j0::Integer, ncols::Integer) where {T}
    out = Matrix{T}(undef, w.n, ncols)
    ncols == 0 && return out
    GC.@preserve out check(ccall((:ks_cols_download, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Int64), w.h, j0, ncols, pointer(out), w.n))
    out
end

# ---------------------------------------------------------------------------------------------- (1) the array-type seam
"""
    HipBasis{T}(ctx, n, ncols)            # n x ncols basis in HBM; ncols = krylov_dimension + 1

`AbstractMatrix{T}` whose storage is the `V` of a library workspace.  Plug it into the reference's own constructor,
`ArnoldiWorkspace(V, zeros(T, ncols, ncols - 1))` (src/ArnoldiMethod.jl:81-92).  `similar(V)` -- what that constructor
uses for `V_tmp` -- returns an ALIAS of the same storage: the restart rotation runs in place on the device, a second
n-sized array is never allocated (DESIGN.md section 2).
"""
struct HipBasis{T} <: AbstractMatrix{T}
    ws::HipWorkspace{T}
    alias::Bool        # true: this is the V_tmp stand-in
end
HipBasis{T}(ctx::HipContext, n::Integer, ncols::Integer) where {T<:HipScalar} = HipBasis{T}(HipWorkspace(ctx, T, n, ncols - 1), false)
Base.size(V::HipBasis) = (V.ws.n, V.ws.maxdim + 1)
Base.similar(V::HipBasis{T}) where {T} = HipBasis{T}(V.ws, true)
# H and Q of `ArnoldiWorkspace(v1, k)` / defaults are `similar(V, k+1, k)`: small HOST matrices (src/ArnoldiMethod.jl:74-77)
Base.similar(V::HipBasis{T}, dims::Tuple{Int,Int}) where {T} = Matrix{T}(undef, dims)
Base.similar(V::HipBasis{T}, m::Int, n::Int) where {T} = Matrix{T}(undef, m, n)

"view(V, :, j): one column in HBM (0-based index `j`)"
struct HipColumn{T} <: AbstractVector{T}
    ws::HipWorkspace{T}
    j::Int
end
"view(V, :, a:b): consecutive columns in HBM, starting at 0-based column `j0`"
struct HipColumns{T} <: AbstractMatrix{T}
    ws::HipWorkspace{T}
    j0::Int
    ncols::Int
    alias::Bool
end
Base.size(v::HipColumn) = (v.ws.n,)
Base.size(V::HipColumns) = (V.ws.n, V.ncols)
Base.view(V::HipBasis{T}, ::Colon, j::Integer) where {T} = HipColumn{T}(V.ws, Int(j) - 1)
Base.view(V::HipBasis{T}, ::Colon, r::AbstractUnitRange{<:Integer}) where {T} = HipColumns{T}(V.ws, Int(first(r)) - 1, length(r), V.alias)
Base.view(V::HipColumns{T}, ::Colon, j::Integer) where {T} = HipColumn{T}(V.ws, V.j0 + Int(j) - 1)
Base.view(V::HipColumns{T}, ::Colon, r::AbstractUnitRange{<:Integer}) where {T} = HipColumns{T}(V.ws, V.j0 + Int(first(r)) - 1, length(r), V.alias)

# scalar indexing exists for `show` and debugging only (one PCIe round trip per element)
Base.getindex(v::HipColumn, i::Int) = Array(v)[i]
Base.getindex(V::HipColumns, i::Int, j::Int) = Array(view(V, :, j))[i]
Base.getindex(V::HipBasis, i::Int, j::Int) = Array(view(V, :, j))[i]
Base.Array(v::HipColumn) = vec(columns(v.ws, v.j, 1))
Base.Array(V::HipColumns) = columns(V.ws, V.j0, V.ncols)
Base.Array(V::HipBasis) = columns(V.ws, 0, V.ws.maxdim + 1)
Base.Matrix(V::HipColumns) = Array(V)
Base.Matrix(V::HipBasis) = Array(V)
Base.Vector(v::HipColumn) = Array(v)
Base.collect(V::Union{HipColumn,HipColumns,HipBasis}) = Array(V)

# populate!(v): rand!(v) (src/expansion.jl:15,21) and copyto!(v, v1) (src/run.jl:126)
function Random.rand!(rng::Random.AbstractRNG, v::HipColumn)
    check(ccall((:ks_col_fill_uniform, LIB), Cint, (Ptr{Cvoid}, Cint, UInt64), v.ws.h, v.j, rand(rng, UInt64)))
    _touched!(v.ws, v.j)
    v
end
Random.rand!(v::HipColumn) = Random.rand!(Random.default_rng(), v)
function Base.copyto!(v::HipColumn{T}, src::AbstractVector) where {T}
    length(src) == v.ws.n || throw(DimensionMismatch("source has length $(length(src)), the column has $(v.ws.n)"))
    buf = convert(Vector{T}, src)
    GC.@preserve buf check(ccall((:ks_col_upload, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), v.ws.h, v.j, pointer(buf)))
    _touched!(v.ws, v.j)
    v
end
# copyto!(view(V, :, k+1), view(V, :, maxdim+1))   src/run.jl:365
function Base.copyto!(dst::HipColumn{T}, src::HipColumn{T}) where {T}
    dst.ws === src.ws || return copyto!(dst, Array(src))
    check(ccall((:ks_col_copy, LIB), Cint, (Ptr{Cvoid}, Cint, Cint), dst.ws.h, dst.j, src.j))
    w = dst.ws
    if w.pending_rot == (src.j, dst.j)
        # second half of the reference's restart (src/run.jl:365): the residual direction moves next to the truncated
        # basis -- a Krylov-Schur decomposition of dst.j steps (the caller's host code transformed H alongside)
        w.pending_rot = (-1, -1)
        w.relation_k = dst.j
    else
        _touched!(w, dst.j)
    end
    dst
end
# copyto!(view(V, :, a:b), view(V_tmp, :, a:b))   src/run.jl:364,383: the rotation already happened in place
function Base.copyto!(dst::HipColumns{T}, src::HipColumns{T}) where {T}
    size(dst) == size(src) || throw(DimensionMismatch("column blocks differ in size"))
    if dst.ws === src.ws && dst.j0 == src.j0
        return dst
    end
    for c in 1:dst.ncols
        copyto!(view(dst, :, c), view(src, :, c))
    end
    dst
end

# norm(v)   src/expansion.jl:24,41,48,81,88,96
function LinearAlgebra.norm(v::HipColumn)
    r = Ref{Cdouble}(0)
    check(ccall((:ks_col_norm, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Cdouble}), v.ws.h, v.j, r))
    r[]
end
LinearAlgebra.norm(v::HipColumn, p::Real) = p == 2 ? norm(v) : norm(Array(v), p)

# v ./= s   src/expansion.jl:28,56,106  (lowers to materialize!(v, broadcasted(/, v, s)))
function Base.Broadcast.materialize!(dest::HipColumn{T}, bc::Base.Broadcast.Broadcasted{S,Ax,typeof(/),Tuple{HipColumn{T},N}}) where {T,S,Ax,N<:Number}
    src, s = bc.args
    (src.ws === dest.ws && src.j == dest.j) || throw(ArgumentError("only the in-place form v ./= s is supported on device columns"))
    check(ccall((:ks_col_div, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble), dest.ws.h, dest.j, Float64(real(s))))
    _touched!(dest.ws, dest.j)
    dest
end

# mul!(view(V, :, j+1), A, view(V, :, j))   src/expansion.jl:121
function LinearAlgebra.mul!(y::HipColumn{T}, A::HipOperator{T}, x::HipColumn{T}) where {T}
    y.ws === x.ws || throw(ArgumentError("source and destination columns must belong to the same basis"))
    check(ccall((:ks_apply, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint), A.h, y.ws.h, x.j, y.j))
    _touched!(y.ws, y.j)
    y
end
# any other operator with a host mul!: one column over PCIe each way (what the reference does for LinearMaps, but
# prefer HipOperator(ctx, A, T), which lets the library batch the transfers)
function _host_mul!(y::HipColumn{T}, A, x::HipColumn{T}) where {T}
    xh = Array(x)
    yh = similar(xh)
    mul!(yh, A, xh)
    copyto!(y, yh)
end
LinearAlgebra.mul!(y::HipColumn{T}, A::AbstractMatrix, x::HipColumn{T}) where {T} = _host_mul!(y, A, x)
LinearAlgebra.mul!(y::HipColumn{T}, A, x::HipColumn{T}) where {T} = _host_mul!(y, A, x)

# mul!(h, Vprev', v) with h a HOST vector (a view of H)   src/expansion.jl:46,84;   Vprev' * v   src/expansion.jl:37,93
function _gemv_t(Vp::HipColumns{T}, v::HipColumn{T}) where {T}
    Vp.j0 == 0 || throw(ArgumentError("the projected-out block must start at the first column"))
    Vp.ws === v.ws || throw(ArgumentError("Vprev and v must belong to the same basis"))
    h = Vector{T}(undef, Vp.ncols)
    Vp.ncols == 0 && return h
    GC.@preserve h check(ccall((:ks_gemv_t, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}), v.ws.h, Vp.ncols, v.j, pointer(h)))
    h
end
Base.:*(Va::Adjoint{T,HipColumns{T}}, v::HipColumn{T}) where {T} = _gemv_t(parent(Va), v)
function LinearAlgebra.mul!(h::AbstractVector, Va::Adjoint{T,HipColumns{T}}, v::HipColumn{T}) where {T}
    copyto!(h, _gemv_t(parent(Va), v))
    h
end

# mul!(v, Vprev, h, -one(T), one(T))   src/expansion.jl:38,47,85,94
function LinearAlgebra.mul!(v::HipColumn{T}, Vp::HipColumns{T}, h::AbstractVector, α::Number, β::Number) where {T}
    (α == -1 && β == 1) || throw(ArgumentError("device columns support mul!(v, Vprev, h, -1, 1) only (the Gram-Schmidt update)"))
    (Vp.j0 == 0 && Vp.ws === v.ws) || throw(ArgumentError("the projected-out block must start at the first column of the same basis"))
    Vp.ncols == 0 && return v
    hb = convert(Vector{T}, h)
    GC.@preserve hb check(ccall((:ks_gemv_n_sub, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}), v.ws.h, Vp.ncols, v.j, pointer(hb)))
    _touched!(v.ws, v.j)
    v
end

# mul!(view(V_tmp, :, a:b), view(V, :, a:c), view(Q, a:c, a:b))   src/run.jl:363, 382: in place on the device.
# The destination must be a block of the alias (V_tmp = similar(V)) starting at the same column as the source.
function LinearAlgebra.mul!(dst::HipColumns{T}, src::HipColumns{T}, Qb::AbstractMatrix) where {T}
    (dst.ws === src.ws && dst.j0 == src.j0) ||
        throw(ArgumentError("device rotation is in place: destination and source blocks must start at the same column of the same basis"))
    c, r = src.ncols, dst.ncols
    size(Qb) == (c, r) || throw(DimensionMismatch("Q block is $(size(Qb)), expected ($c, $r)"))
    (c == 0 || r == 0) && return dst
    Qh = Matrix{T}(Qb)                     # small host copy, column-major, leading dimension c
    GC.@preserve Qh check(ccall((:ks_rotate, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cvoid}, Cint), src.ws.h, src.j0, c, r, pointer(Qh), c))
    w = src.ws
    m, k = src.j0 + c, src.j0 + r
    if w.relation_k == m && r < c
        # first half of the reference's restart (src/run.jl:363-364) on a decomposition of m steps this glue saw being
        # built: it becomes one of k steps once column m moves to column k (copyto! above)
        w.pending_rot = (m, k)
        w.relation_k = src.j0 > 0 ? src.j0 - 1 : -1
    elseif w.relation_k >= m && r == c
        # the final rotation of the converged block (src/run.jl:382-383, Q unitary, r == c): what lies beyond it is gone
        w.pending_rot = (-1, -1)
        w.relation_k = src.j0 > 0 ? src.j0 - 1 : -1
    else
        w.pending_rot = (-1, -1)
        w.relation_k = min(w.relation_k, src.j0 - 1)
    end
    dst
end

# P.Q * vecs   src/eigvals.jl:94  (real basis x complex coefficients -> complex result)
function Base.:*(Qv::HipColumns{T}, Y::AbstractMatrix) where {T}
    Qv.j0 == 0 || throw(ArgumentError("the Schur basis starts at the first column"))
    size(Y, 1) == Qv.ncols || throw(DimensionMismatch("inner dimensions differ"))
    Ty = (T <: Complex || eltype(Y) <: Complex) ? ComplexF64 : Float64
    Yh = Matrix{Ty}(Y)
    out = Matrix{Ty}(undef, Qv.ws.n, size(Yh, 2))
    (Qv.ncols == 0 || size(Yh, 2) == 0) && return fill!(out, zero(Ty))
    GC.@preserve Yh out check(ccall((:ks_basis_times, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Int64),
                                    Qv.ws.h, Qv.ncols, size(Yh, 2), pointer(Yh), size(Yh, 1), dtype_code(Ty), pointer(out), Qv.ws.n))
    out
end

# ---------------------------------------------------------------------------------------------- (2) fused expansion
struct KsExpandStats
    steps::Int32; reorth::Int32; breakdowns::Int32; explicit_steps::Int32
end

"""
iterate_arnoldi!(A, arnoldi, from:to) (src/expansion.jl:116-133) as ONE device call when the operator and the basis
both live in HBM.  The Hessenberg columns the device produced are copied into `arnoldi.H` (the caller's host matrix).
"""
function ArnoldiMethod.iterate_arnoldi!(A::HipOperator{T}, arnoldi::ArnoldiWorkspace{T,<:HipBasis{T}}, range::UnitRange{Int}) where {T}
    isempty(range) && return arnoldi
    w = arnoldi.V.ws
    st = Ref(KsExpandStats(0, 0, 0, 0))
    H = arnoldi.H
    if H !== w.H && first(range) > 1
        # the library's expansion reads the CURRENT Hessenberg matrix (implicit second DGKS pass: g = H c); the reference's
        # restart code rewrote the leading block of the caller's H on the host -> hand it over first
        @views copyto!(w.H[:, 1:first(range)-1], H[:, 1:first(range)-1])
    end
    # V[:, 1:first(range)] and H[:, 1:first(range)-1] are an Arnoldi / Krylov-Schur decomposition of first(range)-1 steps
    # whenever the REFERENCE calls this (src/run.jl:267,272).  Vouch for it only when this glue saw that decomposition
    # being built (relation_k, above): a caller who wrote V by hand and expands from j > 1 -- legal with the reference,
    # whose iterate_arnoldi! reads no earlier column of H -- gets the library's explicit form instead of g = H c computed
    # from a relation that does not hold (ADVICE r3).
    vouched = w.relation_k >= first(range) - 1
    check(ccall((:ks_workspace_assert_arnoldi, LIB), Cint, (Ptr{Cvoid}, Cint), w.h, vouched ? first(range) - 1 : -1))
    check(ccall((:ks_iterate_arnoldi, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Ref{KsExpandStats}), A.h, w.h, first(range), last(range), st))
    w.pending_rot = (-1, -1)
    w.relation_k = vouched ? last(range) : min(w.relation_k, first(range) - 1)
    if H !== w.H
        for j in range
            @views copyto!(H[1:j+1, j], w.H[1:j+1, j])
        end
    end
    arnoldi
end

# ---------------------------------------------------------------------------------------------- (3) drivers
"""
    partialschur(A::HipOperator; v1, nev, which, tol, mindim, maxdim, restarts)

src/run.jl:100-129 with the workspace on the device: builds `ArnoldiWorkspace(HipBasis, H)` and runs the REFERENCE's
`partialschur!` on it.  `decomp.Q` is a `HipColumns` view of V in HBM (`Array(decomp.Q)` downloads it),
`decomp.R` a view of the host `H` -- no copies, as documented at src/run.jl:149-150.
"""
function ArnoldiMethod.partialschur(A::HipOperator{T}; v1::Union{AbstractVector,Nothing} = nothing,
                                    nev::Int = min(6, size(A, 1)), maxdim::Int = min(max(20, 2nev), size(A, 1)), kw...) where {T}
    n = size(A, 1)
    V = HipBasis{T}(A.ctx, n, maxdim + 1)
    arnoldi = ArnoldiWorkspace(V, zeros(T, maxdim + 1, maxdim))
    if v1 === nothing
        return ArnoldiMethod.partialschur!(A, arnoldi; nev = nev, maxdim = maxdim, kw...)
    end
    length(v1) == n || throw(ArgumentError("v1 should have the same dimension as A"))
    ArnoldiMethod.reinitialize!(arnoldi, 0, v -> copyto!(v, v1))       # src/run.jl:126
    ArnoldiMethod.partialschur!(A, arnoldi; initialize = false, nev = nev, maxdim = maxdim, kw...)
end

struct KsParams
    nev::Int32; which::Int32; tol::Float64; mindim::Int32; maxdim::Int32
    restarts::Int32; start_from::Int32; initialize::Int32; reserved::Int32
end
struct KsHistory
    mvproducts::Int32; nconverged::Int32; converged::Int32; nev::Int32
    restarts::Int32; reorth::Int32; breakdowns::Int32; explicit_steps::Int32
    seconds_expand::Float64; seconds_host::Float64; seconds_rotate::Float64
end

"partialschur!(A, arnoldi; ...) (src/run.jl:152-179) as ONE C call: the library's own driver (ks_partialschur)"
function hip_partialschur!(A::HipOperator{T}, w::HipWorkspace{T};
                           start_from::Int = 1, initialize::Bool = start_from == 1,
                           nev::Int = min(6, A.n), which::Symbol = :LM,
                           tol::Real = sqrt(eps(Float64)),
                           mindim::Int = min(max(10, nev), A.n, w.maxdim),
                           maxdim::Int = min(max(20, 2nev), A.n, w.maxdim),
                           restarts::Int = 200, v1::Union{Nothing,AbstractVector} = nothing) where {T}
    haskey(WHICH, which) || throw(ArgumentError("Unknown target: $which"))
    p = Ref(KsParams(nev, WHICH[which], tol, mindim, maxdim, restarts, start_from, initialize, 0))
    h = Ref(KsHistory(0, 0, 0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0))
    eig = zeros(ComplexF64, maxdim)
    v1c = v1 === nothing ? T[] : convert(Vector{T}, v1)
    GC.@preserve v1c eig begin
        check(ccall((:ks_partialschur, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ref{KsParams}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{KsHistory}),
                    A.h, w.h, p, v1 === nothing ? C_NULL : pointer(v1c), pointer(eig), h))
    end
    hh = h[]
    nconv = Int(hh.nconverged)
    V = HipBasis{T}(w, false)
    PartialSchur(view(V, :, 1:nconv), view(w.H, 1:nconv, 1:nconv), eig[1:nconv]),
    ArnoldiMethod.History(Int(hh.mvproducts), nconv, hh.converged != 0, Int(hh.nev))
end

"partialschur(A; ...) (src/run.jl:100-129) through the library's own driver"
function hip_partialschur(A::HipOperator{T}; v1 = nothing, nev::Int = min(6, A.n), maxdim::Int = min(max(20, 2nev), A.n), kw...) where {T}
    w = HipWorkspace(A.ctx, T, A.n, maxdim)
    hip_partialschur!(A, w; nev = nev, maxdim = maxdim, v1 = v1, kw...)
end
hip_partialschur(A::SparseMatrixCSC; ctx::HipContext = HipContext(), kw...) = hip_partialschur(HipOperator(ctx, A); kw...)

"partialeigen(P) (src/eigvals.jl:92-95): eigen(R) on the host, the tall-skinny Q*vecs on the device"
function hip_partialeigen(P::PartialSchur)
    vals, vecs = eigen(Matrix(P.R))
    vals, P.Q * vecs
end

end # module
