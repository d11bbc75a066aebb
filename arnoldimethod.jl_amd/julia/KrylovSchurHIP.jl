# KrylovSchurHIP.jl -- the Julia side of the drop-in boundary (UNTESTED: no Julia runtime exists in the
# build image; every `ccall` below mirrors one prototype of include/kschur.h one-to-one and the same
# entry points are exercised from Python/ctypes by tests/test_gpu_parity.py).
#
# Two levels of integration with ArnoldiMethod.jl:
#
#   (1) whole-solver drop-in:   hip_partialschur(A; nev, which, tol, ...)  ->  (PartialSchur, History)
#       one ccall into ks_partialschur; Q stays in HBM and is downloaded lazily.
#
#   (2) array-type seam (src/ArnoldiMethod.jl:81-92): `HipBasis <: AbstractMatrix{T}` plugged into
#       `ArnoldiWorkspace(V, H; V_tmp, Q)`.  ArnoldiMethod's own `_partialschur` then runs unchanged and
#       only `iterate_arnoldi!` (src/expansion.jl:116-133) and the restart rotation (src/run.jl:363-365,
#       382-383) are specialised below to call the fused device paths.
#
# Usage sketch:
#     using ArnoldiMethod, SparseArrays
#     include("KrylovSchurHIP.jl"); using .KrylovSchurHIP
#     A = ...::SparseMatrixCSC{Float64,Int64}
#     decomp, history = hip_partialschur(A; nev = 20, which = :SR, tol = 1e-10)
#     λ, X = hip_partialeigen(decomp)
module KrylovSchurHIP

using LinearAlgebra, SparseArrays

export HipContext, HipOperator, HipWorkspace, hip_partialschur, hip_partialschur!, hip_partialeigen

const LIB = get(ENV, "KSCHUR_LIB", joinpath(@__DIR__, "..", "libkschur_hip.so"))

const KS_F64, KS_C64 = Cint(0), Cint(1)
const KS_I32, KS_I64 = Cint(0), Cint(1)
const KS_CSR, KS_CSC = Cint(0), Cint(1)
const WHICH = Dict(:LM => Cint(0), :LR => Cint(1), :SR => Cint(2), :LI => Cint(3), :SI => Cint(4))

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
mutable struct HipContext
    h::Ptr{Cvoid}
    function HipContext(device::Integer = 0)
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:ks_ctx_create, LIB), Cint, (Cint, Ref{Ptr{Cvoid}}), device, r))
        c = new(r[])
        finalizer(x -> ccall((:ks_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), c)
        c
    end
end

# ---------------------------------------------------------------------------------------------- operator
# mul!(y, A, x) for A::SparseMatrixCSC (stdlib SparseArrays; call site src/expansion.jl:121):
# colptr / rowval / nzval are handed over as they are (CSC, 1-based, Int64) and converted once to
# int32 0-based CSR in HBM by the library.
mutable struct HipOperator{T}
    h::Ptr{Cvoid}
    n::Int
    ctx::HipContext
end
Base.eltype(::HipOperator{T}) where {T} = T
Base.size(A::HipOperator) = (A.n, A.n)
Base.size(A::HipOperator, i::Integer) = i <= 2 ? A.n : 1

function HipOperator(ctx::HipContext, A::SparseMatrixCSC{T,Int64}) where {T<:Union{Float64,ComplexF64}}
    n = LinearAlgebra.checksquare(A)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve A begin
        check(ccall((:ks_operator_csr, LIB), Cint,
                    (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ref{Ptr{Cvoid}}),
                    ctx.h, n, n, nnz(A), pointer(A.colptr), pointer(A.rowval), pointer(A.nzval),
                    KS_CSC, 1, KS_I64, dtype_code(T), r))
    end
    op = HipOperator{T}(r[], n, ctx)
    finalizer(x -> ccall((:ks_operator_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), op)
    op
end

# mul!(y, A, x) for a dense A::Matrix (column-major; the library keeps a row-major copy in HBM and streams it
# once per product).
function HipOperator(ctx::HipContext, A::Matrix{T}) where {T<:Union{Float64,ComplexF64}}
    n = LinearAlgebra.checksquare(A)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve A begin
        check(ccall((:ks_operator_dense, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Int64, Cint, Cint, Ref{Ptr{Cvoid}}),
                    ctx.h, n, pointer(A), stride(A, 2), #=KS_COL_MAJOR=# 1, dtype_code(T), r))
    end
    op = HipOperator{T}(r[], n, ctx)
    finalizer(x -> ccall((:ks_operator_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), op)
    op
end

# Device layout the library chose for a stored matrix: (bytes streamed per non-zero, dictionary size).
function operator_format(A::HipOperator)
    b = Ref{Cdouble}(0); d = Ref{Cint}(0)
    check(ccall((:ks_operator_format, LIB), Cint, (Ptr{Cvoid}, Ref{Cdouble}, Ref{Cint}), A.h, b, d))
    (bytes_per_nnz = b[], ndict = Int(d[]))
end

# Opaque operators (LinearMaps etc., docs/src/index.md:246-249): the library calls back with two host
# pointers per application.  `@cfunction` is safe here: the call is synchronous on the calling task.
function _host_apply(user::Ptr{Cvoid}, x::Ptr{Cvoid}, y::Ptr{Cvoid})::Cint
    st = unsafe_pointer_to_objref(user)::Tuple
    A, T, n = st
    try
        mul!(unsafe_wrap(Array, Ptr{T}(y), n), A, unsafe_wrap(Array, Ptr{T}(x), n))
        return Cint(0)
    catch
        return Cint(1)
    end
end

function HipOperator(ctx::HipContext, A, ::Type{T}) where {T}
    n = size(A, 1)
    st = (A, T, n)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    cb = @cfunction(_host_apply, Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}))
    check(ccall((:ks_operator_host_callback, LIB), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Cvoid}, Any, Ref{Ptr{Cvoid}}),
                ctx.h, n, dtype_code(T), cb, st, r))
    op = HipOperator{T}(r[], n, ctx)
    # keep `st` alive as long as the operator
    finalizer(x -> (st; ccall((:ks_operator_destroy, LIB), Cint, (Ptr{Cvoid},), x.h)), op)
    op
end

# ---------------------------------------------------------------------------------------------- workspace
# ArnoldiWorkspace{T}: V in HBM, H and Q as Julia views of the library's pinned host arrays.
mutable struct HipWorkspace{T}
    h::Ptr{Cvoid}
    n::Int
    maxdim::Int
    H::Matrix{T}      # (maxdim+1) x maxdim, wraps library memory (do not resize)
    Q::Matrix{T}      # maxdim x maxdim
    ctx::HipContext
end

function HipWorkspace(ctx::HipContext, ::Type{T}, n::Integer, maxdim::Integer) where {T}
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ks_workspace_create, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Cint, Cint, Ref{Ptr{Cvoid}}),
                ctx.h, n, n, 0, maxdim, dtype_code(T), r))
    hp, ld = Ref{Ptr{Cvoid}}(C_NULL), Ref{Cint}(0)
    check(ccall((:ks_workspace_H, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Cint}), r[], hp, ld))
    H = unsafe_wrap(Array, Ptr{T}(hp[]), (maxdim + 1, maxdim))
    check(ccall((:ks_workspace_Q, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Cint}), r[], hp, ld))
    Q = unsafe_wrap(Array, Ptr{T}(hp[]), (maxdim, maxdim))
    w = HipWorkspace{T}(r[], n, maxdim, H, Q, ctx)
    finalizer(x -> ccall((:ks_workspace_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), w)
    w
end

"Array(view(V, :, j0+1:j0+ncols)) -- results are views of V (src/run.jl:375,389)"
function columns(w::HipWorkspace{T}, j0::Integer, ncols::Integer) where {T}
    out = Matrix{T}(undef, w.n, ncols)
    check(ccall((:ks_cols_download, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Int64), w.h, j0, ncols, out, w.n))
    out
end

# --- the verbs of src/expansion.jl on device columns (0-based column index j) ---
col_norm(w::HipWorkspace, j) = (r = Ref{Cdouble}(0); check(ccall((:ks_col_norm, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Cdouble}), w.h, j, r)); r[])
col_div!(w::HipWorkspace, j, s) = check(ccall((:ks_col_div, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble), w.h, j, s))
apply!(A::HipOperator, w::HipWorkspace, jsrc, jdst) = check(ccall((:ks_apply, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint), A.h, w.h, jsrc, jdst))
function gemv_t!(h::AbstractVector{T}, w::HipWorkspace{T}, j, jv) where {T}
    check(ccall((:ks_gemv_t, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}), w.h, j, jv, h))
    h
end
gemv_n_sub!(w::HipWorkspace{T}, j, jv, h::AbstractVector{T}) where {T} =
    check(ccall((:ks_gemv_n_sub, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}), w.h, j, jv, h))

"iterate_arnoldi!(A, arnoldi, from:to)  (src/expansion.jl:116-133), fused and asynchronous on the device"
function iterate_arnoldi!(A::HipOperator, w::HipWorkspace, range::UnitRange{Int})
    stats = zeros(Int32, 4)
    check(ccall((:ks_iterate_arnoldi, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Ptr{Int32}), A.h, w.h, first(range), last(range), stats))
    w
end

"V[:, purge:k] = V[:, purge:maxdim] * Q[purge:maxdim, purge:k]  (src/run.jl:363-364), in place, MFMA"
function rotate!(w::HipWorkspace{T}, purge::Int, k::Int, maxdim::Int) where {T}
    c, r = maxdim - purge + 1, k - purge + 1
    Qb = w.Q[purge:maxdim, purge:k]             # small host copy, column-major, ld = c
    check(ccall((:ks_rotate, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cvoid}, Cint), w.h, purge - 1, c, r, Qb, c))
end

# ---------------------------------------------------------------------------------------------- drivers
struct KsParams
    nev::Int32; which::Int32; tol::Float64; mindim::Int32; maxdim::Int32
    restarts::Int32; start_from::Int32; initialize::Int32; reserved::Int32
end
struct KsHistory
    mvproducts::Int32; nconverged::Int32; converged::Int32; nev::Int32
    restarts::Int32; reorth::Int32; breakdowns::Int32; reserved::Int32
    seconds_expand::Float64; seconds_host::Float64; seconds_rotate::Float64
end

struct HipPartialSchur{T}
    workspace::HipWorkspace{T}
    nconverged::Int
    R::SubArray                 # view(H, 1:nconv, 1:nconv): no copy (src/run.jl:149-150)
    eigenvalues::Vector{ComplexF64}
end
Base.getproperty(P::HipPartialSchur, s::Symbol) =
    s === :Q ? columns(getfield(P, :workspace), 0, getfield(P, :nconverged)) : getfield(P, s)

"partialschur!(A, arnoldi; start_from, initialize, nev, which, tol, mindim, maxdim, restarts)  src/run.jl:152-179"
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
    v1c = v1 === nothing ? C_NULL : convert(Vector{T}, v1)
    GC.@preserve v1c begin
        check(ccall((:ks_partialschur, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ref{KsParams}, Ptr{Cvoid}, Ptr{ComplexF64}, Ref{KsHistory}),
                    A.h, w.h, p, v1c === C_NULL ? C_NULL : pointer(v1c), eig, h))
    end
    hh = h[]
    nconv = Int(hh.nconverged)
    HipPartialSchur{T}(w, nconv, view(w.H, 1:nconv, 1:nconv), eig[1:nconv]),
    ArnoldiMethodHistory(hh.mvproducts, hh.nconverged, hh.converged != 0, hh.nev)
end

"History(mvproducts, nconverged, converged, nev)  src/run.jl:217-222"
struct ArnoldiMethodHistory
    mvproducts::Int; nconverged::Int; converged::Bool; nev::Int
end

"partialschur(A; v1, nev, which, tol, mindim, maxdim, restarts)  src/run.jl:100-129"
function hip_partialschur(A::SparseMatrixCSC{T}; ctx::HipContext = HipContext(), v1 = nothing,
                          nev::Int = min(6, size(A, 1)), maxdim::Int = min(max(20, 2nev), size(A, 1)), kw...) where {T}
    op = HipOperator(ctx, SparseMatrixCSC{T,Int64}(A))
    w = HipWorkspace(ctx, T, size(A, 1), maxdim)
    hip_partialschur!(op, w; nev = nev, maxdim = maxdim, v1 = v1, kw...)
end

"partialeigen(P)  src/eigvals.jl:92-95: eigen(R) on the host, the tall-skinny Q*vecs on the device"
function hip_partialeigen(P::HipPartialSchur{T}) where {T}
    vals, vecs = eigen(Matrix(P.R))
    Y = convert(Matrix{ComplexF64}, vecs)
    w = P.workspace
    out = Matrix{ComplexF64}(undef, w.n, size(Y, 2))
    check(ccall((:ks_basis_times, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Int64),
                w.h, P.nconverged, size(Y, 2), Y, size(Y, 1), KS_C64, out, w.n))
    vals, out
end

end # module
