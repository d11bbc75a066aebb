# First contact with Julia in one line (from the repository root, on a box with a GPU, Julia >= 1.6 and ArnoldiMethod.jl v0.4):
#
#     julia --project=<env with ArnoldiMethod> arnoldimethod.jl_amd/julia/runtests.jl
#
# Replays the reference's own test sets -- test/expansion.jl:12-55 and test/partial_schur.jl:6-138 -- with the Krylov basis in HBM
# (`HipBasis`, the array-type seam of src/ArnoldiMethod.jl:81-92) and the operator on the device (`HipOperator`): the REFERENCE's
# `reinitialize!` / `iterate_arnoldi!` / `partialschur!` run unchanged, every verb they apply to the basis lands in one C entry
# point of libkschur_hip.so (KrylovSchurHIP.jl, the table at its top).  Same assertions, same tolerances as the reference's files;
# what the device path cannot express (BigFloat element types, `@inferred` on a host matrix) is left to the reference's own suite.
#
# NOT EXECUTED in the build image (no Julia runtime there: `bench.py` records `julia_on_box` and runs this file when one exists;
# tests/test_julia_glue.py parses every `ccall` of the module against include/kschur.h and checks that this file only uses
# names the module or the reference export).  The same test sets run from Python through the same C entry points:
# tests/test_gpu_parity.py (KAT-1..5) and tests/test_gpu_seam_replay.py (the verb-by-verb replay of `_partialschur`).
using Test, LinearAlgebra, SparseArrays, Random

using ArnoldiMethod
using ArnoldiMethod: partialschur, partialschur!, partialeigen, eigenvalues, ArnoldiWorkspace, reinitialize!, iterate_arnoldi!

include(joinpath(@__DIR__, "KrylovSchurHIP.jl"))
using .KrylovSchurHIP

const ctx = HipContext(0)

# an ArnoldiWorkspace whose basis lives in HBM (src/ArnoldiMethod.jl:81-92 with a custom array type)
hip_workspace(::Type{T}, n::Int, maxdim::Int) where {T} = ArnoldiWorkspace(HipBasis{T}(ctx, n, maxdim + 1), zeros(T, maxdim + 1, maxdim))

@testset "KrylovSchurHIP: the reference's tests on a device basis" begin

    # ------------------------------------------------------------------------------------------ test/expansion.jl:6-10
    @testset "Initialization" begin
        arnoldi = hip_workspace(Float64, 5, 3)
        reinitialize!(arnoldi)
        @test norm(view(arnoldi.V, :, 1)) ≈ 1
    end

    # ------------------------------------------------------------------------------------------ test/expansion.jl:12-34
    @testset "Arnoldi Factorization" begin
        n, max = 10, 6
        for T in (Float64, ComplexF64)
            Ah = sprand(T, n, n, 0.1) + I
            for A in (HipOperator(ctx, Ah), Ah)           # device operator (fused expansion) and host operator (verb by verb)
                arnoldi = hip_workspace(T, n, max)
                reinitialize!(arnoldi)
                H = arnoldi.H
                iterate_arnoldi!(A, arnoldi, 1:3)
                V = Array(arnoldi.V)
                @test Ah * V[:, 1:3] ≈ V[:, 1:4] * H[1:4, 1:3]
                @test norm(V[:, 1:4]' * V[:, 1:4] - I) < sqrt(eps(Float64)) / 100
                iterate_arnoldi!(A, arnoldi, 4:max)
                V = Array(arnoldi.V)
                @test Ah * V[:, 1:max] ≈ V * H
                @test norm(V' * V - I) < sqrt(eps(Float64)) / 100
            end
        end
    end

    # ------------------------------------------------------------------------------------------ test/expansion.jl:36-55
    @testset "Invariant subspace" begin
        A = [rand(4, 4) zeros(4, 4); zeros(4, 4) rand(4, 4)]
        vh = hip_workspace(Float64, 8, 5)
        e1 = zeros(8)
        e1[1] = 1.0
        copyto!(view(vh.V, :, 1), e1)
        iterate_arnoldi!(HipOperator(ctx, A), vh, 1:5)
        V = Array(vh.V)
        @test norm(V' * V - I) < sqrt(eps(Float64)) / 100
        @test iszero(vh.H[5, 4])
    end

    # ------------------------------------------------------------------------------------------ test/partial_schur.jl:6-28
    @testset "Zero eigenvalues & low-rank matrices" begin
        for T in (Float64, ComplexF64)
            A = rand(T, 10, 3)
            B = A * A'
            schur, history = partialschur(HipOperator(ctx, B), nev = 5, mindim = 5, maxdim = 7, tol = eps())
            Q = Array(schur.Q)
            @test history.converged
            @test history.mvproducts == 7
            @test norm(Q'Q - I) < 1000eps(Float64)
            @test norm(B * Q - Q * schur.R) < 1000eps(Float64)
            @test norm(diag(schur.R)[4:5]) < 1000eps(Float64)
        end
    end

    # ------------------------------------------------------------------------------------------ test/partial_schur.jl:48-53
    @testset "Find all eigenvalues of a small matrix" begin
        A = rand(3, 3)
        schur, history = partialschur(HipOperator(ctx, A))
        @test history.converged
        @test history.mvproducts == 3
    end

    # ------------------------------------------------------------------------------------------ test/partial_schur.jl:55-63
    @testset "Incorrect input" begin
        A = HipOperator(ctx, rand(6, 6))
        @test_throws ArgumentError partialschur(A, mindim = 5, maxdim = 3)
        @test_throws ArgumentError partialschur(A, nev = 5, mindim = 3)
        @test_throws ArgumentError partialschur(A, nev = 5, maxdim = 3)
        @test_throws ArgumentError partialschur(A, nev = 10)
    end

    # ------------------------------------------------------------------------------------------ test/partial_schur.jl:66-77
    @testset "Eigenvector as initial vector is not problematic" begin
        A = rand(30, 30)
        A += A'
        λs, X = eigen(Symmetric(A))
        λ, x = λs[end], X[:, end]
        decomp, history = partialschur(HipOperator(ctx, A), v1 = x, nev = 2, tol = 1e-8)
        Q = Array(decomp.Q)
        @test history.converged
        @test norm(A * Q - Q * decomp.R) < 1e-7
        @test abs(maximum(real(decomp.eigenvalues)) - λ) < 1e-7
    end

    # ------------------------------------------------------------------------------------------ test/partial_schur.jl:80-85
    @testset "Target non-dominant eigenvalues" begin
        A = Matrix(Diagonal([1:0.1:10; 50:53]))
        S, hist = partialschur(HipOperator(ctx, A), which = :SR)
        @test all(x -> real(x) ≤ 10, eigenvalues(S.R))
    end

    # ------------------------------------------------------------------------------------------ test/partial_schur.jl:87-106
    @testset "Repeated eigenvalues" begin
        A = Matrix(Diagonal([1:0.1:9; 9.97; 9.98; 9.99; 10.0; 10.0; 10.0]))
        schur, history = partialschur(HipOperator(ctx, A), nev = 5, maxdim = 20, tol = 1e-12)
        Q = Array(schur.Q)
        @test history.converged
        @test norm(Q'Q - I) < 100 * eps(Float64)
        @test norm(A * Q - Q * schur.R) < size(A, 1) * 1e-12
    end

    # ------------------------------------------------------------------------------------------ test/partial_schur.jl:108-120
    @testset "Zero matrix" begin
        for T in (Float64, ComplexF64)
            A = zeros(T, 5, 5)
            schur, history = partialschur(HipOperator(ctx, A))
            Q = Array(schur.Q)
            @test history.converged
            @test history.mvproducts == history.nconverged == 5
            @test norm(Q'Q - I) < 100 * eps(Float64)
            @test norm(A * Q - Q * schur.R) == 0
        end
    end

    # ------------------------------------------------------------------------------------------ test/partial_schur.jl:122-138
    @testset "Passing an initial Schur decomp" begin
        A = rand(100, 100)
        Ad = HipOperator(ctx, A)
        arnoldi = hip_workspace(Float64, 100, 20)
        F, history = partialschur!(Ad, arnoldi, nev = 3, tol = 1e-12)
        Q = Array(F.Q)
        @test history.converged
        @test history.nconverged in 3:4
        @test norm(A * Q - Q * F.R) < 1e-10
        F, history = partialschur!(Ad, arnoldi, nev = 5, start_from = history.nconverged + 1, tol = 1e-8)
        Q = Array(F.Q)
        @test history.converged
        @test history.nconverged in 5:6
        @test norm(A * Q - Q * F.R) < 1e-6
    end

    # ------------------------------------------------------------------------------------------ readme.md:24-49 (KAT-1) + the C driver
    @testset "README example; the library's own driver agrees with the reference's on the device basis" begin
        A = spdiagm(-1 => fill(-1.0, 99), 0 => fill(2.0, 100), 1 => fill(-1.0, 99))
        Ad = HipOperator(ctx, A)
        decomp, history = partialschur(Ad, nev = 10, tol = 1e-6, which = :SR)
        @test history.converged
        λ, X = partialeigen(decomp)
        exact = [2 - 2cos(k * π / 101) for k = 1:10]
        @test maximum(abs.(sort(real(λ))[1:10] .- exact)) < 1e-6
        decomp2, history2 = hip_partialschur(Ad, nev = 10, tol = 1e-6, which = :SR)
        @test history2.converged
        @test maximum(abs.(sort(real(decomp2.eigenvalues))[1:10] .- exact)) < 1e-6
    end
end
