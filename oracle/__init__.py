"""CPU oracle for the Krylov-Schur hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in the product (`arnoldimethod.jl_amd/`) may import this package; only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`.
See `oracle/README.md`.
"""
