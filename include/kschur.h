/*
 * kschur.h -- C ABI of libkschur_hip.so: the MI355X (gfx950) implementation of the
 * Krylov-Schur Arnoldi hot path of ArnoldiMethod.jl.
 *
 * This header is the drop-in boundary.  Every entry point names the reference
 * interface it replaces as  <file>:<line>  relative to the reference tree
 * (JuliaLinearAlgebra/ArnoldiMethod.jl v0.4.0).  The Julia-side binding a maintainer
 * would add is shown in INTEGRATION.md (one `ccall` per function below).
 *
 * Conventions
 *   - plain C: opaque handles, plain pointers and sizes, no C++/torch types;
 *   - every function returns an int status (KS_OK == 0); no exceptions cross the ABI;
 *     ks_last_error_string() describes the last failure on the calling thread;
 *   - matrices are COLUMN-MAJOR like Julia's `Matrix`; indices in THIS header are 0-based
 *     (column j here is column j+1 of the reference);
 *   - dtype: KS_F64 = Float64, KS_C64 = ComplexF64 (interleaved re,im); host pointers typed
 *     `void*` point at elements of the handle's dtype;
 *   - host pointers are borrowed for the duration of a call only (GC.@preserve suffices);
 *   - one host thread drives one context; calls are stream-ordered on the device and
 *     synchronous with respect to any host value they return;
 *   - multi-GPU: one process per GPU, rows of A and V are block-partitioned across ranks,
 *     the global sums inside the Gram-Schmidt steps are RCCL all-reduces, H/Q stay
 *     replicated on every host.
 */
#ifndef KSCHUR_H
#define KSCHUR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KS_VERSION_MAJOR 0
#define KS_VERSION_MINOR 1

/* ---- status codes -------------------------------------------------------------------- */
enum {
  KS_OK = 0,
  KS_ERR_ARGUMENT = 1,  /* Julia ArgumentError     (src/run.jl:111-116,123-124,165-174,185) */
  KS_ERR_DIMENSION = 2, /* Julia DimensionMismatch (checksquare, src/run.jl:110)            */
  KS_ERR_HIP = 3,       /* HIP runtime failure                                              */
  KS_ERR_RCCL = 4,      /* RCCL failure                                                     */
  KS_ERR_QR = 5,        /* "QR algorithm did not converge" (src/schurfact.jl:406)           */
  KS_ERR_INTERNAL = 6,
  KS_ERR_NO_DEVICE = 7, /* no gfx950 device visible: the product path has NO CPU fallback   */
  KS_ERR_OPERATOR = 8,  /* user operator callback reported failure                          */
  KS_ERR_COMM = 9       /* a peer did not show up within KS_P2P_TIMEOUT_S (peer-to-peer mode) */
};

enum { KS_F64 = 0, KS_C64 = 1 };                 /* element type of A, V, H, Q              */
enum { KS_I32 = 0, KS_I64 = 1 };                 /* index type of an uploaded sparse matrix */
enum { KS_CSR = 0, KS_CSC = 1 };                 /* layout of an uploaded sparse matrix     */
/* Targets, src/targets.jl:7-32 and _symbol_to_target, src/run.jl:181-185 */
enum { KS_LM = 0, KS_LR = 1, KS_SR = 2, KS_LI = 3, KS_SI = 4 };

typedef struct ks_ctx ks_ctx;             /* device + stream (+ RCCL communicator)           */
typedef struct ks_operator ks_operator;   /* anything with mul!(y, A, x)  (src/run.jl:21-22) */
typedef struct ks_workspace ks_workspace; /* ArnoldiWorkspace (src/ArnoldiMethod.jl:41-93)   */

const char* ks_last_error_string(void);
int ks_version(int* major, int* minor);

/* ---- context ------------------------------------------------------------------------- */
/* Single-GPU context on HIP device `device`. */
int ks_ctx_create(int device, ks_ctx** out);
/* Multi-GPU context: rank `rank` of `nranks`, one process per GPU.  `unique_id` is the 128-byte
 * RCCL id produced by ks_comm_unique_id() on rank 0 and broadcast by the host language
 * (torch.distributed / MPI / Distributed.jl).  No reference equivalent: the reference is a
 * single process. */
int ks_comm_unique_id(void* out128);
int ks_ctx_create_dist(int device, int rank, int nranks, const void* unique_id128, ks_ctx** out);
/* Multi-GPU context on the peer-to-peer transport (csrc/ks_p2p.hpp): the per-step reductions and the
 * ghost exchange of the SpMV run as remote stores into IPC-shared, uncached regions over xGMI instead of
 * RCCL calls (same results; sums are formed in rank order on every rank).  Protocol: every rank calls
 * ks_ctx_create_p2p, exports its 64-byte region handle with ks_ctx_p2p_handle, the host language
 * all-gathers the handles, and every rank passes the nranks x 64 bytes (rank order) to ks_ctx_p2p_attach.
 * Needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this image.  ks_ctx_create_dist with KS_TRANSPORT=p2p does the
 * same through one RCCL all-gather.  In this mode ks_operator_csr_dist is a collective call.
 * KS_P2P_CAP (2048 doubles per reduction), KS_P2P_ARENA_MB (64), KS_P2P_TIMEOUT_S (30) tune the region. */
int ks_ctx_create_p2p(int device, int rank, int nranks, ks_ctx** out);
int ks_ctx_p2p_handle(ks_ctx* ctx, void* out64);
int ks_ctx_p2p_attach(ks_ctx* ctx, const void* handles);
/* Multi-GPU context on a HOST-STAGED transport: the library runs exactly the launch structure of the RCCL
 * transport (reduce-only kernels -> all-reduce -> post kernels, src/expansion.jl:84,88,93,96 sharded by rows;
 * pack kernel -> neighbour exchange -> SpMV on the ghost buffer, src/expansion.jl:121), but every exchange is
 * copied to pinned host memory and handed to two caller-supplied functions, e.g. MPI_Allreduce /
 * MPI_Sendrecv or torch.distributed on gloo:
 *   allreduce(user, buf, count)            in-place sum over all ranks of `count` doubles; every rank must
 *                                          obtain the bit-identical result (the DGKS branches depend on it)
 *   exchange(user, npeers, peers, sendbufs, send_bytes, recvbufs, recv_bytes)
 *                                          one grouped neighbour exchange (send i -> peers[i], receive from it)
 * Both return 0 on success; anything else surfaces as KS_ERR_COMM.  No arithmetic happens on the host.  This is how
 * the code around every RCCL call site is exercised with several real ranks on a one-GPU box (RCCL refuses two
 * ranks per device), and a transport of last resort where RCCL is unavailable.  No reference equivalent. */
typedef int (*ks_host_allreduce_fn)(void* user, double* buf, int count);
typedef int (*ks_host_exchange_fn)(void* user, int npeers, const int32_t* peers, const void* const* sendbufs,
                                   const int64_t* send_bytes, void* const* recvbufs, const int64_t* recv_bytes);
int ks_ctx_create_hostcomm(int device, int rank, int nranks, ks_host_allreduce_fn allreduce,
                           ks_host_exchange_fn exchange, void* user, ks_ctx** out);
int ks_ctx_destroy(ks_ctx* ctx);
int ks_ctx_synchronize(ks_ctx* ctx);
int ks_ctx_rank(const ks_ctx* ctx, int* rank, int* nranks);
/* Raw hipStream_t of the context (so a caller can time with hipEvents on the right stream). */
int ks_ctx_stream(ks_ctx* ctx, void** hip_stream);

/* ---- operator seam:  mul!(y, A, x), eltype(A), size(A)   (src/expansion.jl:121) ------- */
/* (i) device-resident CSR operand.  Accepts CSR or CSC (Julia's SparseMatrixCSC: colptr /
 * rowval / nzval, 1-based Int64 -> layout=KS_CSC, index_base=1, index_type=KS_I64) and
 * converts once to int32 0-based CSR in HBM.  `nrows_local` x `ncols` is this rank's row block
 * (single GPU: the whole square matrix).  Replaces SparseArrays' `mul!` for
 * `SparseMatrixCSC` (stdlib; call site src/expansion.jl:121). */
int ks_operator_csr(ks_ctx* ctx, int64_t nrows_local, int64_t ncols, int64_t nnz,
                    const void* ptr, const void* idx, const void* val, int layout, int index_base,
                    int index_type, int dtype, ks_operator** out);
/* Distributed CSR: this rank owns global rows [row_begin, row_begin+nrows_local); column indices
 * are already LOCAL-EXTENDED: 0..nrows_local-1 address owned entries of x, nrows_local+g
 * addresses ghost slot g.  Ghost slots are filled before every product by the halo plan:
 * for each neighbour p: send x[send_idx[send_ptr[p] .. send_ptr[p+1])] to rank neigh[p] and
 * receive recv_cnt[p] values into consecutive ghost slots (in neighbour order). */
int ks_operator_csr_dist(ks_ctx* ctx, int64_t nrows_local, int64_t nghost, int64_t nnz,
                         const int64_t* rowptr, const int32_t* colidx, const void* val, int dtype,
                         int nneigh, const int32_t* neigh, const int64_t* send_ptr,
                         const int32_t* send_idx, const int64_t* recv_cnt, ks_operator** out);
/* (i') dense matrix (mul!(y, A::Matrix, x)): n x n elements of `dtype`, leading dimension `ld` elements,
 * KS_COL_MAJOR (Julia / Fortran) or KS_ROW_MAJOR (C / numpy).  Kept row-major in HBM; y = A*x streams it
 * once per product (8 or 16 bytes per entry).  Single-GPU contexts only. */
#define KS_ROW_MAJOR 0
#define KS_COL_MAJOR 1
int ks_operator_dense(ks_ctx* ctx, int64_t n, const void* a, int64_t ld, int layout, int dtype,
                      ks_operator** out);
/* (ii) opaque host operator (e.g. a LinearMap wrapping ldiv! with a host LU,
 * docs/src/index.md:246-249): `apply(user, x_host, y_host)` computes y = A*x on n_local
 * elements of `dtype`; return nonzero to signal failure.  The library stages the two columns
 * over PCIe. */
typedef int (*ks_host_apply_fn)(void* user, const void* x_host, void* y_host);
int ks_operator_host_callback(ks_ctx* ctx, int64_t n_local, int dtype, ks_host_apply_fn apply,
                              void* user, ks_operator** out);
/* (iii) opaque device operator: `apply(user, x_dev, y_dev, hip_stream)` must enqueue y = A*x on
 * the given stream (x_dev/y_dev are device pointers to n_local elements). */
typedef int (*ks_device_apply_fn)(void* user, const void* x_dev, void* y_dev, void* hip_stream);
int ks_operator_device_callback(ks_ctx* ctx, int64_t n_local, int dtype, ks_device_apply_fn apply,
                                void* user, ks_operator** out);
/* (iv) shift-invert from the caller's triangular factors:  y = P_out U^-1 L^-1 P_in (s o x), both sparse triangular
 * solves on the device, every vector resident in HBM.  Replaces the host `ldiv!(y, F, x)` inside the LinearMap a user of
 * the reference wraps around F = lu(A - sigma I) / factorize(A)  (docs/src/index.md:246-249, 273-287); the
 * factorisation itself stays the caller's, on the host, as there.
 *   L, U       n x n CSR, int64 row offsets, int32 0-based columns, values of `dtype`; L lower triangular (diagonal
 *              entries optional: unit where absent), U upper triangular with every diagonal entry, both non-singular
 *   perm_in    row i of the triangular system takes x[perm_in[i]]      (NULL: identity)
 *   scale      ... multiplied by the real scale[perm_in[i]]            (NULL: none; UMFPACK's row scaling `Rs`)
 *   perm_out   y[perm_out[i]] = entry i of U^-1 L^-1 (...)             (NULL: identity)
 * SuiteSparse UMFPACK (Julia `F = lu(A)`: (Rs .* A)[p, q] = L U): perm_in = p-1, scale = Rs, perm_out = q-1, factors
 * transposed from CSC.  SuperLU (scipy `splu`: Pr A Pc = L U): perm_in = inverse(perm_r), perm_out = inverse(perm_c).
 * One or two launches per factor (synchronisation-free solves: a row's wavefront waits for the entries it needs; independent
 * parts of the factor run on separate XCDs, dense triangles of narrow dependency levels are inverted at upload while their
 * inverses stay tame -- KS_LU_RUN_COND).  The cost is dependency chains and memory round trips, not bytes:
 * ks_operator_lu_info / ks_operator_lu_layout report levels, fill and the device layout -- choose a fill-reducing ordering
 * with a short, bushy elimination tree.  Products are deterministic (bit-identical when repeated).  A wait that exceeds
 * KS_LU_TIMEOUT_S (20) ends the kernel and surfaces ONCE as KS_ERR_OPERATOR at the next synchronisation point of the
 * context (the vectors of that product are garbage).  Single-GPU contexts only. */
int ks_operator_lu(ks_ctx* ctx, int64_t n, int dtype, const int64_t* l_rowptr, const int32_t* l_colind, const void* l_val,
                   const int64_t* u_rowptr, const int32_t* u_colind, const void* u_val, const int32_t* perm_in,
                   const int32_t* perm_out, const double* scale, ks_operator** out);
/* strictly triangular stored entries and dependency-chain lengths ("levels") of the two factors */
int ks_operator_lu_info(const ks_operator* op, int64_t* nnz_l, int64_t* nnz_u, int64_t* levels_l, int64_t* levels_u);
/* how one factor (upper = 0 / 1) is laid out for the device: rows of the system the kernel solves (n + one more per row
 * of an inverted dense run), rows in such runs, rows of the part next to the root of the elimination tree that runs on one
 * XCD, and the number of independent groups the rest was split into (0: one launch for everything) */
int ks_operator_lu_layout(const ks_operator* op, int upper, int64_t* rows, int64_t* run_rows, int64_t* top_rows, int* ngroups);
int ks_operator_destroy(ks_operator* op);
int ks_operator_size(const ks_operator* op, int64_t* n_local, int64_t* nnz, int* dtype);
/* Device layout chosen for a stored matrix at upload (mul!(y, A, x), src/expansion.jl:121; all layouts give bit-identical y):
 *   KS_LAYOUT_STENCIL  one BIT per dictionary slot and row: matrices whose (column - row, value) dictionary has <= 32 entries
 *                      and whose rows are sub-sequences of one ordering of them (constant-coefficient stencils with
 *                      truncated boundary rows); the dictionary travels in the kernel arguments
 *   KS_LAYOUT_DVI      one byte per non-zero into a <= 256-entry dictionary of (column - row, value) pairs (stencils)
 *   KS_LAYOUT_SELL     sliced ELLPACK, 64-row slices stored column-major (lane = row: coalesced loads and, for banded
 *                      matrices, coalesced gathers); taken when slicing pads the matrix by <= 15 % (uniform row lengths)
 *   KS_LAYOUT_CSR_CB   column-blocked CSR row blocks: matrices with scattered columns whose x does not fit one XCD's L2 (config 3)
 *                      are split into column blocks, one launch each, the row sums continued across the launches in CSR order
 *   KS_LAYOUT_CSR      row blocks of <= 256 rows / <= 4096 non-zeros streamed non-zero-parallel through LDS; a longer
 *                      row is a block of its own (ragged and skewed matrices)
 *   ..._VI             the same storage orders with ONE 32-bit word per non-zero (dictionary index << 24 | column):
 *                      matrices with <= 256 distinct stored values and < 2^24 columns
 * KS_SPMV_FORMAT = stencil | dvi | sell | sellvi | csr | vi pins one; KS_SELL_SIGMA sorts rows by length inside windows.
 * Non-zero offsets are 32-bit, 64-bit once nnz >= 2^31 (KS_SPMV_PTR64=1 forces them).
 * *bytes_per_nnz = what the SpMV streams per stored non-zero (padding included: 12 / 20 for plain CSR Float64 /
 * ComplexF64, 4 value-indexed, 1 delta-value-indexed), *ndict the dictionary size, *layout one of the codes below
 * (0 / 0 / -1 for dense and callback operators). */
enum { KS_LAYOUT_CSR = 0, KS_LAYOUT_CSR_VI = 1, KS_LAYOUT_DVI = 2, KS_LAYOUT_SELL = 3, KS_LAYOUT_SELL_VI = 4, KS_LAYOUT_STENCIL = 5, KS_LAYOUT_CSR_CB = 6 };
int ks_operator_format(const ks_operator* op, double* bytes_per_nnz, int* ndict, int* layout);
/* y = A*x on raw device pointers (bench / tests; the solver uses ks_apply below). */
int ks_operator_apply_raw(ks_operator* op, const void* x_dev, void* y_dev);

/* ---- workspace: ArnoldiWorkspace{T}(V, H, V_tmp, Q)  (src/ArnoldiMethod.jl:41-93) ------ */
/* V: n_local x (maxdim+1) in HBM (column-major, leading dimension padded, pad rows zero);
 * H: (maxdim+1) x maxdim on the host, zero-initialised (src/ArnoldiMethod.jl:66,76);
 * Q: maxdim x maxdim on the host.  V_tmp of the reference is not materialised: the restart
 * rotation runs in place (scratch is allocated only for shapes the in-place kernel does not
 * cover).  `n_global` is the order of A (== n_local on one GPU); `row_begin` this rank's
 * first global row (feeds the counter-based RNG so random vectors are partition-independent). */
int ks_workspace_create(ks_ctx* ctx, int64_t n_local, int64_t n_global, int64_t row_begin,
                        int maxdim, int dtype, ks_workspace** out);
int ks_workspace_destroy(ks_workspace* ws);
/* What the placement search of ks_workspace_create did: candidates timed (0: basis too small or
 * KS_PLACE_TRIALS=1, search skipped), the calibration time of the kept / the slowest candidate, and how many
 * allocations the device refused.  Round 6: ON by default (up to 10 candidates, ending with the first one in the fast cluster) for
 * a Float64 workspace of at least KS_PLACE_MIN_MB (1024) MB on one GPU that runs the block expansion -- the second pass of a large
 * block is 12 % faster on some allocations of the basis than on others (bimodal, decided by the physical pages; profiles/
 * r06_bupdate_lottery.txt); off otherwise (KS_PLACE_TRIALS >= 2 forces it).  The search holds its candidates while it runs: never
 * more than KS_PLACE_MAX_X (default 10) times the basis size at once, never more than half of the free device memory, at most
 * KS_PLACE_BUDGET_MS (1500) milliseconds; every loser is freed before ks_workspace_create returns. */
int ks_workspace_placement(const ks_workspace* ws, int* candidates, double* best_ms, double* worst_ms, int* refused);
/* How many times the fused expansion reads the basis per step on this workspace: 2 = implicit second pass (default:
 * the DGKS second projection, src/expansion.jl:93-94, is carried in a small triangular factor instead of being applied
 * to the n-vector), 3 = the second projection is applied to the vector as the reference does (KS_PASSES=3 at creation). */
int ks_workspace_passes(const ks_workspace* ws, int* passes);
/* Per-workspace switch for the above (instead of the KS_PASSES environment variable read at creation): passes = 2 or 3.
 * `max_ratio`: the largest ||c|| / beta (second-pass correction against what is left of the vector,
 * src/expansion.jl:93-96) the implicit form carries; a step whose correction is larger -- a genuine one, as opposed to
 * the rounding-level corrections the DGKS test asks for at every step of a diagonally dominant operator -- is redone
 * with the correction applied to the vector (counted in ks_expand_stats.explicit_steps / ks_history.explicit_steps).
 * Default 1e-3 (KS_IMPLICIT_MAX_RATIO at creation); <= 0 removes the limit; NaN keeps the current value. */
int ks_workspace_set_passes(ks_workspace* ws, int passes, double max_ratio);
/* S-STEP (block) EXPANSION -- a faster form of iterate_arnoldi!(A, arnoldi, from:to), src/expansion.jl:116-133, for
 * device-resident operators on one GPU.  s >= 2: the steps of a range are taken in blocks of up to s (instantiated sizes:
 * 1-20 for Float64, 1-10 for ComplexF64; beyond 5 on the FP64 matrix instruction, whose kernels take the block size at run
 * time -- ks_sstep_partition says how a range of steps is cut into blocks): s operator products build a Newton basis (shifts = Leja-ordered Ritz values
 * of the previous restart, so the first expansion of a run still goes step by step), then TWO passes over the basis
 * orthogonalise the whole block (block classical Gram-Schmidt with Pythagorean inner products, applied twice, both times
 * carried in the triangular factor of the implicit second pass) and the s Hessenberg columns follow from the basis
 * recurrence -- per block what the default expansion does per step.  Same Krylov space, hence the same H, V and Ritz values
 * up to rounding (H to ~1e-12 relative on the reference's test matrices, tests/test_sstep_model.py); the DGKS decisions of
 * src/expansion.jl:91 are not taken (the second projection is always applied).  A block whose Gram matrix has a Cholesky
 * pivot below pivot_min times its diagonal entry (breakdown, src/expansion.jl:99-102, or an ill-conditioned basis) is
 * abandoned before anything is committed and its steps are redone one at a time, which takes the reference's decisions.
 * Like the implicit second pass it needs the library's provenance of the factorisation; otherwise, and for host-callback
 * operators and maxdim > 64 the expansion runs step by step.  Default s = 20 (KS_SSTEP at creation; blocks of up to 20 on
 * up to 24 existing columns, up to 16 on up to 28, up to 12 on up to 48, up to 8 on up to 64; ComplexF64: up to 10 on up to 32 columns,
 * up to 8 on up to 48, single steps beyond); s = 0 / 1: off --
 * every step then takes the reference's DGKS decisions.  A block is also abandoned when the Gram matrix of what its first
 * stage wrote differs from I by more than
 * gram_dev_max in any entry (~ eps cond^2 of the Newton basis; the recovered H carries errors ~ eps cond): default 1e-8
 * keeps H at the per-step path's accuracy.  After an abandoned block the library lowers the block size for the following
 * expansions (s -> s/2 -> 2 -> off) and probes a larger one again after 16 clean batches: spectra the shifts cover badly
 * (a complex disc, with real shifts) simply run with small blocks.  pivot_min / gram_dev_max: NaN keeps the current value
 * (defaults 1e-6 / 1e-8; KS_SSTEP_PIVOT_MIN / KS_SSTEP_GDEV_MAX at creation).
 * ks_workspace_sstep_info: *s = block size in force; blocks completed / abandoned since creation; diag3 = of the last
 * batch { smallest pivot ratio of the first stage, of the second stage, largest |entry| of (Gram matrix of the written
 * block - I) }. */
int ks_workspace_set_sstep(ks_workspace* ws, int s, double pivot_min, double gram_dev_max);
int ks_workspace_sstep_info(const ks_workspace* ws, int* s, int* blocks, int* abandoned, double* diag3);
/* How iterate_arnoldi!(A, arnoldi, from:to) (src/expansion.jl:116-133) is cut into blocks: `count` steps on top of k0 existing
 * columns with block sizes <= smax, for dtype KS_F64 / KS_C64 -- the sizes the library itself would use (they depend on which
 * kernel forms KS_BLK_MFMA left on), for byte accounting in benchmarks.  Writes at most cap sizes to out and returns how many
 * blocks there are in *nblocks (0: the range cannot run in blocks).  The sizes may cover only a PREFIX of the range: where the
 * kernels stop (ComplexF64 beyond 48 columns) the remaining steps run one at a time. */
int ks_sstep_partition(int dtype, int k0, int count, int smax, int* out, int cap, int* nblocks);
/* Restarts of the library's drivers (ks_partialschur, ks_expand_restart, ks_restart) whose selection cut through a 2 x 2 block
 * of the real Schur form: the members of a complex pair are not neighbours in the target's order (imaginary-part targets on a
 * real matrix; src/run.jl:298-339 keeps pairs together only when they are), the truncation of src/run.jl:363-365 then drops the
 * block's sub-diagonal entry and the Arnoldi relation of the kept columns is off by that much from there on -- in the reference
 * as well.  The per-step expansion is indifferent to it; the s-step expansion (which leans on the relation of the earlier
 * columns) is switched off for the rest of the run when the dropped entry exceeds 1e-12 ||H||_F.  *breaks = such restarts
 * since creation, *worst_leak = largest dropped entry / ||H||_F.
 * Counted here as well: a relation that DRIFTS under the block expansion (a non-normal operator with a large cluster at the
 * wanted end: each block expresses A q_j through the relation of the earlier columns, and there an error of it grows from cycle
 * to cycle).  Block runs of the library's drivers measure the relation of the last kept column every first or second restart
 * cycle (one operator product + a row sample behind the speculative chain; ks_workspace_relation_probes counts them); above
 * max(1e-10, 30 tol) ||H||_F the blocks go off for the rest of the run. */
int ks_workspace_relation_info(const ks_workspace* ws, int* breaks, double* worst_leak);
/* A caller that runs the restart itself (the reference's _partialschur on a device basis, src/run.jl:298-365) and then vouches
 * for the result (ks_workspace_assert_arnoldi) is not seen by the guard above.  For such a factorisation the library MEASURES
 * the relation before blocks lean on it: the residual of the last kept column, A v_c - V H[:, c] (one operator product into a
 * dead column + a strided row sample), at the start of the ks_iterate_arnoldi that follows; more than 1e-11 ||H||_F (a measured residual carries the rounding of
 * the column's history, up to ~1e-12 behind large blocks; a cut block shows at 1e-9 .. 1e-5) counts as a
 * break exactly like a leaking restart of the library's own drivers (blocks off for the run).  *probes = measurements taken
 * since creation. */
int ks_workspace_relation_probes(const ks_workspace* ws, int* probes);
/* Restart rotations (src/run.jl:363-365) that ran fused with the first pass of the block expansion that followed them
 * (k_brotdots_mfma: the rotation of a library-run restart stays pending until the next expansion is enqueued; any other reader of
 * the basis flushes it through the ordinary rotation kernel first; KS_ROT_DEFER=0 at workspace creation switches the deferral
 * off), and what became of the speculative Newton chains (the first products of the next expansion, enqueued behind the previous
 * one so that the device works while the host runs the restart step, src/run.jl:278-360; KS_SPEC_CHAIN=0 switches them off):
 * *spec_adopted = chains the next expansion took over, *spec_dropped = chains that were void by then (the restart did not leave
 * its rotation pending, or something else touched the basis in between).  Any pointer may be null. */
int ks_workspace_fused_rotations(const ks_workspace* ws, int* count, int* spec_adopted, int* spec_dropped);
/* Memory: the fused rotation and the speculative chain keep the Newton chain of a block in scratch columns of their own -- 20
 * (Float64) / 10 (ComplexF64) columns of the workspace's leading dimension, allocated at the first restart that leaves its rotation
 * pending (1.6 GB at n = 1e7 next to a 3.3-GB basis of 41 columns); the drift watch keeps one more column, KS_TRUE_START=1 (off by
 * default) another one, the in-chain deflation 131 KB of partial sums.  When the device has no
 * room for them the library falls back to the paths that need none (rotation at once, no speculation, no watch) instead of failing:
 * a workspace that fits the device without these features runs with them switched off. */
/* Pending restart rotations (src/run.jl:363-365) for whose shape or element type there is no fused kernel (ComplexF64; Float64
 * shapes outside the instantiated ones): the ordinary rotation kernel runs when the next expansion is enqueued and BOTH passes of
 * its first block read the Newton chain from scratch columns -- what this buys is that the speculative chain (above) can run
 * behind the previous expansion for these shapes too.  *count = such rotations since creation. */
int ks_workspace_split_rotations(const ks_workspace* ws, int* count);
/* In-chain deflation of the block expansion (round 6).  The reference orthogonalises every new Krylov vector at once
 * (src/expansion.jl:69-109); a block of s steps does so only at its end.  Where LOCKED Schur vectors (src/run.jl:330) belong to
 * eigenvalues that dominate the rest of the spectrum (:LM problems with outliers: test/partial_schur.jl:122-138) the Newton chain in
 * between is projected against those columns step by step -- two small launches per product -- and no shift is placed at their
 * eigenvalues; without it such problems abandon their blocks and run step by step.  Taken: the leading locked columns whose
 * eigenvalue exceeds the largest other Ritz value by a factor r > KS_DEFLATE_RATIO (1.5) with r^(steps of the block - 1) > 1e3, at
 * most 16; several ranks all-reduce the dot products (one more small collective per product); KS_CHAIN_DEFLATE=0 at
 * workspace creation switches it off.  *blocks = blocks whose chain was deflated since creation, *columns = columns the last block
 * batch deflated against.  Any pointer may be null. */
int ks_workspace_deflated_blocks(const ks_workspace* ws, int* blocks, int* columns);
/* Diagnostics (tools/blk_bench.py): average duration of `reps` back-to-back launches of one streaming kernel of the s-step
 * expansion at basis size k and block size s (which = 0: first pass, 1: second pass), timed with HIP events on the
 * library's stream; *grid = workgroups launched.  dbg: probe flags of the kernels (1: second pass without its stores).
 * Overwrites columns k .. k+s-1 and drops the provenance of the factorisation. */
int ks_debug_blk_time(ks_workspace* ws, int k, int s, int which, int reps, int dbg, double* ms_per_launch, int* grid);
/* PROVENANCE.  The implicit second pass reads the columns < from of the host H and relies on the Arnoldi relation
 * A V[:, 0:from-1) = V[:, 0:from) H[0:from, 0:from-1) holding for them -- the reference's iterate_arnoldi! reads
 * neither.  The library therefore takes the implicit form only for a factorisation it produced itself: after
 * ks_reinitialize(ws, 0, ...) and any sequence of ks_iterate_arnoldi / ks_restart / ks_expand_restart, inside
 * ks_partialschur, and only while the host H still equals, bit for bit, what the library last left there.  Any verb
 * that writes to V (ks_col_upload, ks_col_div, ks_gemv_n_sub, ks_rotate, ks_col_copy, ks_apply, ks_orthogonalize,
 * ks_col_fill_uniform, handing out ks_workspace_col_ptr) or a changed / caller-built H makes ks_iterate_arnoldi run the
 * explicit three-pass form instead (same results as the reference's op sequence, never wrong, about 1.5x slower).
 * A host language that runs the restart itself (the reference's own `_partialschur` on a HipBasis: Schur step on its
 * own H, then ks_rotate + ks_col_copy, src/run.jl:278-365) vouches for the result with
 *     ks_workspace_assert_arnoldi(ws, k)   -- "columns 0..k are orthonormal and, with the H now in ks_workspace_H,
 *                                              satisfy the Arnoldi relation of k steps"
 * k = -1 WITHDRAWS whatever the library trusted (a binding that saw the caller write into V through a path the library
 * does not watch).  *k of ks_workspace_provenance: steps the library trusts (-1: none). */
int ks_workspace_assert_arnoldi(ks_workspace* ws, int k);
int ks_workspace_provenance(const ks_workspace* ws, int* k);
/* Debugging aid: with KS_GUARD=1 in the environment a workspace puts 1 MiB canary zones on both sides of the
 * basis; *intact = 0 if any kernel wrote outside V (always 1 without KS_GUARD). */
int ks_workspace_check_guard(ks_workspace* ws, int* intact);
int ks_workspace_dims(const ks_workspace* ws, int64_t* n_local, int* maxdim, int* dtype, int64_t* ldv);
/* Host arrays (valid until the workspace is destroyed); ldh = maxdim+1, ldq = maxdim. */
int ks_workspace_H(ks_workspace* ws, void** H, int* ldh);
int ks_workspace_Q(ks_workspace* ws, void** Q, int* ldq);
/* Device pointer of column j of V (for device-side consumers; PartialSchur.Q is a view of V,
 * src/run.jl:375,389).  The pointer may be written through: handing it out ends the library's provenance (above). */
int ks_workspace_col_ptr(ks_workspace* ws, int j, void** dev_ptr);
int ks_workspace_set_seed(ks_workspace* ws, uint64_t seed);

/* ---- the verbs the reference applies to V (array-type seam) ------ */
/* copyto!(view(V,:,j+1), v1)   src/run.jl:126   (n_local elements from host) */
int ks_col_upload(ks_workspace* ws, int j, const void* host);
/* Array(view(V,:,j+1))   (result views, src/run.jl:375,389; eigvals.jl:94) */
int ks_col_download(ks_workspace* ws, int j, void* host);
/* V[:, j0:j0+ncols) -> host matrix with leading dimension ldhost */
int ks_cols_download(ks_workspace* ws, int j0, int ncols, void* host, int64_t ldhost);
int ks_cols_upload(ks_workspace* ws, int j0, int ncols, const void* host, int64_t ldhost);
/* rand!(view(V,:,j+1))   src/expansion.jl:15,21  -- counter-based uniform [0,1) */
int ks_col_fill_uniform(ks_workspace* ws, int j, uint64_t seed);
/* norm(view(V,:,j+1))   src/expansion.jl:24,41,48,81,88,96  (global 2-norm) */
int ks_col_norm(ks_workspace* ws, int j, double* out);
/* v ./= s   src/expansion.jl:28,56,106 */
int ks_col_div(ks_workspace* ws, int j, double s);
/* copyto!(view(V,:,dst+1), view(V,:,src+1))   src/run.jl:365 */
int ks_col_copy(ks_workspace* ws, int dst, int src);
/* mul!(view(V,:,jdst+1), A, view(V,:,jsrc+1))   src/expansion.jl:121 */
int ks_apply(ks_operator* A, ks_workspace* ws, int jsrc, int jdst);
/* mul!(h, view(V,:,1:j)', view(V,:,jv+1))   src/expansion.jl:37,46,84,93   (h: j host values) */
int ks_gemv_t(ks_workspace* ws, int j, int jv, void* h_host);
/* mul!(view(V,:,jv+1), view(V,:,1:j), h, -1, 1)   src/expansion.jl:38,47,85,94 */
int ks_gemv_n_sub(ks_workspace* ws, int j, int jv, const void* h_host);
/* V[:, c0:c0+r) <- V[:, c0:c0+c) * Q[0:c, 0:r)  with Q a HOST matrix, leading dimension ldq
 * (mul! into V_tmp + copyto! back, src/run.jl:363-364 and :382-383, done in place). */
int ks_rotate(ks_workspace* ws, int c0, int c, int r, const void* Q_host, int ldq);
/* out(n_local x r, host or device scratch) = V[:, 0:c) * Y[0:c, 0:r), Y complex or real host
 * matrix: the tall-skinny product of partialeigen (P.Q * vecs, src/eigvals.jl:94).  Output is
 * written to host memory `out_host` with leading dimension ldout, dtype `ydtype`. */
int ks_basis_times(ks_workspace* ws, int c, int r, const void* Y_host, int ldy, int ydtype,
                   void* out_host, int64_t ldout);

/* ---- fused hot path ---------------------------------------------------------------------- */
/* orthogonalize!(arnoldi, j)   src/expansion.jl:69-109
 * DGKS classical Gram-Schmidt of column j against columns 0..j-1 with all decisions taken on
 * globally reduced norms; writes H[0:j, j-1] and H[j, j-1] into the workspace's host H.
 * *ok = 0 on breakdown (reference returns false). */
int ks_orthogonalize(ks_workspace* ws, int j, int* ok);
/* reinitialize!(arnoldi, j, populate!)   src/expansion.jl:12-59
 * v1_host == NULL -> rand!; otherwise copyto!(v, v1).  Does not touch H. */
int ks_reinitialize(ks_workspace* ws, int j, const void* v1_host, int* ok);
/* iterate_arnoldi!(A, arnoldi, from:to)   src/expansion.jl:116-133  (from/to as in the
 * reference: step j builds 0-based column j from column j-1).  The whole range is enqueued
 * asynchronously (operator apply + fused DGKS per step, decisions on-device) and the host
 * synchronises once at the end to fetch the new columns of H.  Like the reference it needs nothing from columns < from
 * of H: the implicit second pass (ks_workspace_passes == 2), which does read them, is only taken while the library's
 * provenance of the factorisation is intact (see ks_workspace_assert_arnoldi); otherwise the explicit form runs.
 * `reorth`: the DGKS test of the implicit form is taken against max(||A v||, ||y'|| / beta) (the norm of the vector the
 * projection was really applied to; equal to the reference's rnorm, src/expansion.jl:81, unless the previous column
 * carries a correction): it can ask for the second pass where the reference would not, never the other way round. */
typedef struct ks_expand_stats {
  int32_t steps;          /* operator applications performed                                    */
  int32_t reorth;         /* steps whose DGKS test requested the second pass (src/expansion.jl:91) */
  int32_t breakdowns;     /* steps that ended in reinitialize! (src/expansion.jl:127-129)        */
  int32_t explicit_steps; /* steps the implicit second pass handed back to the explicit form (ks_workspace_set_passes) */
} ks_expand_stats;
int ks_iterate_arnoldi(ks_operator* A, ks_workspace* ws, int from, int to, ks_expand_stats* stats);

/* ---- driver:  partialschur / partialschur!   (src/run.jl:100-179, _partialschur :224-392) - */
typedef struct ks_params {
  int32_t nev;        /* default min(6, n)                       src/run.jl:103 */
  int32_t which;      /* KS_LM ...                               src/run.jl:104 */
  double tol;         /* default sqrt(eps)                       src/run.jl:105 */
  int32_t mindim;     /* default min(max(10, nev), n)            src/run.jl:106 */
  int32_t maxdim;     /* default min(max(20, 2nev), n)           src/run.jl:107 */
  int32_t restarts;   /* default 200                             src/run.jl:108 */
  int32_t start_from; /* 1-based as in partialschur!; 1 = fresh  src/run.jl:155 */
  int32_t initialize; /* 1: reinitialize!(arnoldi, start_from-1) src/run.jl:156,177 */
  int32_t reserved;
} ks_params;
typedef struct ks_history { /* src/run.jl:217-222 (+ diagnostics) */
  int32_t mvproducts;
  int32_t nconverged;
  int32_t converged;
  int32_t nev;
  int32_t restarts;   /* outer iterations performed */
  int32_t reorth;     /* DGKS second passes taken   */
  int32_t breakdowns;
  int32_t explicit_steps; /* steps redone with the explicit second pass (ks_workspace_set_passes) */
  double seconds_expand; /* wall time spent waiting for the device in iterate_arnoldi  */
  double seconds_host;   /* wall time in the host Schur / reorder / restore            */
  double seconds_rotate; /* wall time enqueueing+waiting for rotations (mostly async)  */
} ks_history;
/* Fill `p` with the reference defaults for an order-n problem (src/run.jl:103-108). */
int ks_params_default(int64_t n, ks_params* p);
/* partialschur!(A, arnoldi; ...) on a caller-owned workspace.  `v1_host` (may be NULL) is the
 * start vector of partialschur(A; v1) (src/run.jl:122-127), used only when initialize != 0 and
 * start_from == 1.  On return: V[:, 0:nconverged) = Schur vectors, H[0:nconv,0:nconv) = R,
 * eigenvalues[0:nconv) (interleaved complex doubles, src/run.jl:386-389). */
int ks_partialschur(ks_operator* A, ks_workspace* ws, const ks_params* p, const void* v1_host,
                    double* eigenvalues_c64, ks_history* history);

/* One Krylov-Schur restart on the workspace: Schur form of the active block of H, Ritz values and
 * residual estimates, lock/retain/purge grouping, three-way partition, restore_arnoldi! (all on the
 * host, src/run.jl:278-360), then the change of basis V[:, purge:k) <- V[:, purge:maxdim) Q[...] and
 * V[:, k] <- V[:, maxdim] on the device (src/run.jl:363-365).  `active` is 0-based (first non-locked
 * column).  With ks_iterate_arnoldi this lets a host language run `_partialschur`'s loop itself:
 *     ks_iterate_arnoldi(A, ws, k+1, maxdim) ; ks_restart(ws, p, active, &k, &nlock, ...) ; active = nlock
 * Out (optional): eigenvalues / residual estimates / groups of this restart (maxdim entries each). */
int ks_restart(ks_workspace* ws, const ks_params* p, int active, int* k, int* nlock, int* purge,
               double* lams_c64, double* rs, int32_t* groups);

/* One whole cycle of `_partialschur`'s loop (src/run.jl:272-365) in one call: the expansion k_in+1 .. maxdim followed by
 * the restart -- what ks_partialschur does internally.  With the explicit second pass (ks_workspace_passes == 3) the part
 * of the restart's host work that does not need H[maxdim+1, maxdim] (Schur form, Ritz values, unit residuals, ordering:
 * src/run.jl:278-289) runs on the host WHILE the device finishes the last expansion step; with the implicit second pass
 * (default) H is final only when the batch ends and the host step follows it.  Results are bit-identical to
 * ks_iterate_arnoldi + ks_restart either way.  `k_in` is the
 * basis size the previous restart left (mindim after the initial expansion).  seconds[3] (optional): wall time of the
 * expansion (including whatever of the early host part the device did not hide), of the remaining host part, of
 * enqueueing the rotation. */
int ks_expand_restart(ks_operator* A, ks_workspace* ws, const ks_params* p, int active, int k_in, int* k, int* nlock,
                      int* purge, double* eigenvalues_c64, double* residuals, int32_t* groups, ks_expand_stats* stats,
                      double* seconds);

/* Per-kernel-class timing with HIP events on the context's stream (bench.py's roofline figures).
 * Classes: 0 SpMV, 1 dots (V'w), 2 axpy (w -= Vh), 3 scale, 4 rotation, 5 reductions/decisions,
 * 6 fused axpy+dots (first projection + second-pass inner products).
 * `bytes` are ALGORITHMIC bytes accumulated per launch. */
int ks_profile_enable(ks_ctx* ctx, int on);
int ks_profile_reset(ks_ctx* ctx);
int ks_profile_get(ks_ctx* ctx, int nclass, double* ms, double* bytes, int64_t* counts);

/* Residual checks evaluated on the device: ||A*Q - Q*R||_F and ||Q'Q - I||_F for the first
 * `ncols` columns with R = H[0:ncols, 0:ncols) (test/partial_schur.jl:24-25,104-105). */
int ks_residual_norms(ks_operator* A, ks_workspace* ws, int ncols, double* resid, double* orth);
/* Arnoldi relation check ||A V_k - V_{k+1} H_k||_F and ||V'V - I||_F (test/expansion.jl:24-31). */
int ks_arnoldi_relation(ks_operator* A, ks_workspace* ws, int k, double* resid, double* orth);

/* ---- host small dense kernels, exported for the host-logic tests (no device needed) ------ */
/* local_schurfact!(H, start, to, Q)   src/schurfact.jl:393-538; H is m x n column-major. */
int ks_host_schurfact(int dtype, void* H, int m, int n, int ldh, int start, int to, void* Q, int nq,
                      int ldq);
/* One complete restart's worth of host work on (H (maxdim+1 x maxdim), Q): Schur form of the
 * active block, Ritz values/residuals, grouping, 3-way partition, restore_arnoldi!
 * (src/run.jl:278-360).  In/out: active (0-based).  Out: k, nlock, purge, eigenvalues, residuals,
 * groups. */
int ks_host_restart_step(int dtype, void* H, int ldh, void* Q, int ldq, int maxdim, int mindim,
                         int nev, int which, double tol, int active, int* k, int* nlock, int* purge,
                         double* lams_c64, double* rs, int32_t* groups);
/* sortschur!(H, Q, nconv, ordering)   src/run.jl:465-502 */
int ks_host_sortschur(int dtype, void* H, int m, int n, int ldh, void* Q, int nq, int ldq, int nconv,
                      int which);
/* givensAlgorithm (Julia stdlib LinearAlgebra = LAPACK dlartg/zlartg); out: c, s(re,im), r(re,im) */
int ks_host_givens(int dtype, const double* f, const double* g, double* c, double* s, double* r);

/* ---- diagnostics ---------------------------------------------------------------------------- */
/* Last words.  A measuring harness (bench.py with N > 1: a transport meets a new fabric for the first time) hands over the
 * text it wants on standard output should the process die under it -- SIGSEGV / SIGBUS / SIGABRT / SIGFPE / SIGILL (a
 * memory fault reported by the GPU runtime aborts the process) or SIGTERM (the launcher tearing the ranks down after a peer
 * died): the handler writes "\n<line>\n" to file descriptor 1 with write(2) and leaves with _exit(exit_code); nothing else
 * runs.  line == NULL uninstalls.  The text is copied.  No reference equivalent. */
int ks_last_words(const char* line, int exit_code);

#ifdef __cplusplus
}
#endif
#endif /* KSCHUR_H */
