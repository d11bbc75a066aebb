// Launch interface of the streaming kernels of the s-step expansion (ks_block_kernels.hpp).  The kernels are instantiated
// for ~200 (element type, columns per wave, block size) combinations; they live in translation units of their own
// (ks_block_inst.hip, compiled in parts and in parallel with the main unit by build.py) behind this plain interface.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

struct BlkLaunchArgs {
  void* V;             // basis, column-major, leading dimension ld (elements)
  int64_t ld;
  int dtype;           // KS_F64 (0) / KS_C64 (1)
  int k, s;            // existing columns, block size
  void* partial;       // [entry][workgroup] partial sums, column stride pnb
  int pnb;
  const void* coefp;   // second pass: (T P) R1^-1, k x s, leading dimension k
  const void* r1inv;   // second pass: R1^-1, s x s
  const void* zeros;   // >= 16 zero bytes in device memory (ring forms: source of the packs past a workgroup's rows)
  const void* st;      // DevState of the batch
  int dbg;             // probe flags (1: second pass without stores)
  int nt;              // non-temporal loads of the basis
  int num_cu, bpc;     // device CUs, cap on resident workgroups per CU (ks_ctx::bpc)
  hipStream_t stream;
  // block columns read from elsewhere than V[:, k : k + s) (matrix-instruction forms of both passes and the fused
  // rotation: the Newton chain was written to scratch columns); null: in place
  const void* zsrc = nullptr;
  int64_t ldz = 0;
  // which = 2 (restart rotation fused with the first pass, k_brotdots_mfma): V[:, out0 : k) <- V[:, 0:cin) M, M = cin x (k - out0)
  int cin = 0, out0 = 0;
  const void* rotm = nullptr;
};
// which = 0: k_bdots (pass 1), 1: k_bupdate (pass 2), 2: restart rotation + pass 1 in one sweep (part 1 only).  Returns the number of workgroups launched (= partial sums per
// entry); throws std::runtime_error for a shape without an instantiation (ks_blk_shape_ok says which exist).
int ks_blk_launch_part0(int which, const BlkLaunchArgs& a);   // Float64, block sizes 1-4
int ks_blk_launch_part1(int which, const BlkLaunchArgs& a);   // Float64, block sizes 5, 8, 10, 20
int ks_blk_launch_part2(int which, const BlkLaunchArgs& a);   // ComplexF64, block sizes 1-5
// The matrix-instruction forms (ks_block_mfma.hpp) take the block size at run time -- a block of s steps runs on the kernel of
// ceil(s / 4) column tiles, the missing columns are zeros --, so every size up to 4 NT of an instantiated tile count NT exists:
// Float64 NT = 2 (s <= 8, up to 64 columns), 3 (s <= 12, up to 48), 4 (s <= 16, 25-28 columns), 5 (s <= 20, up to 24); ComplexF64 NT = 2 (s <= 8, up to 48 columns), 3 (s <= 10, up to 32).
// Tile count for (dtype, k, s) if those forms are switched on (KS_BLK_MFMA) for BOTH passes, else 0.
int ks_blk_mfma_nt_f64(int k, int s);
int ks_blk_mfma_nt_c64(int k, int s);
inline int ks_blk_mfma_nt(int dtype, int k, int s) { return dtype == 0 ? ks_blk_mfma_nt_f64(k, s) : ks_blk_mfma_nt_c64(k, s); }
// both passes of this block can read its columns from scratch columns (BlkLaunchArgs::zsrc: matrix-instruction forms only)
inline bool ks_blk_zsrc_ok(int dtype, int k, int s) { return s > 5 && ks_blk_mfma_nt(dtype, k, s) > 0; }
// which = 2: is there a fused rotation + first pass for this shape?  (cin old columns -> k new ones, block of s)
bool ks_blk_rot_ok_f64(int cin, int k, int s);
bool ks_blk_rot_ok_c64(int cin, int k, int s);
inline bool ks_blk_rot_ok(int dtype, int cin, int k, int s) { return dtype == 0 ? ks_blk_rot_ok_f64(cin, k, s) : ks_blk_rot_ok_c64(cin, k, s); }
inline int ks_blk_launch(int which, const BlkLaunchArgs& a) {
  if (a.dtype != 0) return ks_blk_launch_part2(which, a);
  return a.s <= 4 ? ks_blk_launch_part0(which, a) : ks_blk_launch_part1(which, a);
}
// shapes with a kernel: block sizes 1-5 in the register forms (ComplexF64 up to 32 columns); Float64 8 (up to 48 columns), 10 (up
// to 32), 20 (up to 24) in register / ring forms; everything else up to 20 (ComplexF64: 10) on the matrix instruction only
inline bool ks_blk_shape_ok(int dtype, int k, int s) {
  if (k < 1 || s < 1 || k + s > 65) return false;
  if (dtype != 0) return (s <= 5 && k <= 32) || ks_blk_mfma_nt_c64(k, s) > 0;   // (33-48 columns: matrix-instruction forms for every size)
  if (s <= 5) return true;
  if ((s == 8 && k <= 48) || (s == 10 && k <= 32) || (s == 20 && k <= 24)) return true;
  return ks_blk_mfma_nt_f64(k, s) > 0;
}
