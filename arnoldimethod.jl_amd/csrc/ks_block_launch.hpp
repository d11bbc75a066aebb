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
  // block columns read from elsewhere than V[:, k : k + s) (matrix-instruction forms of the second pass and the fused
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
// ComplexF64 blocks of 8 / 10 exist on the matrix instruction only: are those forms on (KS_BLK_MFMA bit 7)?
bool ks_blk_cx_mfma_on();
// which = 2: is there a fused rotation + first pass for this shape?  (Float64; cin old columns -> k new ones, block of s)
bool ks_blk_rot_ok(int cin, int k, int s);
inline int ks_blk_launch(int which, const BlkLaunchArgs& a) {
  if (which == 2) return ks_blk_launch_part1(which, a);
  if (a.dtype != 0) return ks_blk_launch_part2(which, a);
  return a.s <= 4 ? ks_blk_launch_part0(which, a) : ks_blk_launch_part1(which, a);
}
// instantiated shapes: Float64 s in {1..5, 8, 10, 20} (8 up to 48 columns, 10 up to 32, 20 up to 24), ComplexF64 s in {1..5, 8, 10} up to 32 columns
inline bool ks_blk_shape_ok(int dtype, int k, int s) {
  if (k < 1 || k + s > 65) return false;
  if (dtype != 0) return ((s >= 1 && s <= 5) || ((s == 8 || s == 10) && ks_blk_cx_mfma_on())) && k <= 32;   // (8, 10: matrix-instruction forms only)
  if (s >= 1 && s <= 5) return true;
  return (s == 8 && k <= 48) || (s == 10 && k <= 32) || (s == 20 && k <= 24);
}
