#!/bin/bash
# Stage timers of k_fin_blk (the block's reduction + small algebra kernel): rebuilds the library ON THE GPU BOX with
# -DKS_FIN_TIMING (device-side wall_clock64 stamps, one printf per kernel of the k = 21, s = 20 block), runs a few cycles of the
# bench workload and of config 2, and leaves the lines in gpurun_out/fin_blk_timing.txt.  The timing build does not travel back.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
KS_EXTRA_HIPCC_FLAGS=-DKS_FIN_TIMING python arnoldimethod.jl_amd/build.py > $OUT/fin_blk_build.log 2>&1 || { tail -5 $OUT/fin_blk_build.log; exit 1; }
{
  echo "== bench workload (216^3), 6 cycles"
  python bench.py --steps 4 --warmup 2 2>&1 | grep "fin_blk" | tail -8
  echo "== config 2 (100^3)"
  python tools/config_bench.py 2 --sstep 20 --steps 4 --warmup 2 2>&1 | grep "fin_blk" | tail -8
} > $OUT/fin_blk_timing.txt
cat $OUT/fin_blk_timing.txt
