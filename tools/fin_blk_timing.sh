#!/bin/bash
# Stage timers of k_fin_blk (reduction + small algebra of the s-step expansion): rebuilds the library ON THE GPU BOX with
# -DKS_FIN_TIMING (device-side wall clock, printed by the last workgroup at k = 31, and at k = 21 for blocks of 20) and runs a few headline cycles.
#   gpurun --timeout 900 -- 'bash tools/fin_blk_timing.sh > gpurun_out/fin_blk_timing.txt 2>&1'
export KS_EXTRA_HIPCC_FLAGS="-DKS_FIN_TIMING"
python arnoldimethod.jl_amd/build.py > /dev/null 2>&1
python bench.py --no-cpu-baseline --no-shift-invert --no-profile --steps 2 --warmup 2 --grid ${1:-100} 2>&1 | grep "fin_blk" | tail -8
unset KS_EXTRA_HIPCC_FLAGS
