#!/bin/bash
# the knobs of the sparse triangular solves (csrc/ks_sptrsv.hpp) on tools/lu_bench.py's 2-D shift-invert problems
export KS_LU_TIMEOUT_S=${KS_LU_TIMEOUT_S:-3}
for g in 64 128 256 512; do
  echo "all XCDs grid $g: $(KS_LU_XCD=0 KS_LU_GRID=$g timeout 100 python tools/lu_bench.py 500 1000 --reps 5 2>&1 | grep 'device product')"
done
for r in 0 64 128 256 512; do
  echo "one XCD, runs of $r: $(KS_LU_RUN=$r timeout 100 python tools/lu_bench.py 500 1000 --reps 5 2>&1 | grep 'device product\|error vs' | tr '\n' ' ')"
done
