#!/bin/bash
# the knobs of the sparse triangular solves (csrc/ks_sptrsv.hpp) on tools/lu_bench.py's 2-D shift-invert problems
export KS_LU_TIMEOUT_S=${KS_LU_TIMEOUT_S:-3}
for m in 3 4 0; do
  echo "KS_LU_XCD=$m: $(KS_LU_XCD=$m timeout 60 python tools/lu_bench.py 200 250 --reps 10 2>&1 | grep 'device product')"
done
KS_LU_STATS=1 timeout 100 python tools/lu_bench.py 200 250 --reps 1 2>&1 | grep "lu L\|lu U" | tail -2
LU_RESTARTS=10 timeout 300 python tools/lu_bench.py 200 250 --reps 10 --solve 2>&1 | tail -8
LU_RESTARTS=5 timeout 600 python tools/lu_bench.py 500 1000 --reps 5 --solve 2>&1 | tail -8
