#!/bin/bash
# the launch / layout knobs of the sparse triangular solves (csrc/ks_sptrsv.hpp) on tools/lu_bench.py's 2-D shift-invert problem
# (n = 500 x 1000 by default; the factorisation runs on the host every time: ~15 s per line)
export KS_LU_TIMEOUT_S=${KS_LU_TIMEOUT_S:-5}
DIMS=${DIMS:-"500 1000"}
run() { echo "$1: $(env $1 timeout 300 python tools/lu_bench.py $DIMS --reps 10 2>&1 | grep 'device product' | cut -c1-60)"; }
run "KS_LU_XCD=3"
run "KS_LU_XCD=4"
run "KS_LU_XCD=0 KS_LU_GROUPS=1"
run "KS_LU_GROUPS=1"
run "KS_LU_LAYERS=1"
run "KS_LU_PREPASS=0"
run "KS_LU_RUN=0"
run "KS_LU_RUN=128"
run "KS_LU_RUN=512"
run "KS_LU_NARROW=16"
run "KS_LU_NARROW=128"
run "KS_LU_RUN_COND=100"
run "KS_LU_BACKOFF=0"
run "KS_LU_BACKOFF=8"
run "KS_LU_GRID_GROUPS=128"
