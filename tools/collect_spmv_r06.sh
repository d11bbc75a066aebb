#!/bin/bash
# Round-6 SpMV evidence (tools/spmv_slab.hip): gpurun --timeout 900 -- 'bash tools/collect_spmv_r06.sh'
# Output: gpurun_out/spmv6/ (copy into profiles/ as r06_spmv_*)
#   variants.txt   every variant (library kernel, slab shapes, marching kernel) with the caches flushed + ping-pong chain of 20
#   sweep.txt      the marching kernel over the workgroups per XCD (one plane per round = 92 at 216^3), ping-pong chain
#   (sweep / columns arguments: S = workgroups per XCD of the marching kernel; 70000 + S the window form; 80000 + R the z-marching form
#    with R z-ranges, 90000 + R with four window buffers, 100000 + R with 512-thread workgroups; 5000 + S a plain copy; 20000.. the slot ladder)
#   columns.txt    the chain the SOLVER runs (product i: column i -> column i + 1 of a 21-column basis): library kernel, marching
#                  kernel, plain copy, and the slot ladder (1 / 3 / 5 / 3-far / 7 slots) -- where the time goes
#   probe_dbg.txt, probe_pmc.txt   parts of the slab kernel switched off; PMC counters of the library kernel and slab shapes
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/spmv6
mkdir -p $OUT
BIN=$REPO/tools/_build/spmv_slab
timeout 300 $BIN 216 > $OUT/variants.txt 2>&1
{ timeout 200 $BIN 216 216 216 sweep 46 60 80 88 90 91 92 94 96 100 112 128 160 182 192 224 256;
  timeout 100 $BIN 216 216 216 sweep 70092 70192 80016 80024 80032 90024 100024;
  timeout 100 $BIN 215 216 217 sweep 88 90 91 92 96 182 70192 80024;
  timeout 100 $BIN 100 100 100 sweep 19 20 39 59 78 96 128 192; } > $OUT/sweep.txt 2>&1
{ timeout 300 $BIN 216 216 216 columns 0 92 -92 192 -192 0 10000 5512 -5512 6024 20092 30092 40092 50092 92 20192 30192 40192 50192 192 70192 80016 80024 80032 90024 100024 0 70192 80024; } > $OUT/columns.txt 2>&1
ls -la $OUT
