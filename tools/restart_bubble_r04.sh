#!/bin/bash
# Restart bubble of the per-step and of the s-step expansion from kernel traces, config 2 (n = 1e6) and the
# headline (n = 1e7), with and without the reverse mailbox (KS_ROT_GATE: the rotation pre-enqueued behind a gate the host
# releases), plus the host-side time of the restart's host step.
#   gpurun --timeout 900 -- 'bash tools/restart_bubble_r04.sh > gpurun_out/restart_bubble_r04.txt 2>&1'
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for cfg in cfg2 headline; do for s in 0 20; do for g in 0 1; do
  rm -rf /tmp/kt_$cfg$s$g
  KS_ROT_GATE=$g KS_SSTEP=$s rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$cfg$s$g -- python tools/run_solver.py $cfg 25 > /tmp/rs_$cfg$s$g.txt 2>&1
  python tools/restart_bubble.py $(find /tmp/kt_$cfg$s$g -name '*kernel_trace.csv' | head -1) "$cfg KS_SSTEP=$s KS_ROT_GATE=$g"
  grep "host step" /tmp/rs_$cfg$s$g.txt | sed "s/^/    /"
done; done; done
echo "# whole solves, wall time, un-profiled, alternating"
for rep in 1 2 3; do for g in 0 1; do
  KS_ROT_GATE=$g KS_SSTEP=20 python tools/run_solver.py cfg2 60 2>/dev/null | sed "s/^.*| restarts/cfg2 sstep=20 gate=$g: restarts/"
done; done
