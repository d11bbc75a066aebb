#!/bin/bash
# A/B of the restart hand-over: round-2 structure (hipMemcpyAsync + hipStreamSynchronize, strictly sequential
# host step: KS_MAILBOX=0 KS_EARLY_RESTART=0), mailbox only, mailbox + early hand-over of H (default).  Restart bubble from
# kernel traces at n = 1e6 (config 2) and n = 1e7 (headline), host-side timeline of a batch (KS_EARLY_DEBUG=1), and wall
# time of whole solves.    gpurun -- 'bash tools/early_restart_ab.sh > gpurun_out/early_ab.txt 2>&1'
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in cfg2 headline; do for mode in "0 0" "1 0" "1 1"; do set -- $mode
  rm -rf /tmp/kt_$cfg$1$2
  KS_MAILBOX=$1 KS_EARLY_RESTART=$2 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$cfg$1$2 -- python tools/run_solver.py $cfg 25 > /dev/null 2>&1
  python tools/restart_bubble.py $(find /tmp/kt_$cfg$1$2 -name '*kernel_trace.csv' | head -1) "$cfg KS_MAILBOX=$1 KS_EARLY_RESTART=$2"
done; done
echo "# host-side timeline of one expansion batch + restart (KS_EARLY_DEBUG=1), last 2 cycles of 8"
for cfg in cfg2 headline; do for mode in "0 0" "1 1"; do set -- $mode
  echo "== $cfg KS_MAILBOX=$1 KS_EARLY_RESTART=$2"; KS_EARLY_DEBUG=1 KS_MAILBOX=$1 KS_EARLY_RESTART=$2 python tools/run_solver.py $cfg 8 2>&1 | grep -v amdgpu.ids | tail -3
done; done
echo "# whole solves, wall time (expand + host + rotate), un-profiled, alternating"
for rep in 1 2 3; do for mode in "0 0" "1 1"; do set -- $mode
  KS_MAILBOX=$1 KS_EARLY_RESTART=$2 python tools/run_solver.py cfg2 60 2>/dev/null | sed "s/^.*| restarts/cfg2 mailbox=$1 early=$2: restarts/"
done; done
for rep in 1 2; do for mode in "0 0" "1 1"; do set -- $mode
  KS_MAILBOX=$1 KS_EARLY_RESTART=$2 python tools/config_bench.py cfg3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 mailbox=$1 early=$2:', {k: d[k] for k in d if k in ('value','iters_per_s','ms_per_step','metric')})"
done; done
