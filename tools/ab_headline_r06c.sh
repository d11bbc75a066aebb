#!/bin/bash
# Headline A/B on ONE box, alternating processes, the driver's flags (--steps 20 --warmup 5):
#   default (cycles back to back, the restart waits for an event) | --sync-cycles | KS_QSTAGE_EVENT=0 (stream synchronisation, cycles back to back)
#   gpurun --timeout 900 -- 'bash tools/ab_headline_r06c.sh'
cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --no-cpu-baseline --no-shift-invert --no-profile --steps 20 --warmup 5"
one() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],4), d['validation']['ok'])"; }
for rep in 1 2 3 4 5; do
  $B 2>/dev/null | one "default          "
  $B --sync-cycles 2>/dev/null | one "sync-cycles      "
  KS_QSTAGE_EVENT=0 $B 2>/dev/null | one "stream-sync      "
  KS_QSTAGE_EVENT=0 $B --sync-cycles 2>/dev/null | one "round-6a (both)  "
done
