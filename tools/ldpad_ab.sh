#!/bin/bash
# does the column stride of V (mod the memory interleave) matter for the streaming kernels?  KS_LD_PAD = extra 512-byte units per column
cd $GRAFT_REPO_ROOT
for pad in ${PADS:-0 1 3 8 9 33 64 129 512 0}; do for rep in $(seq ${REPS:-2}); do
KS_LD_PAD=$pad python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('pad $pad:', 'value %.1f' % d['value'], {k: round(v['GBps']) for k, v in r['per_class'].items() if v.get('GBps')})"
done; done
