#!/bin/bash
# Round 6c A/B on one box: the restart's stream synchronisation replaced by an event (KS_QSTAGE_EVENT), true start of the chain
# (KS_TRUE_START) -- configs 2-4 and the headline, cycles back to back.   gpurun --timeout 900 -- 'bash tools/ab_r06c.sh'
cd ${GRAFT_REPO_ROOT:-.}
one() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
if 'iters_per_s' in d: print('$1', round(d['iters_per_s']), round(d['ms_per_cycle'],4), d['sstep']['chains_adopted'], d['sstep']['fused_rotations'], '%.1e'%d['sstep']['gram_dev'], '%.1e'%d['validation']['arnoldi_rel'])
else: print('$1', round(d['value'],1), round(d['ms_per_step'],4), d['roofline']['fused_step'].get('cycle_frac'), d['validation']['ok'])
"; }
for rep in 1 2; do
for c in cfg3 cfg2 cfg4; do
  for ev in 1 0; do for ts in 1 0; do
    [ $c != cfg3 ] && [ $ts = 0 ] && continue
    KS_QSTAGE_EVENT=$ev KS_TRUE_START=$ts python tools/config_bench.py $c --sstep 20 --steps 20 2>/dev/null | one "$c event=$ev true_start=$ts"
  done; done
done
for ev in 1 0; do KS_QSTAGE_EVENT=$ev python bench.py --no-cpu-baseline --no-shift-invert --steps 20 2>/dev/null | one "headline event=$ev"; done
done
