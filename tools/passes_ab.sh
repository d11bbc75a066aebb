#!/bin/bash
# implicit second pass (KS_PASSES=2, default) vs the second projection applied to the vector (KS_PASSES=3): headline bench
# and configs 2, 3, 4-big; alternating processes.   gpurun -- 'bash tools/passes_ab.sh > gpurun_out/passes_ab.txt 2>&1'
cd $GRAFT_REPO_ROOT
line() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; f = r['fused_step']
print('$1', 'value %.1f' % d.get('value', d.get('iters_per_s', 0)), '| dominant', r.get('kernel', '?')[:18], 'frac %.3f' % r.get('frac', 0), '| moved_frac %.3f algorithmic_frac %.3f' % (f['moved_frac'], f['algorithmic_frac']),
      '| expand %.4f s restart %.4f s' % (f['expand_seconds'], f['restart_seconds']), '| per class', {k: round(v['GBps']) for k, v in r.get('per_class', {}).items() if v.get('GBps')})
"; }
for rep in 1 2; do for p in 3 2; do
  KS_PASSES=$p python bench.py --no-cpu-baseline 2>/dev/null | line "headline KS_PASSES=$p:"
done; done
for cfg in cfg2 cfg3 cfg4big; do for p in 3 2; do
  KS_PASSES=$p python tools/config_bench.py $cfg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg KS_PASSES=$p:', 'iters_per_s %.1f' % d['iters_per_s'])"
done; done
