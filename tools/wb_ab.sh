#!/bin/bash
# write-back staging depth of the projection kernel: KS_FUSED_WB=8 (32 KiB bursts, 2 workgroups / CU) vs 24 (96 KiB bursts, 1 / CU; default)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for wb in 8 24; do
  KS_FUSED_WB=$wb python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('headline WB=$wb:', 'value %.1f' % d['value'], 'dominant %s frac %.3f' % (r['kernel'][:16], r['frac']), 'moved_frac %.3f' % r['fused_step']['moved_frac'], {k: round(v['GBps']) for k, v in r['per_class'].items() if v.get('GBps')})"
done; done
for c in cfg2 cfg3 cfg4big; do for wb in 8 24; do
  KS_FUSED_WB=$wb python tools/config_bench.py $c 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c WB=$wb:', 'iters_per_s %.1f' % d['iters_per_s'], {k: round(v.get('GBps') or 0) for k, v in d['per_class'].items() if v.get('GBps')})"
done; done
for wb in 8 24; do KS_FUSED_WB=$wb python tools/dist_overhead.py 108 plain 2>&1 | grep ms/iter | sed "s/^/shard-size WB=$wb: /"; done
