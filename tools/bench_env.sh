#!/bin/bash
# usage: tools/bench_env.sh "ENV1=.. ENV2=.." ...   (one bench run per argument; prints per-class GB/s)
for envs in "$@"; do
  echo "=== $envs"
  env $envs timeout 600 python bench.py --no-cpu-baseline --steps 6 --warmup 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('iters/s %.1f  fused_frac %.3f' % (d['value'], r['fused_step']['frac']))
print('  '.join('%s %.0f GB/s %.1fms' % (k, (v['GBps'] or 0), v['ms_total']) for k,v in r['per_class'].items()))"
done
