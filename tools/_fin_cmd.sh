python -m pytest tests/test_gpu_sstep.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
python bench.py > gpurun_out/fin_bench.json 2>gpurun_out/fin_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/fin_bench.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('fused_step',{}).get('cycle_frac'))
PY
for s in 20 0; do python tools/config_bench.py cfg2 --sstep $s 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2', d.get('sstep'), d.get('value'))"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 > /dev/null 2>&1
find /tmp/kt -name "*kernel_stats.csv" | head -1 | xargs grep -E "fin_blk|fin_step" | sed 's/(.*)"//' | cut -c1-200
