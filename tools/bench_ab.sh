#!/bin/bash
# A/B of env-switchable variants inside ONE gpurun call: one bench.py run per argument ("ENV=value" or
# "ENV1=a ENV2=b"), prints iterations/s and the per-class average launch times (us).
#   gpurun -- 'bash tools/bench_ab.sh KS_SPMV_FORMAT=csr KS_SPMV_FORMAT=dvi'
for v in "$@"; do
  env $v python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/b.json 2>/dev/null
  python - "$v" <<PY
import json,sys
d=json.load(open("/tmp/b.json")); pc=d["roofline"]["per_class"]
print(sys.argv[1].ljust(34), round(d["value"],1), {k:round(v["ms_total"]/v["launches"]*1e3,1) for k,v in pc.items() if v["launches"] and k in ("spmv","fused","dots","axpy","rotate")})
PY
done
