#!/bin/bash
# Produces profiles/rNN_multirank_one_gpu.txt: the row-partitioned solver with 2-4 real ranks on ONE GPU
# (peer-to-peer transport), the lost-peer timeout, and the fixed cost of the distributed structure.
#   gpurun -- 'bash tools/multirank_evidence.sh > gpurun_out/multirank.txt 2>&1'
# Exits non-zero (and says so in its last line) when any leg failed: a record with a Traceback / CommTimeout / FAIL in it
# is not evidence (round 3 committed one).
export KS_SAME_DEVICE=1 KS_TRANSPORT=p2p HSA_ENABLE_IPC_MODE_LEGACY=0
set -o pipefail
port=29900
failed=0
leg() {  # run one torchrun leg, print its [rank ...] lines, remember a failure (exit status, missing OK, Traceback)
  local out
  out=$("$@" 2>&1)
  local rc=$?
  echo "$out" | grep -o "\[rank [0-9]\][^[]*"
  if [ $rc -ne 0 ] || echo "$out" | grep -q "Traceback\|CommTimeout: \|-> FAIL"; then
    echo "!! leg failed (exit $rc): $*"
    echo "$out" | grep "Traceback\|Error\|Timeout" | head -5
    failed=$((failed+1))
  fi
}
echo "# multi-rank product path on ONE MI355X (all ranks on device 0, peer-to-peer transport), tools/dist_gpu_check.py"
for cfg in "2 laplace 20" "3 laplace 20" "4 laplace 20" "2 hashed 18" "4 hashed 18" "2 wide 16" "3 complex 16"; do
  set -- $cfg; port=$((port+1))
  echo "## $1 ranks, mode $2, m=$3"
  leg timeout 300 python -m torch.distributed.run --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port tools/dist_gpu_check.py $2 $3
done
echo "# the RCCL transport's launch structure with real ranks: host-staged transport (ks_ctx_create_hostcomm over gloo)"
for cfg in "2 laplace 20" "3 hashed 18" "2 complex 16" "3 eager 16"; do
  set -- $cfg; port=$((port+1))
  echo "## $1 ranks, mode $2, m=$3, KS_TRANSPORT=host"
  KS_TRANSPORT=host leg timeout 300 python -m torch.distributed.run --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port tools/dist_gpu_check.py $2 $3
done
echo "# distributed operator in the column-blocked layout (config-3-like hashed matrix, n = 110^3): per-rank product bit-identical to the single-GPU product; KS_SPMV_COLBLOCKS=0 = plain CSR row blocks beside it"
for np_ in 2 4; do
  port=$((port+1))
  echo "## $np_ ranks, cbprod 110 (automatic layout)"
  leg timeout 600 python -m torch.distributed.run --nproc-per-node $np_ --master-addr 127.0.0.1 --master-port $port tools/dist_gpu_check.py cbprod 110
  port=$((port+1))
  echo "## $np_ ranks, cbprod 110, KS_SPMV_COLBLOCKS=0"
  KS_SPMV_COLBLOCKS=0 KS_EXPECT_LAYOUT=csr leg timeout 600 python -m torch.distributed.run --nproc-per-node $np_ --master-addr 127.0.0.1 --master-port $port tools/dist_gpu_check.py cbprod 110
done
echo "# BASELINE config 5 at true per-rank size: 8 ranks x (464 x 464 x 58 rows) on device 0 vs the single-process 464^3 run"
echo "## host-staged transport (RCCL launch structure), 8 ranks"
port=$((port+1))
KS_TRANSPORT=host leg timeout 900 python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $port tools/dist_gpu_check.py shard5 464
echo "## peer-to-peer transport, 6 ranks (8 peer-to-peer processes on ONE device exceed what it schedules concurrently: DESIGN.md section 7)"
port=$((port+1))
leg timeout 900 python -m torch.distributed.run --nproc-per-node 6 --master-addr 127.0.0.1 --master-port $port tools/dist_gpu_check.py shard5 464
echo "## set-up skew: rank 1 dawdles 12 s with KS_P2P_TIMEOUT_S=5 (the barrier between set-up and the first exchange absorbs it)"
port=$((port+1))
KS_P2P_TIMEOUT_S=5 KS_TEST_SETUP_SKEW_S=1:12 leg timeout 300 python -m torch.distributed.run --nproc-per-node 3 --master-addr 127.0.0.1 --master-port $port tools/dist_gpu_check.py laplace 20
echo "## lost peer (KS_P2P_TIMEOUT_S=2): the expected outcome IS a CommTimeout on rank 0"
KS_P2P_TIMEOUT_S=2 timeout 120 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29999 tools/dist_gpu_check.py timeout 2>&1 | grep "^\[rank" || { echo "!! lost-peer leg failed"; failed=$((failed+1)); }
unset KS_SAME_DEVICE KS_TRANSPORT
echo "# fixed cost of the distributed structure on one GPU (tools/dist_overhead.py 108 = the 8-way share of 216^3), separate processes"
for leg in plain rccl p2p plain rccl p2p; do python tools/dist_overhead.py 108 $leg 2>&1 | grep ms/iter; done
if [ $failed -ne 0 ]; then echo "# RESULT: $failed leg(s) FAILED -- this record is not evidence"; exit 1; fi
echo "# RESULT: all legs OK"
