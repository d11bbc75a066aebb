#!/bin/bash
# Produces profiles/rNN_multirank_one_gpu.txt: the row-partitioned solver with 2-4 real ranks on ONE GPU
# (peer-to-peer transport), the lost-peer timeout, and the fixed cost of the distributed structure.
#   gpurun -- 'bash tools/multirank_evidence.sh > gpurun_out/multirank.txt 2>&1'
export KS_SAME_DEVICE=1 KS_TRANSPORT=p2p
port=29900
echo "# multi-rank product path on ONE MI355X (all ranks on device 0, peer-to-peer transport), tools/dist_gpu_check.py"
for cfg in "2 laplace 20" "3 laplace 20" "4 laplace 20" "2 hashed 18" "4 hashed 18" "2 wide 16" "3 complex 16"; do
  set -- $cfg; port=$((port+1))
  echo "## $1 ranks, mode $2, m=$3"
  timeout 300 python -m torch.distributed.run --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port tools/dist_gpu_check.py $2 $3 2>&1 | grep -o "\[rank [0-9]\][^[]*" 
done
echo "# the RCCL transport's launch structure with real ranks: host-staged transport (ks_ctx_create_hostcomm over gloo)"
for cfg in "2 laplace 20" "3 hashed 18" "2 complex 16" "3 eager 16"; do
  set -- $cfg; port=$((port+1))
  echo "## $1 ranks, mode $2, m=$3, KS_TRANSPORT=host"
  KS_TRANSPORT=host timeout 300 python -m torch.distributed.run --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port tools/dist_gpu_check.py $2 $3 2>&1 | grep -o "\[rank [0-9]\][^[]*"
done
echo "# BASELINE config 5 at true per-rank size: 8 ranks x (464 x 464 x 58 rows) on device 0 vs the single-process 464^3 run"
port=$((port+1))
timeout 900 python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $port tools/dist_gpu_check.py shard5 464 2>&1 | grep "^\[rank"
echo "## lost peer (KS_P2P_TIMEOUT_S=2)"
KS_P2P_TIMEOUT_S=2 timeout 120 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29999 tools/dist_gpu_check.py timeout 2>&1 | grep "^\[rank"
unset KS_SAME_DEVICE KS_TRANSPORT
echo "# fixed cost of the distributed structure on one GPU (tools/dist_overhead.py 108 = the 8-way share of 216^3), separate processes"
for leg in plain rccl p2p plain rccl p2p; do python tools/dist_overhead.py 108 $leg 2>&1 | grep ms/iter; done
