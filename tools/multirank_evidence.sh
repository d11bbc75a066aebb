#!/bin/bash
# Produces profiles/r01_multirank_one_gpu.txt: the row-partitioned solver with 2-4 real ranks on ONE GPU
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
echo "## lost peer (KS_P2P_TIMEOUT_S=2)"
KS_P2P_TIMEOUT_S=2 timeout 120 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29999 tools/dist_gpu_check.py timeout 2>&1 | grep "^\[rank"
unset KS_SAME_DEVICE KS_TRANSPORT
echo "# fixed cost of the distributed structure on one GPU (tools/dist_overhead.py 108 = the 8-way share of 216^3), separate processes"
for leg in plain rccl p2p plain rccl p2p; do python tools/dist_overhead.py 108 $leg 2>&1 | grep ms/iter; done
