#!/bin/bash
# Repeats several multi-rank configurations on one GPU (determinism / race hunting).
export KS_SAME_DEVICE=1 KS_TRANSPORT=p2p
port=29700
for rep in 1 2 3; do for cfg in "4 hashed 22" "3 laplace 30" "4 complex 16" "2 wide 20"; do
  set -- $cfg; port=$((port+1))
  out=$(timeout 300 python -m torch.distributed.run --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $port tools/dist_gpu_check.py $2 $3 2>&1)
  ok=$(echo "$out" | grep -o -- "-> OK" | wc -l); same=$(echo "$out" | grep -c "same: True")
  echo "rep $rep: $cfg -> ok=$ok/$1 same=$same $(echo "$out" | grep -o 'in [0-9]* matrix-vector' | head -1)"
done; done
