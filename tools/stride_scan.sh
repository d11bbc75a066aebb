#!/bin/bash
# column-stride rule: for several problem sizes, pad 0 vs pads that put (stride mod 128 KB) at chosen residues
cd $GRAFT_REPO_ROOT
for m in ${MS:-216 232}; do
  for res in ${RES:-0x0f200 0x1f200 0x0fa00 0x1fa00 0x0ea00 0x1ea00 0x0fe00 0x1fe00 0x0f600 0x1f600 0x17200 0x07200}; do
    pad=$(python3 -c "n=$m**3; ld=(n+63)//64*64; print((($res - (ld*8) % 0x20000) % 0x20000)//512)")
    KS_LD_PAD=$pad python tools/stride_probe.py $m 2>/dev/null | tail -1 | sed "s/^/res $res /"
  done
done
