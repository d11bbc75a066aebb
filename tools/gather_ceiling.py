#!/usr/bin/env python3
"""How fast can this GPU gather random 8-byte elements?  (ceiling for the x gathers of config 3's SpMV)
torch.index_select / take on a vector of n doubles with m random indices -> G gathers/s."""
import torch

dev = torch.device("cuda")
for n in (200_000, 1_000_000, 10_000_000):
    x = torch.rand(n, dtype=torch.float64, device=dev)
    m = 5 * n
    idx = torch.randint(0, n, (m,), device=dev, dtype=torch.int32).to(torch.int64)
    idx32 = idx.to(torch.int32)
    for name, f in (("index_select i64", lambda: torch.index_select(x, 0, idx)), ("take i64", lambda: torch.take(x, idx))):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            f()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 20
        print(f"n={n:>9} gathers={m:>9} {name:<18} {us:8.1f} us  {m / us / 1e3:7.1f} G gathers/s  (bytes moved incl. index+output: {(m * 24) / us / 1e6:6.2f} TB/s)")
