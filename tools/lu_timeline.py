"""Where a sparse triangular solve spends its time: KS_LU_TIMELINE=<prefix> makes ks_operator_lu record, per row (level
order), the wall clock at which its wave had its ticket, had summed the entries of earlier chunks, and published the row.
    KS_LU_TIMELINE=gpurun_out/lu_tl python tools/lu_bench.py 500 1000 --reps 1;  python tools/lu_timeline.py gpurun_out/lu_tl.u64"""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).astype(np.float64).reshape(2, -1, 4)
for name, a in (("first factor", raw[0]), ("second factor", raw[1])):
    a = a[a[:, 2] > 0]  # (rows of the other factor's padding)
    n = a.shape[0]
    t0 = a[:, 0].min()
    tick, far, pub, m = (a[:, 0] - t0) * 0.01, (a[:, 1] - t0) * 0.01, (a[:, 2] - t0) * 0.01, a[:, 3]
    front = np.maximum.accumulate(pub)
    print(f"{name}: {n} rows, {front[-1]:.0f} us")
    # rows published per 100 us
    edges = np.arange(0.0, pub.max() + 100.0, 100.0)
    hist, _ = np.histogram(pub, bins=edges)
    print("   rows published per 100 us:", " ".join(str(int(h)) for h in hist))
    for frac in (0.5, 0.9, 0.98, 0.99, 0.995, 1.0):
        k = min(n - 1, int(frac * n) - 1)
        print(f"   {100 * frac:5.1f} % of the rows published after {front[k]:9.1f} us")
    # by position in the numbering (the groups are contiguous ranges): when did each sixteenth of the rows start / finish?
    parts = np.array_split(np.arange(n), 16)
    print("   sixteenths of the numbering, first ticket .. last row published (us): " + "  ".join(f"{tick[q].min():.0f}..{pub[q].max():.0f}" for q in parts))
    # the slowest stretch: 1024 consecutive rows with the largest time span
    w = 1024
    span = front[w:] - front[:-w]
    k = int(np.argmax(span))
    print(f"   slowest {w} rows: [{k}, {k + w}) in {span[k]:.0f} us = {span[k] / w:.3f} us per row; entries from own chunk there: mean {m[k:k + w].mean():.1f}")
    sl = slice(k, k + w)
    chunk_first = np.arange(k - k % 16, k + w, 16)
    chunk_first = chunk_first[(chunk_first >= k) & (chunk_first + 16 <= k + w)]
    gaps_far = np.array([far[c:c + 16].max() - front[c - 1] for c in chunk_first if c > 0])       # previous chunk done -> all far parts summed
    chain = np.array([pub[c:c + 16].max() - far[c:c + 16].max() for c in chunk_first])            # far parts summed -> chunk published
    wait = np.array([far[c:c + 16].min() - tick[c:c + 16].min() for c in chunk_first])
    print(f"   per chunk there: previous chunk published -> far parts summed {np.median(gaps_far):.2f} us (median), far parts summed -> last row published {np.median(chain):.2f} us; rows held their ticket {np.median(wait):.0f} us before that")
    c = chunk_first[len(chunk_first) // 2]
    print("   one chunk, per row (us after the previous chunk's last row): far done / published / own-chunk entries")
    for i in range(c, c + 16):
        print(f"      row {i}: {far[i] - front[c - 1]:7.2f} {pub[i] - front[c - 1]:7.2f}  {int(m[i])}")
