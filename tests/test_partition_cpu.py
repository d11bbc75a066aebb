"""`-m "not gpu"`: ks_sstep_partition (include/kschur.h) -- how iterate_arnoldi!(A, arnoldi, from:to) (src/expansion.jl:116-133) is
cut into blocks by the library's own blk_partition.  No device call: the function only consults which kernel forms exist."""
import numpy as np
import pytest

from __graft_entry__ import import_package

pkg = import_package()


def test_ranges_are_cut_into_as_few_blocks_as_the_tile_counts_allow():
    f, c = np.float64, np.complex128
    assert pkg.sstep_partition(f, 21, 20, 20) == [20]            # the headline: one block per restart cycle at 20/40
    assert pkg.sstep_partition(f, 11, 9, 20) == [9]              # config 3 after a restart that kept a 2 x 2 block whole
    assert pkg.sstep_partition(f, 25, 16, 20) == [16]            # 20/40 with 4 locked vectors (src/run.jl:316): 4-tile kernels
    assert pkg.sstep_partition(f, 28, 13, 20) == [13]
    assert pkg.sstep_partition(f, 29, 12, 20) == [12] and pkg.sstep_partition(f, 29, 15, 20) == [12, 3]
    assert pkg.sstep_partition(f, 33, 7, 20) == [7] and pkg.sstep_partition(f, 50, 14, 20) == [8, 6] and pkg.sstep_partition(f, 31, 30, 20) == [12, 12, 6]
    assert pkg.sstep_partition(c, 11, 9, 20) == [9] and pkg.sstep_partition(c, 6, 14, 20) == [10, 4]
    assert pkg.sstep_partition(c, 33, 7, 20) == [7] and pkg.sstep_partition(c, 45, 5, 20) == [5]   # ComplexF64: blocks of up to 8 up to 48 columns
    assert pkg.sstep_partition(c, 31, 30, 20) == [10, 8]         # ... and the partition ends where the kernels do: the other 12 steps run one at a time
    assert pkg.sstep_partition(c, 49, 3, 20) == []
    assert pkg.sstep_partition(f, 21, 20, 5) == [5, 5, 5, 5] and pkg.sstep_partition(f, 21, 20, 1) == [1] * 20


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_every_partition_covers_a_prefix_of_its_range(dtype):
    for k0 in range(1, 60):
        for count in range(1, 66 - k0):
            for smax in (2, 5, 8, 10, 13, 20):
                p = pkg.sstep_partition(dtype, k0, count, smax)
                if p:
                    assert sum(p) <= count and max(p) <= smax and min(p) >= 1, (k0, count, smax, p)
                    if np.dtype(dtype).kind == "f":
                        assert sum(p) == count, (k0, count, smax, p)      # Float64 kernels reach 64 columns: the whole range


def test_arguments_out_of_range_give_no_partition():
    assert pkg.sstep_partition(np.float64, 0, 5, 5) == [] and pkg.sstep_partition(np.float64, 5, 0, 5) == []
    assert pkg.sstep_partition(np.float64, 60, 10, 5) == [5] and pkg.sstep_partition(np.float64, 64, 3, 5) == [1]   # k + s <= 65 (maxdim <= 64): a prefix
