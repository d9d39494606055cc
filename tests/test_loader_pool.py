"""The loader's buffer recycling (BatchIterator.py _BufferPool): capacity buckets, reuse, and the bound on cached bytes
that keeps a data set with many distinct frame sizes (config/imagenet.lua) from growing device / pinned memory."""
import numpy as np

from frcnn_amd.BatchIterator import _BufferPool, _bucket


def test_bucket_rounding():
    for n in [1, 4096, 4097, 5000, 1 << 20, (1 << 20) + 1, 3 * 450 * 800 * 4, 3 * 1080 * 1920]:
        b = _bucket(n)
        assert b >= n and b >= 4096
        assert b <= max(4096, n) * 1.126, (n, b)          # at most 12.5 % slack
        assert _bucket(b) == b
    # near-equal frame sizes share a bucket
    assert _bucket(3 * 451 * 800 * 4) == _bucket(3 * 450 * 800 * 4 + 1000)


def test_pool_reuses_and_bounds_cached_bytes():
    allocs, frees = [], []
    pool = _BufferPool(lambda n: allocs.append(n) or ("buf", len(allocs), n), 10 << 20, release=lambda b: frees.append(b))
    a, ba = pool.take(1 << 20)
    pool.give(a, ba)
    b, bb = pool.take((1 << 20) - 100)
    assert b is a and len(allocs) == 1                        # recycled, no second allocation
    pool.give(b, bb)
    # 500 distinct frame sizes, each taken and handed back: the cache stays under the bound
    rng = np.random.RandomState(0)
    held = []
    for i in range(500):
        n = int(rng.randint(1 << 19, 4 << 20))
        buf, bk = pool.take(n)
        held.append((buf, bk))
        if len(held) > 3:                                     # a few frames in flight, like a batch
            pool.give(*held.pop(0))
        assert pool.cached <= pool.max_bytes
        assert pool.allocated <= pool.max_bytes + 4 * (4 << 20) * 1.13 + (1 << 20)
    assert frees, "nothing was ever released"
    assert len(allocs) < 500                                  # and buffers were recycled across sizes of one bucket
    # least recently used buckets go first: a bucket that is touched all the time survives
    hot, hb = pool.take(123456)
    for i in range(50):
        pool.give(hot, hb)
        big, bgb = pool.take(int(rng.randint(2 << 20, 4 << 20)))
        pool.give(big, bgb)
        got, gb = pool.take(123456)
        assert got is hot
