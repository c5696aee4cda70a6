"""The multiply-high division the conv prologues and the persistent GEMM's tile decode use (csrc/gemm_common.h `udivmod_m`,
csrc/conv_s1.hip `fast_divmod`): q = mulhi(n, m), r = n - q d, ONE correction step, with m = min(floor(2^32 / d), 2^32 - 1).
This restates the arithmetic on the host and checks it against divmod over the ranges the kernels use it on (n < 2^32, any
d >= 1, including the small group sizes 1..8 of the tile raster and d = 1)."""
import numpy as np


def magic(d: int) -> int:
    return 0xFFFFFFFF if d <= 1 else (1 << 32) // d


def fast_divmod(n: np.ndarray, d: int):
    m = magic(d)
    q = (n.astype(np.uint64) * np.uint64(m)) >> np.uint64(32)
    r = n.astype(np.int64) - q.astype(np.int64) * d
    fix = r >= d
    q = q + fix.astype(np.uint64)
    r = r - fix.astype(np.int64) * d
    return q.astype(np.int64), r


def test_multiply_high_division_is_exact_with_one_correction():
    rng = np.random.default_rng(7)
    ds = list(range(1, 70)) + [196, 200, 784, 3136, 3137, 12544, 65535, 65536, 65537, (1 << 31) - 1, (1 << 31), (1 << 32) - 1]
    ds += [int(x) for x in rng.integers(1, 1 << 31, 200)]
    for d in ds:
        n = np.concatenate([rng.integers(0, 1 << 32, 4000, dtype=np.uint64),
                            np.array([0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 32) - 1, ((1 << 32) // d) * d - 1], dtype=np.uint64) % (1 << 32)])
        q, r = fast_divmod(n, d)
        assert np.array_equal(q, (n // np.uint64(d)).astype(np.int64)), d
        assert np.array_equal(r, (n % np.uint64(d)).astype(np.int64)), d


def test_group_size_reciprocals_of_the_tile_raster():
    # gemm256p_kernel.h decode(): floor(2^32 / gsz) for gsz = 1 .. 8 as literals
    table = {8: 0x20000000, 7: 0x24924924, 6: 0x2AAAAAAA, 5: 0x33333333, 4: 0x40000000, 3: 0x55555555, 2: 0x80000000, 1: 0xFFFFFFFF}
    for g, m in table.items():
        assert m == magic(g), g
