"""The magic-number division of conv_f32.hip / resunit_f32.hip (make_fdiv: n / d == (n * mul) >> shift for 0 <= n < 2^26,
mul = ceil(2^k / d), k = 26 + ceil(log2 d)) restated in Python and checked exhaustively near every boundary: the kernels
derive tile coordinates from it on the scalar unit, so a wrong quotient would silently mis-place a tile."""
import random


def make_fdiv(d):
    lg = 0
    while (1 << lg) < d:
        lg += 1
    k = 26 + lg
    return ((1 << k) + d - 1) // d, k


def test_fastdiv_is_exact_below_2_pow_26():
    rng = random.Random(7)
    ds = list(range(1, 300)) + [2 ** i + j for i in range(2, 26) for j in (-1, 0, 1)] + [rng.randrange(1, 1 << 26) for _ in range(300)]
    for d in ds:
        mul, k = make_fdiv(d)
        assert mul < (1 << 32) and k < 64
        ns = {0, 1, d - 1, d, d + 1, (1 << 26) - 1}
        ns |= {q * d + r for q in (1, 2, 3, 1000, ((1 << 26) - 1) // d) for r in (-1, 0, 1)}
        ns |= {rng.randrange(0, 1 << 26) for _ in range(50)}
        for n in ns:
            if 0 <= n < (1 << 26):
                assert (n * mul) >> k == n // d, (n, d)
