"""Philox4x32-10 in pure Python, for the handful of draws scene construction needs.

The reference builds its random scenes from an OS-seeded generator (main.rs:58-73,687,778; perlin.rs:19), so no two
runs of it agree.  Here scene randomness comes from the same counter-based generator the renderer uses, on its own
stream (counter word 3 = 1), so a seed names one scene."""

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return c0, c1, c2, c3


class SceneRng:
    """Sequential uniform draws; range(a, b) = U * (b - a) + a with 52 random mantissa bits, as rand 0.9 does."""

    def __init__(self, seed, stream=1):
        self.key = (seed & MASK, (seed >> 32) & MASK)
        self.stream = stream
        self.ctr = 0

    def next01(self):
        w = philox4x32_10((0, 0, self.ctr >> 1, self.stream), self.key)
        bits = (w[3] << 32 | w[2]) if (self.ctr & 1) else (w[1] << 32 | w[0])
        self.ctr += 1
        return float(bits >> 12) * (1.0 / 4503599627370496.0)

    def range(self, a, b):
        return self.next01() * (b - a) + a

    def vector(self, a, b):  # random_vector, point3/mod.rs:229-235
        x = self.range(a, b)
        y = self.range(a, b)
        z = self.range(a, b)
        return (x, y, z)
