"""The reference's scene PRNG, vectorised.

Restates /root/reference/src/rand.c: Rand_Next (rand.c:7-13) is a 32-bit Weyl
sequence (state += 0xE120FC15) pushed through two multiply-xorshift rounds, so
the k-th output after seeding depends only on seed + k*increment and a whole
block can be produced at once.  Rand_Double (rand.c:29) and Rand_Int
(rand.c:21) are restated on top.  Seeded explicitly (the reference seeds from
time(NULL), main.c:328).
"""
import numpy as np

_INC = np.uint64(0xE120FC15)
_M1 = np.uint64(0x4A39B70D)
_M2 = np.uint64(0x12FAD5C9)
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


class Rand:
    def __init__(self, seed=0):
        self.state = int(seed) & 0xFFFFFFFF          # randState, rand.c:5

    def next(self, n=None):
        """Rand_Next(): one value, or a block of n values as uint32."""
        cnt = 1 if n is None else int(n)
        k = np.arange(1, cnt + 1, dtype=np.uint64)
        st = (np.uint64(self.state) + k * _INC) & _MASK
        t = st * _M1
        m1 = ((t >> _S32) ^ t) & _MASK
        t = m1 * _M2
        out = (((t >> _S32) ^ t) & _MASK).astype(np.uint32)
        self.state = int(st[-1])
        return int(out[0]) if n is None else out

    def double(self, lo, hi, n=None):
        """Rand_Double(min,max) = min + next / 0xFFFFFFFF * (max - min)."""
        r = self.next(n)
        if n is None:
            return lo + r / float(0xFFFFFFFF) * (hi - lo)
        return lo + r.astype(np.float64) / float(0xFFFFFFFF) * (hi - lo)

    def int(self, lo, hi):
        """Rand_Int(min,max) = next % (max - min) + min."""
        if lo >= hi:
            return 0
        return int(self.next() % (hi - lo)) + lo
