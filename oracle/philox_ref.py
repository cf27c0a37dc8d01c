"""oracle/philox_ref.py -- TEST INFRASTRUCTURE.  The uniform stream of cagpu_generate_cases restated in Python:
Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; constants of Random123), key = the
64-bit seed, counter = (draw block, 0, case index lo, hi), two 53-bit doubles per block built like numpy's legacy
random_sample ((a >> 5) * 2^26 + (b >> 6)) / 2^53.  `PhiloxStream` stands in for `np.random` inside
envs/scenario_generator.py so that the HOST generator (bit-identical to the reference under np.random,
tests/golden/rand_cases.npz) produces what the device generator must produce."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(counter, key):
    c0, c1, c2, c3 = counter
    k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0, c1, c2, c3


class PhiloxStream(object):
    """rand() / rand(n) with the draw order of np.random.rand, for ONE case index"""

    def __init__(self, seed, case):
        self.key = (seed & MASK, (seed >> 32) & MASK)
        self.case = (case & MASK, (case >> 32) & MASK)
        self.block, self.buf = 0, []
        self.draws = 0

    def _one(self):
        if not self.buf:
            w = philox4x32_10((self.block, 0, self.case[0], self.case[1]), self.key)
            self.block += 1
            self.buf = [((w[0] >> 5) * 67108864.0 + (w[1] >> 6)) / 9007199254740992.0,
                        ((w[2] >> 5) * 67108864.0 + (w[3] >> 6)) / 9007199254740992.0]
        self.draws += 1
        return self.buf.pop(0)

    def rand(self, *shape):
        if not shape:
            return self._one()
        return np.array([self._one() for _ in range(int(np.prod(shape)))]).reshape(shape)
