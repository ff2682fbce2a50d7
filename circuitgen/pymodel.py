"""Tiny pure-Python (big-int) models used to cross-check the C oracle on small cases."""
import json
import os

P = 0xFFFFFFFF00000001
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]

_rc = None


def round_constants():
    global _rc
    if _rc is None:
        import re
        txt = open(os.path.join(ROOT, "oracle", "poseidon_constants.h")).read()
        _rc = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ULL", txt)]
        assert len(_rc) == 360
    return _rc


def poseidon(state):
    rc = round_constants()
    s = [int(x) for x in state]
    for r in range(30):
        s = [(s[i] + rc[12 * r + i]) % P for i in range(12)]
        if r < 4 or r >= 26:
            s = [pow(x, 7, P) for x in s]
        else:
            s[0] = pow(s[0], 7, P)
        s = [(sum(s[(i + row) % 12] * CIRC[i] for i in range(12)) + (8 * s[0] if row == 0 else 0)) % P for row in range(12)]
    return s


def hash_no_pad(xs):
    s = [0] * 12
    xs = [int(x) for x in xs]
    for off in range(0, len(xs), 8):
        chunk = xs[off:off + 8]
        s[:len(chunk)] = chunk
        s = poseidon(s)
    return s[:4]


def root_of_unity(k):
    g = 1753635133440165772
    for _ in range(k, 32):
        g = g * g % P
    return g


def ext_mul(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


class Challenger:
    """iop/challenger.rs restated."""

    def __init__(self):
        self.state, self.inp, self.out = [0] * 12, [], []

    def _duplex(self):
        self.state[:len(self.inp)] = self.inp
        self.inp = []
        self.state = poseidon(self.state)
        self.out = self.state[:8]

    def observe(self, xs):
        for x in xs:
            self.out = []
            self.inp.append(int(x))
            if len(self.inp) == 8:
                self._duplex()

    def get(self):
        if self.inp or not self.out:
            self._duplex()
        return self.out.pop()
