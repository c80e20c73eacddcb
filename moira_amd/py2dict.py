"""Iteration order of a CPython-2.7 dict keyed by str, reproduced.

Why this exists: moira collapses identical sequences in a dict and writes the groups with
`sorted(uniques, key=abundance, reverse=True)` (moira/moira.py:492).  The sort is stable, so
groups of equal abundance come out in *dict iteration order*, which in Python 2.7 is the slot
order of an open-addressing table driven by the (unrandomised) string hash.  The reference's
golden output files (moira/test/test_results/*.names, *.fasta) bake that order in, so reproducing
them byte for byte needs the same order.  This is a restatement of CPython 2.7's documented
algorithm (Objects/dictobject.c: PyDict_MINSIZE 8, PERTURB_SHIFT 5, resize at 2/3 fill to
4 x used, or 2 x used above 50000; Objects/stringobject.c: string_hash), insert-only.
"""

_M = (1 << 64) - 1


def py2_str_hash(b):
    """hash(str) of 64-bit CPython 2.7 without -R, as an unsigned 64-bit value."""
    if not b:
        return 0
    x = (b[0] << 7) & _M
    for c in b:
        x = ((1000003 * x) & _M) ^ c
    x ^= len(b)
    if x == _M:          # -1 is reserved
        x = _M - 1
    return x


class Py2Dict:
    """Insert-only dict that remembers where CPython 2.7 would have put each key."""

    def __init__(self):
        self.mask = 7
        self.slots = [None] * 8          # (hash, key)
        self.used = 0
        self.values = {}

    def __contains__(self, key):
        return key in self.values

    def __getitem__(self, key):
        return self.values[key]

    def __len__(self):
        return self.used

    def get(self, key, default=None):
        return self.values.get(key, default)

    def insert(self, key, value, h):
        """__setitem__ for a NEW key whose hash is already known (computed in C for a whole chunk)."""
        self.values[key] = value
        self._place(h, key)
        self.used += 1
        if self.used * 3 >= (self.mask + 1) * 2:
            self._resize((2 if self.used > 50000 else 4) * self.used)

    def _place(self, h, key):
        mask, slots = self.mask, self.slots
        i = h & mask
        if slots[i] is None:
            slots[i] = (h, key)
            return
        perturb = h
        while True:
            i = ((i << 2) + i + perturb + 1) & _M
            j = i & mask
            if slots[j] is None:
                slots[j] = (h, key)
                return
            perturb >>= 5

    def _resize(self, minused):
        newsize = 8
        while newsize <= minused:
            newsize <<= 1
        old = self.slots
        self.mask = newsize - 1
        self.slots = [None] * newsize
        for e in old:                    # old slot order, as dictresize() re-inserts
            if e is not None:
                self._place(e[0], e[1])

    def __setitem__(self, key, value):
        if key in self.values:
            self.values[key] = value
            return
        self.values[key] = value
        b = key.encode("latin-1") if isinstance(key, str) else key
        self._place(py2_str_hash(b), key)
        self.used += 1
        if self.used * 3 >= (self.mask + 1) * 2:        # fill == used: nothing is ever deleted
            self._resize((2 if self.used > 50000 else 4) * self.used)

    def __iter__(self):
        for e in self.slots:
            if e is not None:
                yield e[1]
