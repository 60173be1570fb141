"""Prime and small-extension field arithmetic on Python ints (TEST INFRASTRUCTURE — oracle only).

Restates, at specification level, ark-ff 0.4.2 `Fp`, `QuadExtField`, `CubicExtField`
(un-vendored; Cargo.lock:243-246).  Elements of Fq^d = Fq[u]/(u^d - nr) are tuples
(c0, .., c_{d-1}) of canonical ints; d = 1 uses 1-tuples so one code path serves G1 and G2.
"""

from typing import Tuple

El = Tuple[int, ...]


class Field:
    def __init__(self, q: int, deg: int = 1, nonresidue: int = 0):
        assert deg in (1, 2, 3)
        self.q, self.deg, self.nr = q, deg, nonresidue % q if deg > 1 else 0
        self.order = q ** deg

    # -- constructors
    def zero(self) -> El:
        return (0,) * self.deg

    def one(self) -> El:
        return (1,) + (0,) * (self.deg - 1)

    def from_base(self, c: int) -> El:
        return (c % self.q,) + (0,) * (self.deg - 1)

    def is_zero(self, a: El) -> bool:
        return not any(a)

    # -- ring ops
    def add(self, a: El, b: El) -> El:
        q = self.q
        return tuple((x + y) % q for x, y in zip(a, b))

    def sub(self, a: El, b: El) -> El:
        q = self.q
        return tuple((x - y) % q for x, y in zip(a, b))

    def neg(self, a: El) -> El:
        q = self.q
        return tuple((-x) % q for x in a)

    def mul(self, a: El, b: El) -> El:
        q, nr = self.q, self.nr
        if self.deg == 1:
            return (a[0] * b[0] % q,)
        if self.deg == 2:
            a0, a1 = a
            b0, b1 = b
            return ((a0 * b0 + nr * a1 * b1) % q, (a0 * b1 + a1 * b0) % q)
        a0, a1, a2 = a
        b0, b1, b2 = b
        return ((a0 * b0 + nr * (a1 * b2 + a2 * b1)) % q,
                (a0 * b1 + a1 * b0 + nr * a2 * b2) % q,
                (a0 * b2 + a1 * b1 + a2 * b0) % q)

    def sqr(self, a: El) -> El:
        return self.mul(a, a)

    def mul_base(self, a: El, c: int) -> El:
        q = self.q
        return tuple(x * c % q for x in a)

    def inv(self, a: El) -> El:
        q, nr = self.q, self.nr
        if self.deg == 1:
            return (pow(a[0], -1, q),)
        if self.deg == 2:
            a0, a1 = a
            n = pow(a0 * a0 - nr * a1 * a1, -1, q)
            return (a0 * n % q, (-a1) * n % q)
        a0, a1, a2 = a
        # adjugate method for Fq3 (same as ark-ff CubicExtField::inverse)
        t0 = (a0 * a0 - nr * a1 * a2) % q
        t1 = (nr * a2 * a2 - a0 * a1) % q
        t2 = (a1 * a1 - a0 * a2) % q
        n = pow(a0 * t0 + nr * (a2 * t1 + a1 * t2), -1, q)
        return (t0 * n % q, t1 * n % q, t2 * n % q)

    def pow(self, a: El, e: int) -> El:
        if self.deg == 1:
            return (pow(a[0], e, self.q),)
        res = self.one()
        for bit in bin(e)[2:]:
            res = self.sqr(res)
            if bit == "1":
                res = self.mul(res, a)
        return res

    # -- ordering used by arkworks for the y-sign flag.
    # ark-ff Ord for Fp compares canonical integers; QuadExtField / CubicExtField
    # compare the HIGHEST coefficient first (c1 then c0; c2, c1, c0).
    def cmp_key(self, a: El):
        return tuple(reversed(a))

    def is_square(self, a: El) -> bool:
        if self.is_zero(a):
            return True
        return self.pow(a, (self.order - 1) // 2) == self.one()

    def sqrt(self, a: El):
        """Some square root of `a` or None (Tonelli-Shanks over the full field order).
        Which root is returned is irrelevant: callers pick by the sign flag."""
        if self.is_zero(a):
            return self.zero()
        if not self.is_square(a):
            return None
        n = self.order - 1
        s = (n & -n).bit_length() - 1
        t = n >> s
        z = self._nonsquare()
        c = self.pow(z, t)
        x = self.pow(a, (t + 1) // 2)
        b = self.pow(a, t)
        m = s
        one = self.one()
        while b != one:
            i, b2 = 0, b
            while b2 != one:
                b2 = self.sqr(b2)
                i += 1
            e = c
            for _ in range(m - i - 1):
                e = self.sqr(e)
            x = self.mul(x, e)
            c = self.sqr(e)
            b = self.mul(b, c)
            m = i
        assert self.sqr(x) == a
        return x

    _ns_cache = None

    def _nonsquare(self) -> El:
        if self._ns_cache is None:
            k = 1
            while True:
                cand = tuple((k + i) % self.q for i in range(self.deg))
                if not self.is_square(cand):
                    self._ns_cache = cand
                    break
                k += 1
        return self._ns_cache
