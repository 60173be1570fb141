"""Short-Weierstrass affine group law on Python ints (TEST INFRASTRUCTURE — oracle only).

Specification-level restatement of ark-ec 0.4.2 `short_weierstrass::{Affine, Projective}`
(un-vendored; Cargo.lock:172-176).  Results are canonical affine points, so any correct
scalar-mul algorithm yields the same bytes as the reference's Jacobian double-and-add
(`sw_double_and_add_affine`) followed by `batch_normalization`.
A point is `None` (infinity) or `(x, y)` with x, y field tuples.
"""

from .curves import Curve, GroupParams
from .field import Field


class Group:
    def __init__(self, curve: Curve, which: int):
        self.curve, self.which = curve, which
        gp: GroupParams = curve.group(which)
        self.F = Field(curve.q, gp.deg, gp.nonresidue)
        self.a, self.b = tuple(gp.a), tuple(gp.b)
        self.gen = (tuple(gp.gen[0]), tuple(gp.gen[1]))
        self.r = curve.r
        self.comp_size = curve.comp_size(which)
        self.uncomp_size = curve.uncomp_size(which)

    def on_curve(self, P) -> bool:
        if P is None:
            return True
        F = self.F
        x, y = P
        rhs = F.add(F.add(F.mul(F.sqr(x), x), F.mul(self.a, x)), self.b)
        return F.sqr(y) == rhs

    def neg(self, P):
        return None if P is None else (P[0], self.F.neg(P[1]))

    def double(self, P):
        if P is None:
            return None
        F = self.F
        x, y = P
        if F.is_zero(y):
            return None
        x2 = F.sqr(x)
        num = F.add(F.add(F.add(x2, x2), x2), self.a)
        lam = F.mul(num, F.inv(F.add(y, y)))
        x3 = F.sub(F.sub(F.sqr(lam), x), x)
        y3 = F.sub(F.mul(lam, F.sub(x, x3)), y)
        return (x3, y3)

    def add(self, P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        F = self.F
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if y1 == y2:
                return self.double(P)
            return None
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
        x3 = F.sub(F.sub(F.sqr(lam), x1), x2)
        y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
        return (x3, y3)

    def mul(self, P, k: int):
        """k*P, MSB-first double-and-add (ark-ec `sw_double_and_add_affine`).  k is NOT reduced:
        callers pass canonical Fr integers (or cofactors / r for subgroup tests)."""
        if k < 0:
            return self.mul(self.neg(P), -k)
        res = None
        for bit in bin(k)[2:] if k else "":
            res = self.double(res)
            if bit == "1":
                res = self.add(res, P)
        return res

    def in_subgroup(self, P) -> bool:
        return self.mul(P, self.r) is None
