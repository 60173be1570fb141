"""Reduced Tate pairing and check_same_ratio on Python ints (TEST INFRASTRUCTURE — oracle only).

Restates the boolean behaviour of setup_utils::check_same_ratio (`E::pairing(a, d) == E::pairing(b, c)`;
reference call sites /root/reference/src/bin/verify_transcript.rs:812-822, src/bin/control.rs:587-591).
arkworks computes an optimal-ate pairing; ANY non-degenerate bilinear pairing on G1 x G2 gives the same
boolean, so one generic construction serves all four curves:

  Fq^k = Fq[w]/(w^k - nu), the G2 twist field embedded by u -> w^(k/e), the untwist
  psi(x', y') = (x' w^(+-2), y' w^(+-3)), Miller loop over the group order with P in E(Fq) (affine, slopes in
  Fq), denominator elimination (x(psi(Q)) lies in Fq^(k/2)), final exponentiation (q^k - 1)/r split into the
  Frobenius-computable part and the hard part Phi_k(q)/r.
PARITY UNPINNED (see oracle/c/ptau_oracle.h).  Self-checks: bilinearity in tests/test_oracle_pairing.py.
"""
from .curves import Curve
from .ec import Group

# (embedding degree k, nu with w^k = nu, untwist sign: +1 -> (x' w^2, y' w^3), -1 -> (x'/w^2, y'/w^3))
PAIRING_PARAMS = {
    "bls12_377": (12, -5, +1),
    "bw6": (6, -4, -1),
    "mnt4_753": (4, 13, -1),
    "mnt6_753": (6, 11, -1),
}


def cyclotomic(k, q):
    return {4: q * q + 1, 6: q * q - q + 1, 12: q ** 4 - q * q + 1}[k]


class ExtK:
    """Fq[w]/(w^k - nu), elements = lists of k ints."""

    def __init__(self, q, k, nu):
        self.q, self.k, self.nu = q, k, nu % q
        assert (q - 1) % k == 0
        self.gamma1 = pow(self.nu, (q - 1) // k, q)      # w^q = gamma1 * w

    def one(self):
        return [1] + [0] * (self.k - 1)

    def mul(self, a, b):
        k, q, nu = self.k, self.q, self.nu
        t = [0] * (2 * k - 1)
        for i, ai in enumerate(a):
            if ai:
                for j, bj in enumerate(b):
                    if bj:
                        t[i + j] += ai * bj
        return [(t[i] + (nu * t[i + k] if i + k < 2 * k - 1 else 0)) % q for i in range(k)]

    def sqr(self, a):
        return self.mul(a, a)

    def frobenius(self, a, j):
        """a^(q^j): coefficient i times gamma1^(i*j)."""
        g = pow(self.gamma1, j, self.q)
        out, c = [], 1
        for ai in a:
            out.append(ai * c % self.q)
            c = c * g % self.q
        return out

    def inv(self, a):
        """Solve (multiplication-by-a matrix) x = e0 over Fq."""
        k, q, nu = self.k, self.q, self.nu
        # column j of M = a * w^j
        M = [[0] * (k + 1) for _ in range(k)]
        col = list(a)
        for j in range(k):
            for i in range(k):
                M[i][j] = col[i]
            col = [nu * col[k - 1] % q] + col[:k - 1]
        M[0][k] = 1
        for c in range(k):
            piv = next(r for r in range(c, k) if M[r][c])
            M[c], M[piv] = M[piv], M[c]
            iv = pow(M[c][c], -1, q)
            M[c] = [v * iv % q for v in M[c]]
            for r in range(k):
                if r != c and M[r][c]:
                    f = M[r][c]
                    M[r] = [(vr - f * vc) % q for vr, vc in zip(M[r], M[c])]
        return [M[i][k] for i in range(k)]

    def pow(self, a, e):
        r = self.one()
        for bit in bin(e)[2:]:
            r = self.sqr(r)
            if bit == "1":
                r = self.mul(r, a)
        return r


class Pairing:
    def __init__(self, curve: Curve):
        self.curve = curve
        self.k, nu, self.sign = PAIRING_PARAMS[curve.name]
        self.K = ExtK(curve.q, self.k, nu)
        self.G1, self.G2 = Group(curve, 1), Group(curve, 2)
        self.e = self.G2.F.deg
        assert (nu - (self.G2.F.nr if self.e > 1 else nu)) % curve.q == 0     # w^k must equal the twist field's non-residue
        self.hard = cyclotomic(self.k, curve.q)
        assert self.hard % curve.r == 0
        self.hard //= curve.r

    def untwist(self, Q):
        """G2 affine point on the twist -> (x, y) in Fq^k (coefficient lists)."""
        k, e, q = self.k, self.e, self.curve.q
        step = k // e

        def embed(el, shift):
            out = [0] * k
            for i, c in enumerate(el):
                idx = i * step + shift
                coef = c
                while idx < 0:           # w^-1 = w^(k-1) / nu
                    idx += k
                    coef = coef * pow(self.K.nu, -1, q) % q
                while idx >= k:
                    idx -= k
                    coef = coef * self.K.nu % q
                out[idx] = (out[idx] + coef) % q
            return out
        x, y = Q
        return embed(x, 2 * self.sign), embed(y, 3 * self.sign)

    def miller(self, P, Q):
        """f_{r,P}(psi(Q)) without the final exponentiation; P in G1 (affine Fq), Q in G2 (twist)."""
        if P is None or Q is None:
            return self.K.one()
        K, q, a = self.K, self.curve.q, self.G1.a[0]
        xq, yq = self.untwist(Q)
        xp, yp = P[0][0], P[1][0]
        xt, yt = xp, yp
        f = K.one()

        def line(lam, x0, y0):
            # (y_Q - y0) - lam (x_Q - x0)
            l = [(yc - lam * xc) % q for yc, xc in zip(yq, xq)]
            l[0] = (l[0] - y0 + lam * x0) % q
            return l
        for bit in bin(self.curve.r)[3:]:
            lam = (3 * xt * xt + a) * pow(2 * yt, -1, q) % q
            f = K.mul(K.sqr(f), line(lam, xt, yt))
            x3 = (lam * lam - 2 * xt) % q
            yt = (lam * (xt - x3) - yt) % q
            xt = x3
            if bit == "1":
                if xt == xp:             # T = -P: vertical line, eliminated; only at the last step (r*P = O)
                    assert (yt + yp) % q == 0
                    xt = yt = None
                    continue
                lam = (yt - yp) * pow(xt - xp, -1, q) % q
                f = K.mul(f, line(lam, xt, yt))
                x3 = (lam * lam - xt - xp) % q
                yt = (lam * (xt - x3) - yt) % q
                xt = x3
        assert xt is None, "r*P != O"
        return f

    def final_exp(self, f):
        K, k = self.K, self.k
        finv = K.inv(f)
        if k == 12:
            t = K.mul(K.frobenius(f, 6), finv)
            t = K.mul(K.frobenius(t, 2), t)
        elif k == 6:
            t = K.mul(K.frobenius(f, 3), finv)
            t = K.mul(K.frobenius(t, 1), t)
        else:
            t = K.mul(K.frobenius(f, 2), finv)
        return K.pow(t, self.hard)

    def pairing(self, P, Q):
        return self.final_exp(self.miller(P, Q))

    def check_same_ratio(self, g1_pair, g2_pair) -> bool:
        """setup_utils::check_same_ratio((a, b), (c, d)): e(a, d) == e(b, c), as e(a, d) * e(-b, c) == 1."""
        (a, b), (c, d) = g1_pair, g2_pair
        if a is None or b is None or c is None or d is None:
            return False
        f = self.K.mul(self.miller(a, d), self.miller(self.G1.neg(b), c))
        return self.final_exp(f) == self.K.one()
