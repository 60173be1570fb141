"""arkworks-0.4 canonical point (de)serialisation (TEST INFRASTRUCTURE — oracle only).

Restates ark-serialize 0.4.2 / ark-ec 0.4.2 `Affine::serialize_with_mode`,
`SWFlags` and ark-ff `Fp::serialize_with_flags` (un-vendored; Cargo.lock:331-334), as used by
setup-utils' `Serializer/Deserializer for [u8]` — reference call sites
/root/reference/src/bin/contribute.rs:816-819, 971-980 (UseCompression / CheckForCorrectness).
Layout: SURVEY.md App. A.2.
"""

from .ec import Group

FLAG_NEG = 0x80       # SWFlags::YIsNegative
FLAG_INF = 0x40       # SWFlags::PointAtInfinity


def _fq_bytes(G: Group) -> int:
    return G.curve.fq_bytes


def _el_to_bytes(G: Group, el, flags: int = 0) -> bytes:
    n = _fq_bytes(G)
    out = bytearray()
    for c in el:
        out += int(c).to_bytes(n, "little")
    out[-1] |= flags
    return bytes(out)


def _el_from_bytes(G: Group, data: bytes, with_flags: bool):
    n = _fq_bytes(G)
    d = G.F.deg
    assert len(data) == n * d
    buf = bytearray(data)
    flags = 0
    if with_flags:
        flags = buf[-1] & 0xC0
        buf[-1] &= 0x3F
    el = tuple(int.from_bytes(buf[i * n:(i + 1) * n], "little") for i in range(d))
    if any(c >= G.curve.q for c in el):
        raise ValueError("non-canonical field element")
    return el, flags


def y_is_negative(G: Group, y) -> bool:
    """ark-ec `to_flags`: positive iff y <= -y in the field's Ord."""
    F = G.F
    return F.cmp_key(y) > F.cmp_key(F.neg(y))


def write_point(G: Group, P, compressed: bool) -> bytes:
    F = G.F
    if P is None:
        x, y, flags = F.zero(), F.zero(), FLAG_INF
    else:
        x, y = P
        flags = FLAG_NEG if y_is_negative(G, y) else 0
    if compressed:
        return _el_to_bytes(G, x, flags)
    return _el_to_bytes(G, x) + _el_to_bytes(G, y, flags)


def read_point(G: Group, data: bytes, compressed: bool):
    """Validate::No read: no on-curve / subgroup check (done separately, CheckForCorrectness)."""
    F = G.F
    if compressed:
        x, flags = _el_from_bytes(G, data, True)
        if flags & FLAG_INF:
            return None
        rhs = F.add(F.add(F.mul(F.sqr(x), x), F.mul(G.a, x)), G.b)
        y = F.sqrt(rhs)
        if y is None:
            raise ValueError("x is not on the curve")
        ny = F.neg(y)
        lo, hi = (y, ny) if F.cmp_key(y) <= F.cmp_key(ny) else (ny, y)
        return (x, hi if flags & FLAG_NEG else lo)
    half = len(data) // 2
    x, _ = _el_from_bytes(G, data[:half], False)
    y, flags = _el_from_bytes(G, data[half:], True)
    if flags & FLAG_INF:
        return None
    return (x, y)


def read_batch(G: Group, data: bytes, compressed: bool):
    sz = G.comp_size if compressed else G.uncomp_size
    assert len(data) % sz == 0
    return [read_point(G, data[i:i + sz], compressed) for i in range(0, len(data), sz)]


def write_batch(G: Group, pts, compressed: bool) -> bytes:
    return b"".join(write_point(G, P, compressed) for P in pts)
