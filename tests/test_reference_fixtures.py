"""The only reference-held bytes that touch this path's serialisation: /root/reference/e2e/circuit_* (committed verbatim
under tests/golden/reference_e2e/).  They pin the canonical little-endian layout and the byte length of an ark-serialize
field element per curve — everything else about the oracle stays "parity unpinned" (DESIGN.md §2)."""
import os
import struct

import pytest

from conftest import CURVE_NAMES, ROOT
from oracle.pyref.curves import CURVES

FIX = os.path.join(ROOT, "tests", "golden", "reference_e2e")


def _read_fr(curve, data, off):
    """ark-serialize Fp::serialize_compressed: ceil(bits / 8) little-endian bytes of the canonical value (no flags)."""
    n = curve.fr_bytes
    v = int.from_bytes(data[off:off + n], "little")
    assert v < curve.r, "non-canonical scalar in a reference fixture"
    return v, off + n


def _parse(curve, data):
    hdr = struct.unpack("<6Q", data[:48])
    off = 48
    mats = []
    for _ in range(3):
        (rows,) = struct.unpack_from("<Q", data, off)
        off += 8
        m = []
        for _ in range(rows):
            (k,) = struct.unpack_from("<Q", data, off)
            off += 8
            row = []
            for _ in range(k):
                v, off = _read_fr(curve, data, off)
                (idx,) = struct.unpack_from("<Q", data, off)
                off += 8
                row.append((v, idx))
            m.append(row)
        mats.append(m)
    assert off == len(data), "fixture not consumed exactly: wrong field element size"
    return hdr, mats


@pytest.mark.parametrize("name", CURVE_NAMES)
def test_reference_circuit_files_pin_the_scalar_field_encoding(name, pkg):
    c = CURVES[name]
    data = open(os.path.join(FIX, "circuit_" + name), "rb").read()
    hdr, (a, b, cm) = _parse(c, data)
    n_inst, n_wit, n_cons, nza, nzb, nzc = hdr
    assert (n_inst, n_wit, n_cons) == (2, 1, 6)
    assert [sum(len(r) for r in m) for m in (a, b, cm)] == [nza, nzb, nzc]
    assert all(len(m) == n_cons for m in (a, b, cm))
    assert all(idx < n_inst + n_wit for m in (a, b, cm) for r in m for (_, idx) in r)
    assert a[0][0] == (1, 2) and all(v == 1 for m in (a, b, cm) for r in m for (v, _) in r)   # value 1 encodes as 01 00 .. 00: little-endian, canonical (not Montgomery)
    # sizes: 32 (BLS12-377 Fr), 48 (BW6-761 Fr = BLS12-377 Fq), 95 (the 753-bit fields: ceil(753 / 8), NOT 96)
    assert c.fr_bytes == {"bls12_377": 32, "bw6": 48, "mnt4_753": 95, "mnt6_753": 95}[name]
    # the same field types are base fields elsewhere in the path: BW6-761's Fr is BLS12-377's Fq, the MNT curves form a
    # cycle — so these files also pin the coordinate size the product and the oracle use for those base fields
    L = pkg.load_library()
    partner = {"bw6": "bls12_377", "mnt4_753": "mnt6_753", "mnt6_753": "mnt4_753"}.get(name)
    if partner:
        assert CURVES[partner].q == c.r and CURVES[partner].fq_bytes == c.fr_bytes
        assert L.ptau_point_size(CURVE_NAMES.index(partner), 1, 1) == c.fr_bytes
    assert L.ptau_fr_limbs(CURVE_NAMES.index(name)) == (c.r.bit_length() + 63) // 64
