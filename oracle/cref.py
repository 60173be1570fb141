"""ctypes binding of the C oracle (oracle/c/libptau_oracle.so) — TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "libptau_oracle.so")
_lib = None

CURVE_IDS = {"bls12_377": 0, "bw6": 1, "mnt4_753": 2, "mnt6_753": 3}


def build(force=False):
    src = os.path.join(_HERE, "c", "ptau_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "c"), "-B"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        u8p, u64p = ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)
        L.oracle_blake2b512.argtypes = [u8p, ctypes.c_size_t, u8p]
        L.oracle_blake2b512.restype = None
        L.oracle_params.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, u64p]
        L.oracle_apply_powers.argtypes = [ctypes.c_int, ctypes.c_int, u8p, ctypes.c_int, u8p, ctypes.c_int,
                                          ctypes.c_uint64, ctypes.c_uint64, u64p, u64p, ctypes.c_int]
        L.oracle_computation.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                         u8p, ctypes.c_size_t, u8p, ctypes.c_size_t, u64p, u64p, u64p, ctypes.c_int]
        L.oracle_decompress.argtypes = [ctypes.c_int, ctypes.c_int, u8p, u8p, ctypes.c_uint64, ctypes.c_int]
        L.oracle_scalar_mul.argtypes = [ctypes.c_int, ctypes.c_int, u8p, ctypes.c_int, u8p, ctypes.c_int,
                                        ctypes.c_uint64, u64p, ctypes.c_int, ctypes.c_int]
        L.oracle_msm.argtypes = [ctypes.c_int, ctypes.c_int, u8p, ctypes.c_int, ctypes.c_uint64, u64p, ctypes.c_int,
                                 u8p, ctypes.c_int]
        _lib = L
    return _lib


def limbs(x: int, n: int):
    return (ctypes.c_uint64 * n)(*[(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)])


def _cid(curve):
    return curve if isinstance(curve, int) else CURVE_IDS[curve]


def blake2b512(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(64)
    lib().oracle_blake2b512(data, len(data), out)
    return out.raw


def point_size(curve, group, compressed):
    return lib().oracle_point_size(_cid(curve), group, int(compressed))


def fr_limbs(curve):
    return lib().oracle_fr_limbs(_cid(curve))


def params(curve, mode, chunk_index, chunk_size, power):
    out = (ctypes.c_uint64 * 6)()
    rc = lib().oracle_params(_cid(curve), mode, chunk_index, chunk_size, power, out)
    if rc:
        raise ValueError("oracle_params rc=%d" % rc)
    return dict(zip(("g1", "other", "accumulator_size", "contribution_size", "public_key_size", "start"), out))


def apply_powers(curve, group, data: bytes, n, start, tau, coeff=None, compressed_in=False, compressed_out=True,
                 nthreads=1) -> bytes:
    c = _cid(curve)
    nl = fr_limbs(c)
    out = ctypes.create_string_buffer(n * point_size(c, group, compressed_out))
    rc = lib().oracle_apply_powers(c, group, data, int(compressed_in), out, int(compressed_out), n, start,
                                   limbs(tau, nl), None if coeff is None else limbs(coeff, nl), nthreads)
    if rc:
        raise ValueError("oracle_apply_powers rc=%d" % rc)
    return out.raw


def computation(curve, mode, chunk_index, chunk_size, power, challenge: bytes, tau, alpha, beta, nthreads=1) -> bytes:
    """Returns a response-sized buffer with only the element region filled."""
    c = _cid(curve)
    nl = fr_limbs(c)
    p = params(c, mode, chunk_index, chunk_size, power)
    out = ctypes.create_string_buffer(p["contribution_size"])
    rc = lib().oracle_computation(c, mode, chunk_index, chunk_size, power, challenge, len(challenge), out, len(out),
                                  limbs(tau, nl), limbs(alpha, nl), limbs(beta, nl), nthreads)
    if rc:
        raise ValueError("oracle_computation rc=%d" % rc)
    return out.raw


def decompress(curve, group, data: bytes, n, nthreads=1) -> bytes:
    c = _cid(curve)
    out = ctypes.create_string_buffer(n * point_size(c, group, False))
    rc = lib().oracle_decompress(c, group, data, out, n, nthreads)
    if rc:
        raise ValueError("oracle_decompress rc=%d" % rc)
    return out.raw


def scalar_mul(curve, group, data: bytes, n, scalars, limbs_per_scalar, shared=False, compressed_in=False,
               compressed_out=False) -> bytes:
    c = _cid(curve)
    out = ctypes.create_string_buffer(n * point_size(c, group, compressed_out))
    flat = []
    for s in ([scalars] if shared else scalars):
        flat += [(s >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(limbs_per_scalar)]
    arr = (ctypes.c_uint64 * len(flat))(*flat)
    rc = lib().oracle_scalar_mul(c, group, data, int(compressed_in), out, int(compressed_out), n, arr,
                                 limbs_per_scalar, int(shared))
    if rc:
        raise ValueError("oracle_scalar_mul rc=%d" % rc)
    return out.raw


def msm(curve, group, data: bytes, n, scalars, limbs_per_scalar, compressed_in=False, compressed_out=False) -> bytes:
    c = _cid(curve)
    out = ctypes.create_string_buffer(point_size(c, group, compressed_out))
    flat = []
    for s in scalars:
        flat += [(s >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(limbs_per_scalar)]
    arr = (ctypes.c_uint64 * max(1, len(flat)))(*flat)
    rc = lib().oracle_msm(c, group, data, int(compressed_in), n, arr, limbs_per_scalar, out, int(compressed_out))
    if rc:
        raise ValueError("oracle_msm rc=%d" % rc)
    return out.raw
