"""ctypes binding of libptau_hip.so (include/ptau.h).  Plumbing only — no arithmetic here.

Fails loudly when the HIP extension is missing: there is no CPU fallback in the product path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptau_hip.so")
_lib = None

PTAU_OK = 0
ERR_NAMES = {-1: "PTAU_E_ARG", -2: "PTAU_E_LEN", -3: "PTAU_E_POINT", -4: "PTAU_E_POINT_AT_INFINITY",
             -5: "PTAU_E_NOT_IN_SUBGROUP", -6: "PTAU_E_INVALID_RATIO", -7: "PTAU_E_HASH_CHAIN", -8: "PTAU_E_DEVICE",
             -9: "PTAU_E_IO", -10: "PTAU_E_UNSUPPORTED"}

# every symbol include/ptau.h declares (tests check the library exports each one)
DECLARED_SYMBOLS = [
    "ptau_strerror", "ptau_device_count", "ptau_device_info", "ptau_params_sizes", "ptau_point_size",
    "ptau_fr_limbs", "ptau_blake2b512", "ptau_apply_powers", "ptau_apply_powers_dev", "ptau_phase1_computation",
    "ptau_contribute_file", "ptau_decompress", "ptau_last_timing", "ptau_subgroup_check", "ptau_verify_chunk",
    "ptau_transform_pok_and_correctness_file", "ptau_msm", "ptau_power_pairs", "ptau_check_same_ratio", "ptau_new_challenge", "ptau_new_challenge_file",
    "ptau_combine", "ptau_transform_ratios", "ptau_verify_pok", "ptau_new_challenge_with_generators", "ptau_generator",
    "ptau_chacha20_rng_words", "ptau_key_generation", "ptau_private_key_from_seed", "ptau_compute_g2_s", "ptau_contribute_file_seeded",
    "ptau_check_same_ratio_batch", "ptau_blake2b_implementation", "ptau_device_busy_ms",
    "ptau_combine_file", "ptau_transform_ratios_file", "ptau_transform_ratios_devices", "ptau_transform_ratios_file_devices",
]


class PtauError(RuntimeError):
    """Raised where the reference's L1 functions would panic (contribute.rs:842-856)."""

    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "PTAU_E_?"), code, message))
        self.code = code


class ptau_params(ctypes.Structure):
    _fields_ = [("curve", ctypes.c_int32), ("proving_system", ctypes.c_int32), ("contribution_mode", ctypes.c_int32),
                ("power", ctypes.c_uint32), ("chunk_index", ctypes.c_uint64), ("chunk_size", ctypes.c_uint64),
                ("batch_size", ctypes.c_uint64)]


class ptau_sizes(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("powers_length", "powers_g1_length", "g1_chunk_size", "other_chunk_size",
                                               "accumulator_size", "contribution_size", "public_key_size", "hash_size",
                                               "start")]


class ptau_timing(ctypes.Structure):
    _fields_ = [("h2d_ms", ctypes.c_double), ("kernel_ms", ctypes.c_double), ("d2h_ms", ctypes.c_double),
                ("scalar_mul_ms", ctypes.c_double), ("normalize_ms", ctypes.c_double), ("points", ctypes.c_uint64),
                ("decompress_ms", ctypes.c_double), ("subgroup_ms", ctypes.c_double), ("msm_ms", ctypes.c_double),
                ("ratio_checks", ctypes.c_uint32), ("pok_checks", ctypes.c_uint32),
                ("msm_tail_ms", ctypes.c_double)]


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libptau_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                          "there is no CPU fallback")
    L = ctypes.CDLL(LIB_PATH)
    vp, u8p, u64p, cp = ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p
    i, u64, sz = ctypes.c_int, ctypes.c_uint64, ctypes.c_size_t
    L.ptau_strerror.restype = ctypes.c_char_p
    L.ptau_strerror.argtypes = [i]
    L.ptau_device_info.argtypes = [i, cp, sz, ctypes.POINTER(ctypes.c_int), u64p]
    L.ptau_params_sizes.argtypes = [ctypes.POINTER(ptau_params), ctypes.POINTER(ptau_sizes)]
    L.ptau_point_size.argtypes = [i, i, i]
    L.ptau_fr_limbs.argtypes = [i]
    L.ptau_blake2b512.argtypes = [u8p, sz, u8p]
    L.ptau_blake2b512.restype = None
    L.ptau_apply_powers.argtypes = [i, i, u8p, i, u8p, i, u64, u64, u64p, u64p, i, i, i, cp, sz]
    L.ptau_apply_powers_dev.argtypes = [i, i, vp, i, vp, i, u64, u64, u64p, u64p, i, i, i, cp, sz]
    L.ptau_phase1_computation.argtypes = [ctypes.POINTER(ptau_params), u8p, sz, u8p, sz, u64p, u64p, u64p, i, i, i, cp, sz]
    L.ptau_contribute_file.argtypes = [ctypes.POINTER(ptau_params), cp, cp, cp, cp, u64p, u64p, u64p, u8p, sz, i, i, i, cp, sz]
    L.ptau_decompress.argtypes = [i, i, u8p, u8p, u64, i, i, cp, sz]
    L.ptau_last_timing.argtypes = [i, ctypes.POINTER(ptau_timing)]
    L.ptau_msm.argtypes = [i, i, u8p, i, u64, u64p, u8p, i, i, cp, sz]
    L.ptau_power_pairs.argtypes = [i, i, u8p, i, u64, u8p, u64, u8p, u8p, i, cp, sz]
    L.ptau_check_same_ratio.argtypes = [i, u8p, u8p, u8p, u8p, cp, sz]
    L.ptau_new_challenge.argtypes = [ctypes.POINTER(ptau_params), u8p, sz, cp, sz]
    L.ptau_new_challenge_file.argtypes = [ctypes.POINTER(ptau_params), cp, cp, cp, sz]
    L.ptau_combine.argtypes = [ctypes.POINTER(ptau_params), u64, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz), u64, u8p, sz, i, cp, sz]
    L.ptau_transform_ratios.argtypes = [ctypes.POINTER(ptau_params), u8p, sz, i, i, cp, sz]
    L.ptau_verify_pok.argtypes = [ctypes.POINTER(ptau_params), u8p, sz, u8p, sz, u8p, cp, sz]
    L.ptau_subgroup_check.argtypes = [i, i, u8p, i, u64, i, cp, sz]
    L.ptau_verify_chunk.argtypes = [ctypes.POINTER(ptau_params), u8p, sz, u8p, sz, u8p, sz, i, i, i, i, i, u8p, u8p, i, cp, sz]
    L.ptau_transform_pok_and_correctness_file.argtypes = [ctypes.POINTER(ptau_params), cp, cp, i, cp, cp, i, cp, cp, i, i, u8p, u8p, i, cp, sz]
    L.ptau_combine_file.argtypes = [ctypes.POINTER(ptau_params), cp, cp, i, cp, sz]
    L.ptau_transform_ratios_file.argtypes = [ctypes.POINTER(ptau_params), cp, i, i, cp, sz]
    L.ptau_transform_ratios_devices.argtypes = [ctypes.POINTER(ptau_params), u8p, sz, i, ctypes.POINTER(ctypes.c_int), i, cp, sz]
    L.ptau_transform_ratios_file_devices.argtypes = [ctypes.POINTER(ptau_params), cp, i, ctypes.POINTER(ctypes.c_int), i, cp, sz]
    L.ptau_new_challenge_with_generators.argtypes = [ctypes.POINTER(ptau_params), u8p, u8p, u8p, sz, cp, sz]
    L.ptau_generator.argtypes = [i, i, u8p, sz, cp, sz]
    L.ptau_chacha20_rng_words.argtypes = [u8p, u64, ctypes.POINTER(ctypes.c_uint32), sz]
    L.ptau_key_generation.argtypes = [i, u8p, sz, u8p, u64p, u64p, u64p, u8p, sz, cp, sz]
    L.ptau_private_key_from_seed.argtypes = [i, u8p, sz, u64p, u64p, u64p, cp, sz]
    L.ptau_compute_g2_s.argtypes = [i, u8p, u8p, sz, u8p, cp, sz]
    L.ptau_contribute_file_seeded.argtypes = [ctypes.POINTER(ptau_params), cp, cp, cp, cp, u8p, sz, i, i, i, cp, sz]
    L.ptau_blake2b_implementation.argtypes = [i]
    L.ptau_device_busy_ms.argtypes = [i]
    L.ptau_device_busy_ms.restype = ctypes.c_double
    L.ptau_check_same_ratio_batch.argtypes = [i, u8p, u64, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint32), i, cp, sz]
    _lib = L
    return L


def limbs(x: int, n: int):
    return (ctypes.c_uint64 * n)(*[(x >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(n)])


def check(rc, errbuf):
    if rc != PTAU_OK:
        raise PtauError(rc, errbuf.value.decode(errors="replace"))


def errbuf():
    return ctypes.create_string_buffer(256)
