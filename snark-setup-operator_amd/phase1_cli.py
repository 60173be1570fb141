"""Host-side mirror of the phase1-cli / phase1 / setup-utils surface the operator uses.

Same names, argument meaning and error behaviour as the reference's L1 calls — where the Rust
functions panic, these raise PtauError:
  * Phase1Parameters::{new_chunk,new_full}          /root/reference/src/utils.rs:326-352
  * phase1_cli::contribute                          /root/reference/src/bin/contribute.rs:811-823
  * batch_exp_mode_from_str / subgroup_check_mode_from_str   contribute.rs:134-145
  * CheckForCorrectness / UseCompression            contribute.rs:38-43
  * calculate_hash                                  /root/reference/src/bin/verify_transcript.rs:344-345
All arithmetic runs in libptau_hip.so (HIP, gfx950) through the C ABI; nothing here computes.
The private key is an INPUT (`rng` in the reference): key generation stays on the Rust host.
"""
import ctypes
import enum
import os
from dataclasses import dataclass

from . import _lib

CURVE_KINDS = {"bls12_377": 0, "bw6": 1, "mnt4_753": 2, "mnt6_753": 3}


class ContributionMode(enum.IntEnum):
    Chunked = 0
    Full = 1


class ProvingSystem(enum.IntEnum):
    Groth16 = 0
    Marlin = 1


class CheckForCorrectness(enum.IntEnum):
    No = 0
    OnlyNonZero = 1
    OnlyInGroup = 2
    Full = 3


class UseCompression(enum.IntEnum):
    No = 0
    Yes = 1


class BatchExpMode(enum.IntEnum):
    Auto = 0
    Direct = 1
    BatchInversion = 2


# include/ptau.h PTAU_BATCH_EXP_INPUT_IN_SUBGROUP: OR into a BatchExpMode when the caller KNOWS every challenge element is
# in the prime-order subgroup (the challenge came out of a verifier that ran the subgroup checks): enables the
# endomorphism-accelerated multiplication without a membership pass.  Not a reference mode: an integration choice.
INPUT_IN_SUBGROUP = 0x100


class SubgroupCheckMode(enum.IntEnum):
    Auto = 0
    Direct = 1
    Batched = 2
    No = 3


def batch_exp_mode_from_str(s: str) -> BatchExpMode:
    try:
        return {"auto": BatchExpMode.Auto, "direct": BatchExpMode.Direct, "batch-inversion": BatchExpMode.BatchInversion}[s]
    except KeyError:
        raise ValueError("unknown batch exp mode %r" % s)


def subgroup_check_mode_from_str(s: str) -> SubgroupCheckMode:
    try:
        return {"auto": SubgroupCheckMode.Auto, "direct": SubgroupCheckMode.Direct, "batched": SubgroupCheckMode.Batched,
                "no": SubgroupCheckMode.No}[s]
    except KeyError:
        raise ValueError("unknown subgroup check mode %r" % s)


def proving_system_from_str(s: str) -> ProvingSystem:
    try:
        return {"groth16": ProvingSystem.Groth16, "marlin": ProvingSystem.Marlin}[s]
    except KeyError:
        raise ValueError("unknown proving system %r" % s)


def curve_from_str(s: str) -> int:
    """curve_kind dispatch (contribute.rs:1156-1211)."""
    try:
        return CURVE_KINDS[s]
    except KeyError:
        raise ValueError("unknown curve kind %r" % s)


def upgrade_correctness_check_config(default: CheckForCorrectness, force: bool) -> CheckForCorrectness:
    """contribute.rs:816-819: --force-correctness-checks upgrades every check to Full."""
    return CheckForCorrectness.Full if force else default


@dataclass(frozen=True)
class Phase1Parameters:
    curve: int
    contribution_mode: ContributionMode
    chunk_index: int
    chunk_size: int
    proving_system: ProvingSystem
    total_size_in_log2: int
    batch_size: int

    @staticmethod
    def new_chunk(curve_kind, contribution_mode, chunk_index, chunk_size, proving_system, power, batch_size):
        c = curve_from_str(curve_kind) if isinstance(curve_kind, str) else int(curve_kind)
        return Phase1Parameters(c, ContributionMode(contribution_mode), chunk_index, chunk_size,
                                ProvingSystem(proving_system), power, batch_size)

    @staticmethod
    def new_full(curve_kind, proving_system, power, batch_size):
        c = curve_from_str(curve_kind) if isinstance(curve_kind, str) else int(curve_kind)
        return Phase1Parameters(c, ContributionMode.Full, 0, 1 << power, ProvingSystem(proving_system), power, batch_size)

    def _c(self):
        return _lib.ptau_params(self.curve, int(self.proving_system), int(self.contribution_mode), self.total_size_in_log2,
                                self.chunk_index, self.chunk_size, self.batch_size)

    def _sizes(self):
        s = _lib.ptau_sizes()
        rc = _lib.load_library().ptau_params_sizes(ctypes.byref(self._c()), ctypes.byref(s))
        if rc:
            raise _lib.PtauError(rc, "invalid parameters")
        return s

    # fields of Phase1Parameters the reference reads (new_setup.rs:274-277, utils.rs:526-532)
    powers_length = property(lambda self: self._sizes().powers_length)
    powers_g1_length = property(lambda self: self._sizes().powers_g1_length)
    g1_chunk_size = property(lambda self: self._sizes().g1_chunk_size)
    other_chunk_size = property(lambda self: self._sizes().other_chunk_size)
    accumulator_size = property(lambda self: self._sizes().accumulator_size)
    contribution_size = property(lambda self: self._sizes().contribution_size)
    public_key_size = property(lambda self: self._sizes().public_key_size)
    hash_size = property(lambda self: self._sizes().hash_size)


@dataclass(frozen=True)
class PrivateKey:
    """phase1::PrivateKey{tau, alpha, beta} as canonical integers + the serialized PublicKey
    (uncompressed, 6 G1 + 3 G2) produced by the host's Phase1::key_generation."""
    tau: int
    alpha: int
    beta: int
    public_key: bytes


def calculate_hash(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(64)
    _lib.load_library().ptau_blake2b512(data, len(data), out)
    return out.raw


def contribute(challenge_filename, challenge_hash_filename, response_filename, response_hash_filename,
               check_input_correctness, batch_exp_mode, parameters: Phase1Parameters, key: PrivateKey, device: int = 0):
    """phase1_cli::contribute (contribute.rs:811-823); `key` replaces `rng`."""
    L = _lib.load_library()
    nl = L.ptau_fr_limbs(parameters.curve)
    eb = _lib.errbuf()
    rc = L.ptau_contribute_file(ctypes.byref(parameters._c()), os.fsencode(challenge_filename),
                                os.fsencode(challenge_hash_filename), os.fsencode(response_filename),
                                os.fsencode(response_hash_filename), _lib.limbs(key.tau, nl), _lib.limbs(key.alpha, nl),
                                _lib.limbs(key.beta, nl), key.public_key, len(key.public_key),
                                int(check_input_correctness), int(batch_exp_mode), device, eb, len(eb))
    _lib.check(rc, eb)


def computation(challenge: bytes, parameters: Phase1Parameters, key: PrivateKey, check_input=CheckForCorrectness.No,
                batch_exp_mode=BatchExpMode.Auto, device: int = 0) -> bytes:
    """Phase1::computation on in-memory buffers: returns a contribution_size buffer whose element
    region is filled (hash header and public-key tail left zero)."""
    L = _lib.load_library()
    nl = L.ptau_fr_limbs(parameters.curve)
    out = ctypes.create_string_buffer(parameters.contribution_size)
    eb = _lib.errbuf()
    rc = L.ptau_phase1_computation(ctypes.byref(parameters._c()), challenge, len(challenge), out, len(out),
                                   _lib.limbs(key.tau, nl), _lib.limbs(key.alpha, nl), _lib.limbs(key.beta, nl),
                                   int(check_input), int(batch_exp_mode), device, eb, len(eb))
    _lib.check(rc, eb)
    return out.raw


def apply_powers(curve, group, data: bytes, n, start, tau, coeff=None, compressed_in=False, compressed_out=True,
                 check_input=CheckForCorrectness.No, batch_exp_mode=BatchExpMode.Auto, device=0) -> bytes:
    """setup-utils apply_powers / batch_exp on one vector (host buffers)."""
    L = _lib.load_library()
    c = curve_from_str(curve) if isinstance(curve, str) else int(curve)
    nl = L.ptau_fr_limbs(c)
    out = ctypes.create_string_buffer(max(1, n * L.ptau_point_size(c, group, int(compressed_out))))
    eb = _lib.errbuf()
    rc = L.ptau_apply_powers(c, group, data, int(compressed_in), out, int(compressed_out), n, start, _lib.limbs(tau, nl),
                             None if coeff is None else _lib.limbs(coeff, nl), int(check_input), int(batch_exp_mode), device,
                             eb, len(eb))
    _lib.check(rc, eb)
    return out.raw[:n * L.ptau_point_size(c, group, int(compressed_out))]


def decompress(curve, group, data: bytes, n, check_output=CheckForCorrectness.No, device=0) -> bytes:
    """Phase1::decompress on one vector (host buffers)."""
    L = _lib.load_library()
    c = curve_from_str(curve) if isinstance(curve, str) else int(curve)
    out = ctypes.create_string_buffer(max(1, n * L.ptau_point_size(c, group, 0)))
    eb = _lib.errbuf()
    rc = L.ptau_decompress(c, group, data, out, n, int(check_output), device, eb, len(eb))
    _lib.check(rc, eb)
    return out.raw[:n * L.ptau_point_size(c, group, 0)]


def subgroup_check(curve, group, data: bytes, n, compressed_in=False, device=0) -> None:
    """setup-utils check_subgroup (Direct) + on-curve; raises PtauError(-5 / -3) like the reference panics."""
    L = _lib.load_library()
    c = curve_from_str(curve) if isinstance(curve, str) else int(curve)
    eb = _lib.errbuf()
    _lib.check(L.ptau_subgroup_check(c, group, data, int(compressed_in), n, device, eb, len(eb)), eb)


def verification_and_decompress(challenge: bytes, response: bytes, parameters: Phase1Parameters,
                                check_input=CheckForCorrectness.No, check_output=CheckForCorrectness.OnlyNonZero,
                                subgroup_check_mode=SubgroupCheckMode.Auto, ratio_check=False, verify_pok=False,
                                g2_s: bytes = None, tau_g2_check: bytes = None, device=0) -> bytes:
    """Phase1::verification + Phase1::decompress on in-memory buffers; returns new_challenge.  verify_pok=False runs the
    element checks only (the parity tests' synthetic public keys carry no proofs of knowledge)."""
    L = _lib.load_library()
    out = ctypes.create_string_buffer(parameters.accumulator_size)
    eb = _lib.errbuf()
    rc = L.ptau_verify_chunk(ctypes.byref(parameters._c()), challenge, len(challenge), response, len(response), out, len(out),
                             int(check_input), int(check_output), int(subgroup_check_mode), int(bool(ratio_check)),
                             int(bool(verify_pok)), g2_s, tau_g2_check, device, eb, len(eb))
    _lib.check(rc, eb)
    return out.raw


def transform_pok_and_correctness(challenge_filename, challenge_hash_filename, check_input_correctness, response_filename,
                                  response_hash_filename, check_output_correctness, new_challenge_filename,
                                  new_challenge_hash_filename, subgroup_check_mode, ratio_check,
                                  parameters: Phase1Parameters, g2_s: bytes = None, tau_g2_check: bytes = None, device: int = 0):
    """phase1_cli::transform_pok_and_correctness (contribute.rs:968-986), the reference's eleven arguments in its order.
    Proofs of knowledge are always checked; g2_s=None lets the library derive the compute_g2_s points itself.
    tau_g2_check: (tauG2[0] | tauG2[1]) uncompressed, of the same contribution — lets ratio_check cover a tauG1-only chunk;
    without it last_timing()["ratio_checks"] == 0 reports that such a chunk's power ratios were left to transform_ratios."""
    L = _lib.load_library()
    eb = _lib.errbuf()
    rc = L.ptau_transform_pok_and_correctness_file(
        ctypes.byref(parameters._c()), os.fsencode(challenge_filename), os.fsencode(challenge_hash_filename),
        int(check_input_correctness), os.fsencode(response_filename), os.fsencode(response_hash_filename),
        int(check_output_correctness), os.fsencode(new_challenge_filename), os.fsencode(new_challenge_hash_filename),
        int(subgroup_check_mode), int(bool(ratio_check)), g2_s, tau_g2_check, device, eb, len(eb))
    _lib.check(rc, eb)


def derive_rng_words(seed: bytes, n_words: int, word_offset: int = 0):
    """setup_utils::derive_rng_from_seed (contribute.rs:783-789): the ChaCha20Rng word stream of the first 32 seed bytes."""
    L = _lib.load_library()
    out = (ctypes.c_uint32 * n_words)()
    rc = L.ptau_chacha20_rng_words(seed[:32], word_offset, out, n_words)
    if rc:
        raise _lib.PtauError(rc, "ptau_chacha20_rng_words")
    return list(out)


def key_generation(curve, seed: bytes, digest: bytes) -> PrivateKey:
    """Phase1::key_generation(derive_rng_from_seed(seed), digest) -> PrivateKey (with the serialized PublicKey)."""
    L = _lib.load_library()
    c = curve_from_str(curve) if isinstance(curve, str) else int(curve)
    nl = L.ptau_fr_limbs(c)
    t, a, b = [(ctypes.c_uint64 * nl)() for _ in range(3)]
    pk = ctypes.create_string_buffer(6 * L.ptau_point_size(c, 1, 0) + 3 * L.ptau_point_size(c, 2, 0))
    eb = _lib.errbuf()
    _lib.check(L.ptau_key_generation(c, seed, len(seed), digest, t, a, b, pk, len(pk), eb, len(eb)), eb)
    val = lambda arr: sum(int(v) << (64 * k) for k, v in enumerate(arr))
    return PrivateKey(val(t), val(a), val(b), pk.raw)


def compute_g2_s(curve, digest: bytes, public_key: bytes) -> bytes:
    """The three setup_utils::compute_g2_s points of a serialized PublicKey (uncompressed G2 x 3)."""
    L = _lib.load_library()
    c = curve_from_str(curve) if isinstance(curve, str) else int(curve)
    out = ctypes.create_string_buffer(3 * L.ptau_point_size(c, 2, 0))
    eb = _lib.errbuf()
    _lib.check(L.ptau_compute_g2_s(c, digest, public_key, len(public_key), out, eb, len(eb)), eb)
    return out.raw


def contribute_seeded(challenge_filename, challenge_hash_filename, response_filename, response_hash_filename,
                      check_input_correctness, batch_exp_mode, parameters: Phase1Parameters, seed: bytes, device: int = 0):
    """phase1_cli::contribute with rng = derive_rng_from_seed(seed) (contribute.rs:783-789, 811-823)."""
    L = _lib.load_library()
    eb = _lib.errbuf()
    rc = L.ptau_contribute_file_seeded(ctypes.byref(parameters._c()), os.fsencode(challenge_filename),
                                       os.fsencode(challenge_hash_filename), os.fsencode(response_filename),
                                       os.fsencode(response_hash_filename), seed, len(seed), int(check_input_correctness),
                                       int(batch_exp_mode), device, eb, len(eb))
    _lib.check(rc, eb)


def dense_multiexp(curve, group, data: bytes, scalars, compressed_in=False, compressed_out=False, device=0) -> bytes:
    """setup_utils::dense_multiexp: sum_i scalars[i] * P_i -> one serialized point."""
    L = _lib.load_library()
    c = curve_from_str(curve) if isinstance(curve, str) else int(curve)
    nl = L.ptau_fr_limbs(c)
    n = len(scalars)
    flat = []
    for s in scalars:
        flat += [(s >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(nl)]
    arr = (ctypes.c_uint64 * max(1, len(flat)))(*flat)
    out = ctypes.create_string_buffer(L.ptau_point_size(c, group, int(compressed_out)))
    eb = _lib.errbuf()
    _lib.check(L.ptau_msm(c, group, data, int(compressed_in), n, arr, out, int(compressed_out), device, eb, len(eb)), eb)
    return out.raw


def _chacha12_block(key_words, counter):
    M = 0xFFFFFFFF
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [counter & M, (counter >> 32) & M, 0, 0]
    x = list(s)
    rot = lambda v, n: ((v << n) | (v >> (32 - n))) & M

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M; x[d] = rot(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M; x[b] = rot(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M; x[d] = rot(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M; x[b] = rot(x[b] ^ x[c], 7)
    for _ in range(6):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & M for a, b in zip(x, s)]


def power_pairs_coefficient(seed: bytes, idx: int, fr_bits: int = 0) -> int:
    """The coefficient rho_idx that ptau_power_pairs uses (documented PRF, kernels_impl.hpp::prf_rho): ChaCha12 in counter
    mode — key = the 16-byte seed followed by the 16-byte tag "ptau-rho perindx", block counter = the element's global index —
    words 0..3 of the block, top bit cleared (PRF_BITS = 127)."""
    key = [int.from_bytes(seed[4 * i:4 * i + 4], "little") for i in range(4)] + [0x75617470, 0x6F68722D, 0x72657020, 0x78646E69]
    w = _chacha12_block(key, idx)
    v = w[0] | (w[1] << 32) | (w[2] << 64) | (w[3] << 96)
    return v & ((1 << 127) - 1)


def power_pairs(curve, group, data: bytes, n, seed: bytes = None, index_base=0, compressed_in=False, device=0):
    """setup_utils::power_pairs: (sum rho_i P_i, sum rho_i P_{i+1}) as two uncompressed points."""
    L = _lib.load_library()
    c = curve_from_str(curve) if isinstance(curve, str) else int(curve)
    sz = L.ptau_point_size(c, group, 0)
    a, b = ctypes.create_string_buffer(sz), ctypes.create_string_buffer(sz)
    eb = _lib.errbuf()
    _lib.check(L.ptau_power_pairs(c, group, data, int(compressed_in), n, seed, index_base, a, b, device, eb, len(eb)), eb)
    return a.raw, b.raw


def check_same_ratio(curve, g1_pair, g2_pair) -> bool:
    """setup_utils::check_same_ratio((a, b), (c, d)): e(a, d) == e(b, c).  Points are uncompressed bytes."""
    L = _lib.load_library()
    c = curve_from_str(curve) if isinstance(curve, str) else int(curve)
    eb = _lib.errbuf()
    rc = L.ptau_check_same_ratio(c, g1_pair[0], g1_pair[1], g2_pair[0], g2_pair[1], eb, len(eb))
    if rc == -6:
        return False
    _lib.check(rc, eb)
    return True


def check_same_ratio_batch(curve, checks, want_gt=False, device=0):
    """n same-ratio checks in one launch of the GPU pairing kernel.  checks: list of ((g1_a, g1_b), (g2_c, g2_d)) uncompressed
    byte strings.  Returns the list of results (1 / 0 / -1) and, with want_gt, per check the GT value as k lists of ints."""
    L = _lib.load_library()
    c = curve_from_str(curve) if isinstance(curve, str) else int(curve)
    n = len(checks)
    blob = b"".join(g1[0] + g1[1] + g2[0] + g2[1] for g1, g2 in checks)
    res = (ctypes.c_int * max(1, n))()
    k = {0: 12, 1: 6, 2: 4, 3: 6}[c]
    n32 = (L.ptau_point_size(c, 1, 1) * 8 + 31) // 32
    gt = (ctypes.c_uint32 * (n * k * n32))() if want_gt else None
    eb = _lib.errbuf()
    _lib.check(L.ptau_check_same_ratio_batch(c, blob, n, res, gt, device, eb, len(eb)), eb)
    out = list(res)[:n]
    if not want_gt:
        return out
    vals = []
    for t in range(n):
        vals.append([sum(int(gt[(t * k + j) * n32 + w]) << (32 * w) for w in range(n32)) for j in range(k)])
    return out, vals


def new_challenge(challenge_filename, challenge_hash_filename, parameters: Phase1Parameters):
    """phase1_cli::new_challenge (new_setup.rs:105-109)."""
    L = _lib.load_library()
    eb = _lib.errbuf()
    _lib.check(L.ptau_new_challenge_file(ctypes.byref(parameters._c()), os.fsencode(challenge_filename),
                                         os.fsencode(challenge_hash_filename), eb, len(eb)), eb)


def generator(curve, group) -> bytes:
    """Built-in group generator, uncompressed (PtauError -10 for the G2 of MNT4-753 / MNT6-753)."""
    L = _lib.load_library()
    c = curve_from_str(curve) if isinstance(curve, str) else int(curve)
    out = ctypes.create_string_buffer(L.ptau_point_size(c, group, 0))
    eb = _lib.errbuf()
    _lib.check(L.ptau_generator(c, group, out, len(out), eb, len(eb)), eb)
    return out.raw


def initialization_with_generators(parameters: Phase1Parameters, g1_generator: bytes, g2_generator: bytes) -> bytes:
    """Phase1::initialization with caller-supplied generators (uncompressed): the MNT curves' path."""
    L = _lib.load_library()
    out = ctypes.create_string_buffer(parameters.accumulator_size)
    eb = _lib.errbuf()
    _lib.check(L.ptau_new_challenge_with_generators(ctypes.byref(parameters._c()), g1_generator, g2_generator, out, len(out),
                                                    eb, len(eb)), eb)
    return out.raw


def initialization(parameters: Phase1Parameters) -> bytes:
    """Phase1::initialization on a buffer."""
    L = _lib.load_library()
    out = ctypes.create_string_buffer(parameters.accumulator_size)
    eb = _lib.errbuf()
    _lib.check(L.ptau_new_challenge(ctypes.byref(parameters._c()), out, len(out), eb, len(eb)), eb)
    return out.raw


def combine(responses, chunk_size, full_parameters: Phase1Parameters, device=0) -> bytes:
    """phase1_cli::combine / Phase1::aggregation on in-memory responses -> Full-mode uncompressed accumulator."""
    L = _lib.load_library()
    n = len(responses)
    arr = (ctypes.c_char_p * n)(*responses)
    lens = (ctypes.c_size_t * n)(*[len(r) for r in responses])
    out = ctypes.create_string_buffer(full_parameters.accumulator_size)
    eb = _lib.errbuf()
    _lib.check(L.ptau_combine(ctypes.byref(full_parameters._c()), chunk_size, arr, lens, n, out, len(out), device, eb, len(eb)), eb)
    return out.raw


def transform_ratios(combined: bytes, check_input_correctness, full_parameters: Phase1Parameters, device=0):
    """phase1_cli::transform_ratios / Phase1::aggregate_verification; raises PtauError(-6) on a bad ratio."""
    L = _lib.load_library()
    eb = _lib.errbuf()
    _lib.check(L.ptau_transform_ratios(ctypes.byref(full_parameters._c()), combined, len(combined),
                                       int(check_input_correctness), device, eb, len(eb)), eb)


def combine_file(response_list_filename, combined_filename, parameters: Phase1Parameters, device=0):
    """phase1_cli::combine(response_list_filename, combined_filename, &parameters) — control.rs:564-568,
    verify_transcript.rs:603-607: `parameters` = create_parameters_for_chunk(.., 0); the list file holds one chunk-response path
    per line; `combined_filename` is created (create_new)."""
    L = _lib.load_library()
    eb = _lib.errbuf()
    _lib.check(L.ptau_combine_file(ctypes.byref(parameters._c()), os.fsencode(response_list_filename),
                                   os.fsencode(combined_filename), device, eb, len(eb)), eb)


def transform_ratios_file(combined_filename, check_input_correctness, full_parameters: Phase1Parameters, device=0, devices=None):
    """phase1_cli::transform_ratios(combined_filename, check_input_correctness, &full_parameters) — control.rs:587-591,
    verify_transcript.rs:646-653, 811-822.  devices=[ordinals]: the reductions are spread over several GPUs from this one
    process (ptau_transform_ratios_file_devices)."""
    L = _lib.load_library()
    eb = _lib.errbuf()
    if devices is None:
        rc = L.ptau_transform_ratios_file(ctypes.byref(full_parameters._c()), os.fsencode(combined_filename),
                                          int(check_input_correctness), device, eb, len(eb))
    else:
        arr = (ctypes.c_int * len(devices))(*devices)
        rc = L.ptau_transform_ratios_file_devices(ctypes.byref(full_parameters._c()), os.fsencode(combined_filename),
                                                  int(check_input_correctness), arr, len(devices), eb, len(eb))
    _lib.check(rc, eb)


def transform_ratios_devices(combined: bytes, check_input_correctness, full_parameters: Phase1Parameters, devices):
    """transform_ratios on a buffer with the vector reductions spread over `devices` from this one process."""
    L = _lib.load_library()
    eb = _lib.errbuf()
    arr = (ctypes.c_int * len(devices))(*devices)
    _lib.check(L.ptau_transform_ratios_devices(ctypes.byref(full_parameters._c()), combined, len(combined),
                                               int(check_input_correctness), arr, len(devices), eb, len(eb)), eb)


def transform_ratios_sharded(combined, full_parameters: Phase1Parameters, dist=None, device=0, seed: bytes = None) -> None:
    """phase1_cli::transform_ratios with the vector reductions split over the ranks of an initialised
    torch.distributed group (one process per GPU; farm.sharded_transform_ratios): each rank runs ptau_power_pairs on
    its slice with the shared coefficient stream, ONE all_gather of the partial pairs per vector (RCCL over xGMI),
    the partial sums go through ptau_msm with unit scalars, the pairing tail runs on the host.
    Raises PtauError(-6) on every rank when a ratio check fails."""
    from . import farm
    curve = full_parameters.curve
    L = _lib.load_library()
    sizes = {1: L.ptau_point_size(curve, 1, 0), 2: L.ptau_point_size(curve, 2, 0)}

    def pp(grp, data, n, sd, index_base):
        return power_pairs(curve, grp, data, n, seed=sd, index_base=index_base, device=device)

    def sum_points(grp, pts):
        return dense_multiexp(curve, grp, b"".join(pts), [1] * len(pts), device=device)

    def check(g1p, g2p):     # the pairing kernel (K5 on the GPU)
        return check_same_ratio_batch(curve, [(g1p, g2p)], device=device)[0] == 1
    ok = farm.sharded_transform_ratios(combined, sizes, full_parameters.powers_g1_length, full_parameters.powers_length,
                                       pp, sum_points, check, dist=dist, seed=seed, device=device)
    if not ok:
        raise _lib.PtauError(-6, "power ratio check failed (sharded transform_ratios)")


def verify_pok(challenge: bytes, response: bytes, g2_s: bytes, parameters: Phase1Parameters):
    """Proof-of-knowledge + chunk-0 before/after checks of Phase1::verification; g2_s from the host's compute_g2_s."""
    L = _lib.load_library()
    eb = _lib.errbuf()
    _lib.check(L.ptau_verify_pok(ctypes.byref(parameters._c()), challenge, len(challenge), response, len(response), g2_s,
                                 eb, len(eb)), eb)


def last_timing(device=0):
    t = _lib.ptau_timing()
    _lib.load_library().ptau_last_timing(device, ctypes.byref(t))
    return {k: getattr(t, k) for k, _ in _lib.ptau_timing._fields_}
