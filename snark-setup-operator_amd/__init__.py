"""MI355X-native chunked powers-of-tau engine — host-side mirror of the reference's L1 interface.

The directory name carries a hyphen (it mirrors the reference repository's name), so import it with
`importlib` (see tests/conftest.py::load_package) under the module name `snark_setup_operator_amd`.

Only what the hot path needs lives here:
  csrc/            HIP kernels + the C ABI (libptau_hip.so, declared in include/ptau.h)
  _lib.py          ctypes binding of the C ABI (plumbing)
  phase1_cli.py    same names / argument meaning / error behaviour as the phase1_cli functions the
                   operator calls (/root/reference/src/bin/contribute.rs:811-823, 968-986)
"""
from . import _lib, phase1_cli  # noqa: F401
from ._lib import PtauError, load_library  # noqa: F401
from .phase1_cli import (  # noqa: F401
    Phase1Parameters, PrivateKey, ContributionMode, ProvingSystem, CheckForCorrectness, BatchExpMode,
    SubgroupCheckMode, UseCompression, contribute, batch_exp_mode_from_str, subgroup_check_mode_from_str,
    proving_system_from_str, curve_from_str, calculate_hash, transform_pok_and_correctness, new_challenge, combine, transform_ratios,
)
