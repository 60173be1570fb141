import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CURVE_NAMES = ["bls12_377", "bw6", "mnt4_753", "mnt6_753"]

# torch first, libptau_hip.so second — the order bench.py uses.  torch bundles its own HIP runtime; when the library's
# runtime came up first in a process, a later torch.cuda.is_available() was observed to report False on the GPU box
# (tests/test_gpu_parity.py::test_native_library_loaded run after the verify tests).  Querying torch here fixes the order
# for every test selection; on the CPU-only container it is simply False.
try:
    import torch
    torch.cuda.is_available()
except Exception:   # torch is plumbing for the tests, never a requirement of the product
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_package():
    """Import the hyphenated package directory under the name snark_setup_operator_amd."""
    if "snark_setup_operator_amd" in sys.modules:
        return sys.modules["snark_setup_operator_amd"]
    p = os.path.join(ROOT, "snark-setup-operator_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("snark_setup_operator_amd", p,
                                                  submodule_search_locations=[os.path.dirname(p)])
    m = importlib.util.module_from_spec(spec)
    sys.modules["snark_setup_operator_amd"] = m
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def cref():
    from oracle import cref as c
    c.lib()
    return c


def load_golden(curve):
    with open(os.path.join(ROOT, "tests", "golden", "phase1_%s.json" % curve)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return {c: load_golden(c) for c in CURVE_NAMES}


def initial_challenge(pkg, name, prm):
    """Phase1::initialization through the C ABI.  The product holds no G2 generator for MNT4-753 / MNT6-753 (arkworks'
    constants are not recoverable offline): there the oracle's synthetic order-r point goes through the
    caller-supplied-generators form, as the Rust host would pass G2Affine::generator()."""
    if not name.startswith("mnt"):
        return pkg.phase1_cli.initialization(prm)
    from oracle.pyref import serde
    from oracle.pyref.curves import CURVES
    from oracle.pyref.ec import Group
    G1, G2 = Group(CURVES[name], 1), Group(CURVES[name], 2)
    return pkg.phase1_cli.initialization_with_generators(prm, serde.write_point(G1, G1.gen, False), serde.write_point(G2, G2.gen, False))
