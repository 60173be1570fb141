"""Per-curve / per-group tau-update kernel timings through the C ABI (device-resident buffers).
Usage (GPU box):  python tools/bench_all_groups.py [log2_n]"""
import ctypes, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("snark_setup_operator_amd", os.path.join(ROOT, "snark-setup-operator_amd", "__init__.py"),
                                              submodule_search_locations=[os.path.join(ROOT, "snark-setup-operator_amd")])
pkg = importlib.util.module_from_spec(spec); sys.modules["snark_setup_operator_amd"] = pkg; spec.loader.exec_module(pkg)
import torch
from oracle.pyref.curves import CURVES
from oracle.pyref import serde
from oracle.pyref.ec import Group

L = pkg.load_library()
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
eb = ctypes.create_string_buffer(256)
for name, c in CURVES.items():
    for grp in (1, 2):
        n = 1 << log2n
        G = Group(c, grp)
        cid = c.curve_id
        nl = L.ptau_fr_limbs(cid)
        gen = serde.write_point(G, G.gen, False)
        d_gen = torch.frombuffer(bytearray(gen * n), dtype=torch.uint8).cuda()
        d_a = torch.empty(n * G.uncomp_size, dtype=torch.uint8, device="cuda")
        d_b = torch.empty(n * G.comp_size, dtype=torch.uint8, device="cuda")
        s, tau = c.r // 3, c.r // 7 * 5
        def run(i, ci, o, co, sc):
            rc = L.ptau_apply_powers_dev(cid, grp, ctypes.c_void_p(i.data_ptr()), ci, ctypes.c_void_p(o.data_ptr()), co, n, 0,
                                         pkg._lib.limbs(sc, nl), None, 0, 0, eb, 256)
            assert rc == 0, eb.value
        run(d_gen, 0, d_a, 0, s)
        run(d_a, 0, d_b, 1, tau)
        run(d_a, 0, d_b, 1, tau)
        t = pkg.phase1_cli.last_timing(0)
        print("%-10s G%d n=2^%d  scalar_mul %9.2f ms  normalize %7.2f ms  -> %10.0f updates/s" %
              (name, grp, log2n, t["scalar_mul_ms"], t["normalize_ms"], n / (t["kernel_ms"] / 1e3)), flush=True)
