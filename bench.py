#!/usr/bin/env python3
"""bench.py — G1 point-updates/s of the chunk tau-update (BASELINE.json metric), one rank per GPU.

Step = one pass of the hot path over one chunk vector: by default (--config 2 = BASELINE configs[1], the configuration the
metric is quoted on) the tau-update of the 2^20-point tauG1 vector of a BLS12-377 power-20 chunk, device-resident in and out
(uncompressed challenge points -> compressed response points).  --config 3 / 4 run the tauG1 vector of one BW6-761 (2^18) /
MNT4-753 (2^16) chunk and list the other groups of that config; --config 5 times the verification of a 2^20 BLS12-377 chunk.
N > 1: every rank updates its own chunk (independent chunks, no data-path collective) -> weak scaling.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BLS_R = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001
G1_GEN_X = 0x008848defe740a67c8fc6225bf87ff5485951e2caa9d41bb188282c8bd37cb5cd5481512ffcd394eeab9b16eb21be9ef
G1_GEN_Y = 0x01914a69c5102eff1f674f5d30afeec4bd7fb348ca3e52d96d182ad44fb82305c2fe3d3634a9591afd82de55559c8ea6
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def load_pkg():
    p = os.path.join(ROOT, "snark-setup-operator_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("snark_setup_operator_amd", p, submodule_search_locations=[os.path.dirname(p)])
    m = importlib.util.module_from_spec(spec)
    sys.modules["snark_setup_operator_amd"] = m
    spec.loader.exec_module(m)
    return m


def det_scalar(tag):
    import hashlib
    return int.from_bytes(hashlib.blake2b(tag.encode(), digest_size=64).digest(), "little") % BLS_R


def usable_cores():
    """Threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, int(os.environ.get("PTAU_BENCH_CPU_THREADS", "64")))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown CPU"


def cpu_baseline(curve, group, sample_points, chal_bytes, start, tau, single_core=True):
    """The C oracle ("port": ark-ec's MSB-first double-and-add with mixed additions + batch normalisation, the algorithm the
    reference's batch_exp runs; 64-bit CIOS Montgomery, no assembly) timed on this host over a bounded sample of the same
    vector: all usable cores, and one core on a smaller slice."""
    from oracle import cref
    cores = usable_cores()
    cref.lib()
    usz0 = len(chal_bytes) // sample_points
    cref.apply_powers(curve, group, chal_bytes[:usz0], 1, start, tau, None, False, True, nthreads=1)   # one-time table set-up of the oracle: untimed
    t0 = time.time()
    out = cref.apply_powers(curve, group, chal_bytes, sample_points, start, tau, None, False, True, nthreads=cores)
    dt = time.time() - t0
    res = {"value": sample_points / dt, "unit": "G1 point-updates/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
           "sample": "%d points of the same vector, C oracle (oracle/c), %d threads, %.1f s wall" % (sample_points, cores, dt)}
    if single_core:
        usz = len(chal_bytes) // sample_points
        m = max(1, min(sample_points, sample_points // max(1, cores) // 2))
        t0 = time.time()
        one = cref.apply_powers(curve, group, chal_bytes[:m * usz], m, start, tau, None, False, True, nthreads=1)
        d1 = time.time() - t0
        if one != out[:len(one)]:
            raise SystemExit("CPU oracle disagrees with itself across thread counts")
        res["single_core"] = {"value": m / d1, "sample": "%d points, 1 thread, %.1f s wall" % (m, d1)}
    return out, res


def cpu_chunk_baseline(curve, log2_indices=14):
    """CPU denominator of the stated target ("host-CPU chunk-contribution throughput"): Phase1::computation of the C oracle on
    chunk 0 of a power-`log2_indices` ceremony — ALL FIVE vectors: per index one tauG1 + tauG2 + alphaG1 + betaG1 update (a
    chunk index costs the same whatever the ceremony's power: the exponents are full-width) — on every usable core, and on ONE
    core over a 16x smaller chunk.  Reported per chunk INDEX, and scaled to the whole chunk the GPU lines time."""
    from oracle import cref
    from oracle.pyref import phase1 as P
    from oracle.pyref.curves import CURVES
    cores = usable_cores()
    cref.lib()
    tau, alpha, beta = det_scalar("bench/tau") % (1 << 250), det_scalar("bench/alpha") % (1 << 250), det_scalar("bench/beta") % (1 << 250)
    out = {"kind": "port", "cpu_model": cpu_model(), "cores": cores, "unit": "chunk indices/s (1 tauG1 + 1 tauG2 + 1 alphaG1 + 1 betaG1 update each)"}
    warm = P.initialization(P.Params(CURVES[curve], P.CHUNKED, 0, 2, 1, 2))
    cref.computation(curve, 0, 0, 2, 1, warm, tau, alpha, beta, nthreads=1)   # one-time table set-up of the oracle: untimed
    for tag, lg, nt in (("all_cores", log2_indices, cores), ("single_core", max(6, log2_indices - 4), 1)):
        n = 1 << lg
        ch = P.initialization(P.Params(CURVES[curve], P.CHUNKED, 0, n, lg, n))
        t0 = time.time()
        cref.computation(curve, 0, 0, n, lg, ch, tau, alpha, beta, nthreads=nt)
        dt = time.time() - t0
        out[tag] = {"value": n / dt, "sample": "Phase1::computation (oracle/c) on chunk 0 of a 2^%d ceremony: %d G1 + %d G2 updates, %d threads, %.1f s wall"
                                               % (lg, 3 * n, n + 1, nt, dt)}
    out["value"] = out["all_cores"]["value"]
    return out


def cpu_verify_baseline(log2n=20):
    """CPU denominator of "chunks verified/s" (--config 5): what the oracle spends per response element of a BLS12-377 chunk —
    sqrt decompression, prime-order membership by [r]P (double-and-add) and its share of the random-linear-combination MSM
    (bucket method, 127-bit coefficients as the GPU path draws them) — timed on bounded samples on every usable core
    (slices on threads: the C calls release the GIL) and scaled to the 3 x 2^n G1 + 2^n G2 elements of chunk 0.  BLAKE2b and
    the pairings are left out (they are host work in both implementations)."""
    import random
    from concurrent.futures import ThreadPoolExecutor
    from oracle import cref
    from oracle.pyref.curves import CURVES
    curve, cores = "bls12_377", usable_cores()
    cref.lib()
    c = CURVES[curve]
    rng = random.Random(7)
    per_elem = {}
    for group, sample in ((1, 1 << 12), (2, 1 << 10)):
        nl = cref.fr_limbs(curve)
        gen = cref.scalar_mul(curve, group, _generator_bytes(curve, group), 1, 1, nl, shared=True)
        usz, csz = cref.point_size(curve, group, False), cref.point_size(curve, group, True)
        # `sample` distinct subgroup points: k_i * G
        pts = cref.scalar_mul(curve, group, gen * sample, sample, [rng.randrange(1, c.r) for _ in range(sample)], nl)
        comp = cref.scalar_mul(curve, group, pts, sample, 1, nl, shared=True, compressed_out=True)
        per = max(1, sample // cores)
        slices = [(i, min(sample, i + per)) for i in range(0, sample, per)]
        with ThreadPoolExecutor(max_workers=cores) as ex:
            t0 = time.time()
            dec = cref.decompress(curve, group, comp, sample, nthreads=cores)
            t1 = time.time()
            list(ex.map(lambda ab: cref.scalar_mul(curve, group, pts[ab[0] * usz:ab[1] * usz], ab[1] - ab[0], c.r, nl, shared=True), slices))
            t2 = time.time()
            rho = [rng.getrandbits(127) for _ in range(sample)]
            list(ex.map(lambda ab: cref.msm(curve, group, pts[ab[0] * usz:ab[1] * usz], ab[1] - ab[0], rho[ab[0]:ab[1]], nl), slices))
            t3 = time.time()
        assert dec == pts
        per_elem[group] = {"decompress_s": (t1 - t0) / sample, "subgroup_s": (t2 - t1) / sample, "msm_s": 2 * (t3 - t2) / sample,
                           "sample": sample}
    n = 1 << log2n
    chunk_s = sum(per_elem[1][k] for k in ("decompress_s", "subgroup_s", "msm_s")) * 3 * n + \
        sum(per_elem[2][k] for k in ("decompress_s", "subgroup_s", "msm_s")) * n
    # the GPU path now runs setup-utils' BATCHED membership test on vectors this long (riding on the MSM); a CPU running the
    # same mode would pay about one more pass of bucket additions instead of [r]P per element: lower bound = no membership cost
    chunk_s_batched = sum(per_elem[1][k] for k in ("decompress_s", "msm_s")) * 3 * n + sum(per_elem[2][k] for k in ("decompress_s", "msm_s")) * n
    return {"value": 1.0 / chunk_s, "value_if_membership_were_free": 1.0 / chunk_s_batched, "unit": "chunks verified/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "sample": "C oracle (oracle/c) on %d G1 + %d G2 elements, %d threads: per element sqrt decompression + [r]P membership + "
                      "2 x MSM share (power_pairs sums the vector and its shift); scaled to 3 x 2^%d G1 + 2^%d G2 elements; BLAKE2b and "
                      "pairings excluded" % (per_elem[1]["sample"], per_elem[2]["sample"], cores, log2n, log2n),
            "seconds_per_element": {"g1": {k: v for k, v in per_elem[1].items() if k != "sample"},
                                    "g2": {k: v for k, v in per_elem[2].items() if k != "sample"}},
            "projected_chunk_seconds": chunk_s}


def _generator_bytes(curve, group):
    from oracle.pyref import serde
    from oracle.pyref.curves import CURVES
    from oracle.pyref.ec import Group
    G = Group(CURVES[curve], group)
    return serde.write_point(G, G.gen, False)


def initial_challenge_file(pkg, curve, prm, path, hash_path):
    """phase1_cli::new_challenge; on MNT4-753 / MNT6-753 through the caller-supplied-generators form (the product holds no
    G2 generator for them: the golden fixture's synthetic order-r point is read as data)."""
    if not curve.startswith("mnt"):
        pkg.new_challenge(path, hash_path, prm)
        return
    L = pkg.load_library()
    cid = pkg.phase1_cli.curve_from_str(curve)
    chal, n1 = golden_generator(curve, 2)
    u1, u2 = L.ptau_point_size(cid, 1, 0), L.ptau_point_size(cid, 2, 0)
    buf = pkg.phase1_cli.initialization_with_generators(prm, pkg.phase1_cli.generator(curve, 1), chal[64 + n1 * u1:64 + n1 * u1 + u2])
    open(path, "wb").write(buf)
    open(hash_path, "wb").write(pkg.calculate_hash(buf))


def host_cpu_share():
    """Host threads this process may actually run at once: the affinity mask, capped by the cgroup's CPU quota (a container
    on a 256-thread host with `cpu.max = 1600000 100000` has 16)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:   # noqa: BLE001
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:   # noqa: BLE001
            continue
    return n


def pick_workdir(need_bytes, prefix):
    """A scratch directory for the file-level measurements with room for `need_bytes`: /dev/shm (tmpfs: the files are
    mmap'd, written by DMA and hashed, so a disk would measure the disk), else the default temp dir; a CLEAN error when neither
    has the room — a tmpfs that fills up ends in SIGBUS inside an mmap'd write, not in an error message."""
    import shutil
    import tempfile
    cands = [c for c in ("/dev/shm", tempfile.gettempdir()) if os.path.isdir(c) and os.access(c, os.W_OK)]
    seen = []
    for c in cands:
        free = shutil.disk_usage(c).free
        seen.append("%s: %.1f GB free" % (c, free / 1e9))
        if free >= int(need_bytes * 1.15) + (64 << 20):
            return tempfile.mkdtemp(prefix=prefix, dir=c), c
    raise SystemExit("bench.py: the file-level measurement needs %.1f GB of scratch space (%s) — free some, or lower --lanes / --gpus"
                     % (need_bytes * 1.15 / 1e9, "; ".join(seen) or "no writable temp dir"))


def whole_chunk_extras(pkg, curve, power, log2n, dev):
    """Second half of the BASELINE metric ("chunks verified/sec") and the whole-chunk contribution, on chunk 0 of a
    power-`power` ceremony of `curve` with chunk_size 2^log2n (all five vectors), through the FILE-level
    entry points the operator calls (phase1_cli::contribute / transform_pok_and_correctness: mmap'd files on tmpfs,
    BLAKE2b of every file, `.hash` side files, PCIe both ways, pairing tail).  Untimed set-up: new_challenge -> one
    contribution; timed: verify (ratio_check on / off) and a second contribution of the verified new challenge."""
    import shutil
    import tempfile
    n = 1 << log2n
    prm = pkg.Phase1Parameters.new_chunk(curve, 0, 0, n, 0, power, n)
    n1, n2 = prm.g1_chunk_size, prm.other_chunk_size
    seed0, seed1 = b"bench seed 0".ljust(64, b"."), b"bench seed 1".ljust(64, b".")
    # plain Auto is what the operator integration passes (a downloaded challenge is data of unknown origin);
    # the asserted form (a caller that ran CheckForCorrectness::Full itself) is measured beside it
    verified = int(pkg.BatchExpMode.Auto) | pkg.phase1_cli.INPUT_IN_SUBGROUP
    # inputs + outputs of the single calls, up to 8 lanes' outputs at a time, the combined accumulator of the tail
    d, base = pick_workdir(3 * prm.accumulator_size + 3 * prm.contribution_size + 8 * prm.accumulator_size, "ptau_bench_")
    f = lambda name: os.path.join(d, name)
    elems = n1 + 3 * n2 + 1
    out = {"host_blake2b": (lambda v: {0: "portable", 1: "avx2", 2: "bmi2", 3: "avx512"}.get(v, str(v)))(pkg.load_library().ptau_blake2b_implementation(-1)),
           "workload": "%s power %d chunk 0, chunk_size 2^%d: %d G1 + %d G2 elements; challenge %d B, response %d B; "
                       "file-level entry points, files on %s" % (curve, power, log2n, n1 + 2 * n2, n2 + 1, prm.accumulator_size,
                                                                  prm.contribution_size, base or "the default tmp dir")}
    try:
        initial_challenge_file(pkg, curve, prm, f("c0"), f("c0.hash"))
        pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.hash2"), f("r0"), f("r0.hash"), pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto,
                                         prm, seed0, device=dev)
        t0 = time.perf_counter()
        pkg.transform_pok_and_correctness(f("c0"), f("v.c.hash"), pkg.CheckForCorrectness.No, f("r0"), f("v.r.hash"),
                                          pkg.CheckForCorrectness.OnlyNonZero, f("c1"), f("c1.hash"), pkg.SubgroupCheckMode.Auto,
                                          True, prm, device=dev)
        dt = time.perf_counter() - t0
        tm = pkg.phase1_cli.last_timing(dev)
        out["verify_chunk"] = {"seconds": dt, "chunks_verified_per_s": 1.0 / dt, "elements_per_s": elems / dt,
                               "includes": "3 BLAKE2b digests (one host thread per file, overlapped with the device), compute_g2_s + 3 proofs "
                                           "of knowledge + 5 chunk-0 before/after checks, sqrt decompression, non-zero + subgroup check of "
                                           "every element, power_pairs of 4 vectors, 4 power-ratio pairing checks, new challenge written, "
                                           "PCIe both ways",
                               "device_ms": tm["kernel_ms"], "decompress_ms": tm["decompress_ms"], "subgroup_ms": tm["subgroup_ms"],
                               "msm_ms": tm["msm_ms"], "h2d_ms": tm["h2d_ms"], "d2h_ms": tm["d2h_ms"]}
        os.remove(f("c1"))
        t0 = time.perf_counter()
        pkg.transform_pok_and_correctness(f("c0"), f("v.c.hash"), pkg.CheckForCorrectness.No, f("r0"), f("v.r.hash"),
                                          pkg.CheckForCorrectness.OnlyNonZero, f("c1"), f("c1.hash"), pkg.SubgroupCheckMode.Auto,
                                          False, prm, device=dev)
        dt2 = time.perf_counter() - t0
        out["verify_chunk_skip_ratio_check"] = {"seconds": dt2, "chunks_verified_per_s": 1.0 / dt2}
        modes = [("contribute_chunk_exact", pkg.BatchExpMode.Auto, "Auto, CheckForCorrectness::No — the operator's default flags, exact on any curve "
                  "point: batched membership probe of each vector, endomorphism kernels where it passes (it does: verified challenge)"),
                 ("contribute_chunk_direct", pkg.BatchExpMode.Direct, "Direct: plain signed windows, no probe")]
        modes.insert(0, ("contribute_chunk", verified, "Auto | INPUT_IN_SUBGROUP (challenge produced by the verification above: endomorphism path"
                                                       + (" — on the MNT curves that is GLS on the G2 vector; G1 has no endomorphism)" if curve.startswith("mnt") else ")")))
        for tag, mode, what in modes:
            if os.path.exists(f("r1")):
                os.remove(f("r1"))
            t0 = time.perf_counter()
            pkg.phase1_cli.contribute_seeded(f("c1"), f("c1.hash2"), f("r1"), f("r1.hash"), pkg.CheckForCorrectness.No, mode, prm, seed1, device=dev)
            dt3 = time.perf_counter() - t0
            tm = pkg.phase1_cli.last_timing(dev)
            out[tag] = {"seconds": dt3, "chunks_per_s": 1.0 / dt3, "g1_equiv_updates_per_s": elems / dt3, "batch_exp_mode": what,
                        "includes": "phase1_cli::contribute on files from a seed: key generation, BLAKE2b(challenge) overlapped, 5 vectors "
                                    "updated (PCIe both ways), public key + BLAKE2b(response) + msync",
                        "device_ms": tm["kernel_ms"], "membership_probe_ms": tm["subgroup_ms"], "h2d_ms": tm["h2d_ms"], "d2h_ms": tm["d2h_ms"]}
        assert open(f("c1"), "rb").read(64) == open(f("v.r.hash"), "rb").read()   # new challenge extends the response it verified
        if curve != "bls12_377":
            return out
        # several in-process lanes on ONE GPU (the operator's --max-in-process-lane, contribute.rs:120-121,467-492): the
        # library serialises only the device section of a call, so one lane's BLAKE2b overlaps another lane's kernels
        import threading
        Lb = pkg.load_library()
        for lanes in (1, 2, 3, 4, 6, 8):
            def lane_verify(k):
                pkg.transform_pok_and_correctness(f("c0"), f("L%d.c.hash" % k), pkg.CheckForCorrectness.No, f("r0"), f("L%d.r.hash" % k),
                                                  pkg.CheckForCorrectness.OnlyNonZero, f("L%d.c1" % k), f("L%d.c1.hash" % k),
                                                  pkg.SubgroupCheckMode.Auto, True, prm, device=dev)
            def lane_contribute(k):
                pkg.phase1_cli.contribute_seeded(f("c1"), f("L%d.h" % k), f("L%d.r1" % k), f("L%d.r1.hash" % k), pkg.CheckForCorrectness.No,
                                                 pkg.BatchExpMode.Auto, prm, seed1, device=dev)
            for name, fn, outs in (("verify", lane_verify, ("L%d.c1",)), ("contribute", lane_contribute, ("L%d.r1",))):
                th = [threading.Thread(target=fn, args=(k,)) for k in range(lanes)]
                busy0 = Lb.ptau_device_busy_ms(dev)
                t0 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                dtl = time.perf_counter() - t0
                busy = (Lb.ptau_device_busy_ms(dev) - busy0) / 1e3 / dtl
                for k in range(lanes):
                    for o in outs:
                        assert os.path.getsize(f(o % k)) > 0
                        os.remove(f(o % k))
                out.setdefault("lanes", {})["%s_%d_lanes" % (name, lanes)] = {"seconds": dtl, "chunks_per_s": lanes / dtl, "gpu_busy": busy}
        # whole-transcript tail (BASELINE configs[4]: verify_transcript's combine + transform_ratios): the round's second
        # chunk (tauG1 only, 2^n - 1 elements) contributed with the same key, both responses combined, ratio checks run
        prm1 = pkg.Phase1Parameters.new_chunk(curve, 0, 1, n, 0, log2n, n)
        full = pkg.Phase1Parameters.new_full(curve, 0, log2n, n)
        pkg.new_challenge(f("d0"), f("d0.hash"), prm1)
        pkg.phase1_cli.contribute_seeded(f("d0"), f("d0.hash2"), f("s0"), f("s0.hash"), pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto, prm1,
                                         seed0, device=dev)
        responses = [open(f("r0"), "rb").read(), open(f("s0"), "rb").read()]
        t0 = time.perf_counter()
        combined = pkg.combine(responses, n, full, device=dev)
        dt4 = time.perf_counter() - t0
        t0 = time.perf_counter()
        pkg.transform_ratios(combined, pkg.CheckForCorrectness.No, full, device=dev)
        dt5 = time.perf_counter() - t0
        out["transcript_tail"] = {"combine_seconds": dt4, "transform_ratios_seconds": dt5, "combined_bytes": len(combined),
                                  "includes": "combine: sqrt decompression of both chunk responses into the Full-mode accumulator; "
                                              "transform_ratios: power_pairs (bucket MSM) over tauG1 (2^%d - 1), tauG2, alphaG1, betaG1 "
                                              "(2^%d each) + 5 pairing checks; host buffers" % (log2n + 1, log2n)}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


# ---- per-config workloads (BASELINE.json configs[1..4]) -----------------------------------------------------------------
# (curve, group, log2 points of the vector one step updates, ceremony power, log2 chunk_size, limbs of the base field as the
#  kernels hold it, extension degree, GLV/GLS available).  Algorithmic bytes per update = uncompressed read + compressed
# write = 3 x the compressed size: 144 / 288 / 285 / 570 / 855 B.
CONFIGS = {
    2: {"groups": [("bls12_377", 1, 20), ("bls12_377", 2, 20)], "power": 20, "log2_chunk": 20,
        "workload": "BLS12-377 phase-1 power 20, chunk_size 2^20: tauG1 vector of one chunk per GPU"},
    3: {"groups": [("bw6", 1, 18), ("bw6", 2, 18)], "power": 20, "log2_chunk": 18,
        "workload": "BW6-761 phase-1 power 20, chunk_size 2^18 (8 chunks, one per GPU): tauG1 vector of one chunk per GPU"},
    4: {"groups": [("mnt4_753", 1, 16), ("mnt4_753", 2, 16), ("mnt6_753", 1, 16), ("mnt6_753", 2, 16)], "power": 18, "log2_chunk": 16,
        "workload": "MNT4-753 + MNT6-753 phase-1 power 18, chunk_size 2^16: tauG1 vector of one MNT4-753 chunk per GPU"},
}
KERNEL_TAG = {("bls12_377", 1): "BlsG1", ("bls12_377", 2): "BlsG2", ("bw6", 1): "Bw6G1", ("bw6", 2): "Bw6G2",
              ("mnt4_753", 1): "Mnt4G1", ("mnt4_753", 2): "Mnt4G2", ("mnt6_753", 1): "Mnt6G1", ("mnt6_753", 2): "Mnt6G2"}
FIELD_LIMBS = {"bls12_377": 13, "bw6": 28, "mnt4_753": 26, "mnt6_753": 26}     # limbs of 29 bits (BW6-761: 28, one spare, so the element splits over a lane pair)
SCALAR_BITS = {"bls12_377": 253, "bw6": 377, "mnt4_753": 753, "mnt6_753": 753}
MAC_PEAK = 32.7e12        # measured v_mad_u64_u32 issue peak, lane-MACs/s chip-wide (profiles/r01_ubench_issue.txt)


def macs_per_update(curve, group, endo):
    """32x32->64 multiply-accumulates of ONE update (all lanes of the point together), from the formulas the kernels run
    (DESIGN.md §4).  Base products, per point:
      FpZ   (BLS12-377, 14 limbs of 28 bits, one lane per coefficient; p = 1 mod 2^28): product 378, square 287
      FpLZ  (BW6-761 and the MNT fields: 28 limbs of 28 bits over a lane pair): product 2 N^2 = 1568, square 1568 - 2 * 91 = 1386,
            fused k-term product (ONE reduction, FpLZ::row_mul_multi): 784 (k + 1)
    Tower products: BLS12-377 Fq2 over a lane pair = 2 products per lane (schoolbook), square = 1 per lane (complex);
    MNT4 Fq2 over a quad = one fused 2-term product per pair, square = one product per pair (complex);
    MNT6 Fq3 over six lanes = one fused 3-term product per pair, square = one fused 2-term product per pair.
    Point formulas: a = 0: doubling 2M + 5S, mixed addition on the common-Z table 8M + 3S (G2: 7M + 4S), table = 7 mixed
    additions + ~50 products + 8 squares to rescale; a != 0 (MNT): doubling 1M + 8S, full addition 11M + 5S, table = 7 additions.
    Signed 4-bit windows, 15/16 of the digits non-zero."""
    bits = SCALAR_BITS[curve]
    windows = (bits + 3 + 3) // 4
    # FpZ (14 limbs of 28 bits, p = 1 mod 2^28: the m_i p_0 terms are additions), counted in the ISA:
    # product 196 + 182 = 378, square 105 + 182 = 287, fused two-term product (ps_mul2, one reduction) 392 + 182 = 574
    zm, zs, zm2 = 378, 287, 574
    if curve == "bls12_377" and group == 1:
        m, sq = zm, zs
        dbl, madd = 2 * m + 5 * sq, 8 * m + 3 * sq
        table = 7 * madd + 50 * m + 8 * sq
        if endo:
            return (4 * 32 - 4) * dbl + 2 * 32 * 15 // 16 * madd + 32 * 15 // 16 * m + m + table
        return (4 * windows - 4) * dbl + windows * 15 // 16 * madd + table
    if curve == "bls12_377":      # G2: Fq2 over a lane pair, FpZ coefficients
        m, sq = 2 * zm2, 2 * zm                                        # per lane ONE fused two-term product; complex square = one product
        dbl, madd = 2 * m + 5 * sq, 7 * m + 4 * sq
        table = 7 * madd + 50 * m + 8 * sq + 10 * 2 * zm
        if endo:      # GLS: 17 windows of four 64-bit sub-scalars, psi^j on 3 of 4 entries (two Fq2-by-Fq products)
            return (4 * 17 - 4) * dbl + 4 * 17 * 15 // 16 * madd + 3 * 17 * 15 // 16 * 2 * 2 * zm + table
        return (4 * windows - 4) * dbl + windows * 15 // 16 * madd + table
    m, sq = 2 * 28 * 28, 2 * 28 * 28 - 2 * 91
    if curve == "bw6":
        dbl, madd = 2 * m + 5 * sq, 8 * m + 3 * sq
        table = 7 * madd + 50 * m + 8 * sq
        if endo:
            return (4 * 48 - 4) * dbl + 2 * 48 * 15 // 16 * madd + 48 * 15 // 16 * m + m + table
        return (4 * windows - 4) * dbl + windows * 15 // 16 * madd + table
    # MNT4 / MNT6 (k1_lazy_a.hpp): a != 0, Jacobian table entries
    fused = lambda k: 784 * (k + 1)
    deg = 1 if group == 1 else (2 if curve == "mnt4_753" else 3)
    if group == 1:
        M, S = m, sq
    elif curve == "mnt4_753":
        M, S = 2 * fused(2), 2 * m
    else:
        M, S = 3 * fused(3), 3 * fused(2)
    dbl, add = M + 8 * S, 11 * M + 5 * S
    io = 5 * m * deg   # 2 converting products in, 3 out, per coefficient
    if endo and group == 2:
        # GLS on the twists (psi = [t - 1]): two 377-bit halves = 96 interleaved windows over ONE table; psi of
        # a Jacobian entry on the second half's additions = one base product per lane pair and coordinate (3 coordinates x deg pairs)
        w2 = 96
        return (4 * w2 - 4) * dbl + (2 * w2 * 15 // 16 + 7) * add + w2 * 15 // 16 * 3 * deg * m + io
    return (4 * windows - 4) * dbl + (windows * 15 // 16 + 7) * add + io


def sqrt_macs(curve="bls12_377"):
    """32x32->64 multiply-accumulates of ONE base-field square root as decompress_kernel runs it (fp29.hpp sqrt_ex; BLS12-377 Fq,
    13 limbs of 29 bits, fully reduced: product 325, square 247 MACs in the ISA): a^((T-1)/2) by sliding 4-bit windows over the
    330-bit exponent (329 squares + 66 products + an 8-entry table: 1 + 7), x = a w and b = x w (2 products), the Pohlig-Hellman
    stage in the 2^46 torsion on ONE squaring chain (37 squares + 20 products for the windows' corrections + 5 for g^-d;
    round 3: 111 squares + 12 products), the root (1 product) and its check (1 square)."""
    assert curve == "bls12_377"
    m, sq = 325, 247
    return (329 + 1 + 37 + 1) * sq + (66 + 7 + 2 + 25 + 1) * m


def decompress_macs_per_chunk(n):
    """decompress_kernel over a 2^20 BLS12-377 chunk: 3 n G1 elements (one square root, x^3 + b, Montgomery conversions and the
    sign: ~6 products more) and n + 1 G2 elements (complex method: the norm, sqrt(norm), then sqrt_or_nr of (a0 + alpha) / 2 — one
    exponentiation serves both candidates, the torsion stage runs twice for half the elements — and ~25 products of glue)."""
    g1 = sqrt_macs() + 6 * 325
    g2 = 2 * sqrt_macs() + (37 * 247 + 25 * 325) // 2 + 25 * 325
    return 3 * n * g1 + (n + 1) * g2


def msm_accumulate_macs_per_chunk(n, windows=11):
    """msm_accumulate_kernel over the four power_pairs reductions of a 2^20 BLS12-377 chunk verification: every point lands in one
    bucket per window (signed 13-bit digits of a 142-bit coefficient: 11 windows; a 2^-13 share of zero digits ignored) in BOTH
    sums (the vector and its shift), one mixed addition each — G1 (one lane, FpZ): 8 products + 3 squares; G2 (lane pair,
    fused two-term products): 7 Fq2 products + 4 Fq2 squares + the two converting products per coordinate and lane."""
    zm, zs, zm2 = 378, 287, 574
    g1 = 8 * zm + 3 * zs
    g2 = 7 * 2 * zm2 + 4 * 2 * zm + 2 * 2 * zm
    return 2 * windows * (3 * n * g1 + n * g2)


def pmc_for(tag, config, prefix="window_scalar_mul"):
    """Counters of window_scalar_mul*kernel<tag> from the rocprofv3 PMC summary of THIS config and the NEWEST round that has one
    (profiles/rNN_pmc_cfg<config>.json, written by tools/gpu_round.sh before the bench line is taken).  No fallback to
    another config's file: a summary that lacks the kernel yields None."""
    import glob
    import re
    files = glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_cfg%d.json" % config))
    env = os.environ.get("PTAU_BENCH_PMC")          # gpu_round.sh points at the summary it has just written
    if env and os.path.exists(env):
        files = [env]
    best = None
    for f in sorted(files, key=lambda p: int(re.search(r"r(\d+)", os.path.basename(p)).group(1)) if re.search(r"r(\d+)", os.path.basename(p)) else -1):
        try:
            k = json.load(open(f)).get("kernels", {})
        except Exception:
            continue
        for name, d in k.items():
            if name.startswith(prefix) and ("<%s>" % tag) in name and "hbm_bytes_per_launch" in d:
                best = (d, os.path.basename(f))
    return best if best else (None, None)


def golden_generator(curve, group):
    """Uncompressed generator bytes.  MNT G2: the product holds none — element 0 of tauG2 of the committed
    golden challenge (a synthetic order-r point; any order-r base serves a throughput measurement) is read as data."""
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "phase1_%s.json" % curve)))
    ch = doc["chunks"][0]
    chal = bytes.fromhex(ch["challenge"])
    n1 = min(doc["chunk_size"], (2 << doc["power"]) - 1)
    return chal, n1


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank, world, port, base_env=None):
    """Environment of rank `rank` of a `world`-rank job on this node (what torchrun would export)."""
    env = dict(os.environ if base_env is None else base_env)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                "PTAU_BENCH_CHILD": "1"})
    return env


def launch_ranks(world, argv, script=None, popen=None, timeout=None):
    """`python bench.py --gpus N` without torchrun: the parent — which has imported neither torch nor the HIP library, so no
    process that initialised the GPU ever forks or execs — starts one child per GPU with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, relays rank 0's stdout (the ONE JSON line) and every rank's stderr, and exits non-zero if any rank failed
    (the survivors are terminated: a dead rank would leave them in the barrier)."""
    import subprocess
    popen = popen or subprocess.Popen
    script = script or os.path.abspath(__file__)
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(popen([sys.executable, script] + list(argv), env=rank_env(r, world, port),
                           stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
    deadline = time.time() + timeout if timeout else None
    rc = 0
    out0 = b""
    pending = set(range(world))
    try:
        while pending:
            for r in sorted(pending):
                p = procs[r]
                if r == 0 and p.stdout is not None:
                    try:
                        o, _ = p.communicate(timeout=0.5)
                        out0 += o or b""
                    except subprocess.TimeoutExpired:
                        pass
                code = p.poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write("bench.py: rank %d exited with code %d\n" % (r, code))
            if rc or (deadline and time.time() > deadline):
                if not rc:
                    rc = 124
                    sys.stderr.write("bench.py: ranks timed out\n")
                break
            if pending and 0 not in pending:
                time.sleep(0.2)
    finally:
        for r in pending:
            try:
                procs[r].terminate()
            except Exception:
                pass
        for r in pending:
            try:
                procs[r].wait(timeout=20)
            except Exception:
                try:
                    procs[r].kill()
                except Exception:
                    pass
    if out0:
        # rank 0's ONE JSON line goes to stdout; anything else a library printed there (gloo's "[Gloo] Rank 0 is connected ...")
        # is passed on through stderr, so that stdout stays the one line the contract promises
        for line in out0.decode(errors="replace").splitlines():
            if line.lstrip().startswith("{"):
                sys.stdout.write(line + "\n")
            elif line.strip():
                sys.stderr.write(line + "\n")
        sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5),
                    help="BASELINE.json configs[] index: 2 = BLS12-377 2^20 G1 tau-update (the metric's config, default), 3 = BW6-761 chunk, "
                         "4 = MNT4/MNT6 chunks, 5 = verification of a 2^20 BLS12-377 chunk")
    ap.add_argument("--log2-points", type=int, default=None, help="override the vector length of the config")
    ap.add_argument("--cpu-sample", type=int, default=None, help="points the CPU oracle updates (parity check + cpu_baseline)")
    ap.add_argument("--no-extras", action="store_true", help="skip the whole-chunk contribute / verify measurements")
    ap.add_argument("--lanes", type=int, default=0,
                    help="with --config 2 / 5: WHOLE chunks through the file-level entry points with this many in-process lanes per GPU "
                         "(the operator's --max-in-process-lane); the line's value is chunks contributed / verified per second over all ranks")
    ap.add_argument("--exact-only", "--value-path-only", dest="exact_only", action="store_true",
                    help="launch the path of `value` only (BatchExpMode::Auto; no Direct / asserted-membership lines): under rocprofv3 "
                         "every launch of the kernel is then the one `value` and `roofline` describe, so the profile's per-kernel "
                         "averages are comparable to them")
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher around it: start the N ranks here, before torch or HIP are touched
    # (under torchrun / as a child RANK is set and this process IS a rank)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus) and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size is used\n" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # rehearsal knobs for a one-GPU box (the driver's N > 1 runs use neither): PTAU_BENCH_BACKEND=gloo lets several ranks
    # share one device (RCCL refuses duplicate GPUs), PTAU_BENCH_DEVICE pins every rank to that device
    backend = os.environ.get("PTAU_BENCH_BACKEND", "nccl")
    if "PTAU_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["PTAU_BENCH_DEVICE"])
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs GPU %d but this node has %d (one rank per GPU; PTAU_BENCH_BACKEND=gloo "
                         "PTAU_BENCH_DEVICE=0 rehearses N ranks on one device)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    pkg = load_pkg()
    L = pkg.load_library()
    dev = local_rank
    eb = ctypes.create_string_buffer(256)
    ENDO = int(pkg.BatchExpMode.Auto) | pkg.phase1_cli.INPUT_IN_SUBGROUP
    DIRECT = int(pkg.BatchExpMode.Direct)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.lanes and args.config not in (2, 5):
        raise SystemExit("bench.py: --lanes measures whole BLS12-377 chunks: use it with --config 2 (contribute) or --config 5 (verify)")
    if args.config == 5:
        return bench_verify(args, pkg, dev, rank, world, dist, backend, barrier)
    if args.lanes:
        return bench_lanes(args, pkg, dev, rank, world, dist, backend, barrier, verify=False)

    cfg = CONFIGS[args.config]

    def make_vector(curve, group, n, start):
        """A genuine accumulator vector [s^(start+i) G] of n uncompressed points, made on the device (untimed)."""
        cid = pkg.phase1_cli.curve_from_str(curve)
        usz, csz = L.ptau_point_size(cid, group, 0), L.ptau_point_size(cid, group, 1)
        if curve.startswith("mnt") and group == 2:
            chal, n1 = golden_generator(curve, group)
            u1 = L.ptau_point_size(cid, 1, 0)
            gen = chal[64 + n1 * u1:64 + n1 * u1 + usz]
        else:
            gen = pkg.phase1_cli.generator(curve, group)
        nl = L.ptau_fr_limbs(cid)
        d_gen = torch.frombuffer(bytearray(gen * n), dtype=torch.uint8).cuda()
        d_chal = torch.empty(n * usz, dtype=torch.uint8, device="cuda")
        d_resp = torch.empty(n * csz, dtype=torch.uint8, device="cuda")
        s = det_scalar("bench/s") % (1 << 250)

        def run(d_in, cin, d_out, cout, tau, mode=0):
            rc = L.ptau_apply_powers_dev(cid, group, ctypes.c_void_p(d_in.data_ptr()), cin, ctypes.c_void_p(d_out.data_ptr()), cout,
                                         n, start, pkg._lib.limbs(tau, nl), None, 0, mode, dev, eb, len(eb))
            if rc:
                raise RuntimeError("ptau_apply_powers_dev rc=%d: %s" % (rc, eb.value.decode()))
        run(d_gen, 0, d_chal, 0, s, 0)   # the challenge vector itself: s^i * generator
        del d_gen
        return d_chal, d_resp, run, usz, csz

    # ---- the timed workload: first group of the config -----------------------------------------------------------------
    curve, group, lg = cfg["groups"][0]
    if args.log2_points is not None:
        lg = args.log2_points
    n = 1 << lg
    start = rank * n      # rank r owns chunk index r (chunk_size = n): independent chunks, no data-path collective
    tau = det_scalar("bench/tau") % (1 << 250)      # below every curve's group order
    d_chal, d_resp, run, usz, csz = make_vector(curve, group, n, start)
    for _ in range(args.warmup):
        run(d_chal, 0, d_resp, 1, tau)
    barrier()
    busy0 = L.ptau_device_busy_ms(dev)
    t0 = time.perf_counter()
    smul_ms = norm_ms = probe_ms = 0.0
    for _ in range(args.steps):
        run(d_chal, 0, d_resp, 1, tau)
        tm = pkg.phase1_cli.last_timing(dev)      # HIP events on the library's own stream
        smul_ms += tm["scalar_mul_ms"]
        norm_ms += tm["normalize_ms"]
        probe_ms += tm["subgroup_ms"]
    barrier()
    dt = time.perf_counter() - t0
    # kernel time of THIS rank's device over the timed region (ptau_device_busy_ms): one entry per rank in the line
    busy = [(L.ptau_device_busy_ms(dev) - busy0) / 1e3 / dt]
    if dist is not None:
        allb = [None] * world
        dist.all_gather_object(allb, busy[0])
        busy = allb
    # the same step with the membership ASSERTED by the caller (what an operator that trusts its coordinator's verifier runs: no
    # probe) and on the plain path (BatchExpMode::Direct): reported next to `value`, never as `value`
    def side_line(mode):
        run(d_chal, 0, d_resp, 1, tau, mode)
        k_ms = 0.0
        te = time.perf_counter()
        for _ in range(args.steps):
            run(d_chal, 0, d_resp, 1, tau, mode)
            k_ms += pkg.phase1_cli.last_timing(dev)["scalar_mul_ms"]
        torch.cuda.synchronize()
        return (n * args.steps / (time.perf_counter() - te), k_ms / args.steps)
    endo = direct = None
    if not args.exact_only:
        if not curve.startswith("mnt"):
            endo = side_line(ENDO)
        direct = side_line(DIRECT)
        run(d_chal, 0, d_resp, 1, tau)      # d_resp again holds the value path's output (parity sample below)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        value = world * n * args.steps / dt
        kernel_s = (smul_ms / args.steps) / 1e3
        algo = 3 * csz
        achieved = algo * n / kernel_s / 1e9
        # parity check + CPU baseline on a bounded sample of the same vector (the timed baseline only at N = 1: at
        # N > 1 the other ranks would idle behind rank 0's host cores, so only a small parity sample is checked)
        default_sample = {"bls12_377": 1 << 20, "bw6": 1 << 16, "mnt4_753": 1 << 15, "mnt6_753": 1 << 15}[curve]
        sample = min((args.cpu_sample if args.cpu_sample is not None else default_sample) if world == 1 else 2048, n)
        chal_host = d_chal[:sample * usz].cpu().numpy().tobytes()
        ref, base = cpu_baseline(curve, group, sample, chal_host, start, tau, single_core=(world == 1))
        if world > 1:
            base = None
        got = d_resp[:sample * csz].cpu().numpy().tobytes()
        if got != ref:
            raise SystemExit("PARITY FAILURE: GPU response differs from the CPU oracle on the sampled points")
        tag = KERNEL_TAG[(curve, group)]
        pmc, pmc_src = pmc_for(tag, args.config)
        traffic = pmc["hbm_bytes_per_launch"] if pmc else None
        traffic_raw = pmc.get("hbm_bytes_per_launch_raw_fetch") if pmc else None
        probed = probe_ms > 0     # Auto's membership probe ran and, the vector being genuine, the endomorphism kernel followed
        macs = macs_per_update(curve, group, endo=probed)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": pmc_src,
                "traffic_ratio": (traffic / (algo * n)) if traffic else None,
                "traffic_raw_fetch": traffic_raw,
                "traffic_note": "FETCH_SIZE x 2 + WRITE_SIZE (the guide's gfx950 correction holds for 16-B-per-lane streaming reads; "
                                "this kernel's table reads are dword / dwordx2 per lane: traffic_raw_fetch = FETCH_SIZE + WRITE_SIZE is "
                                "the uncorrected figure, tools/ubench_fetch.hip calibrates the two access patterns — "
                                "profiles/r03_ubench_fetch.txt); Infinity-Cache hits are counted",
                "algorithmic_bytes_per_update": algo, "updates_per_launch": n,
                "kernel": "window_scalar_mul_kernel<%s>" % tag, "kernel_ms": smul_ms / args.steps,
                "normalize_kernel_ms": norm_ms / args.steps, "membership_probe_ms": probe_ms / args.steps,
                "macs_per_update": macs, "alu_peak_mac_per_s": MAC_PEAK,
                "alu_frac": (macs * n / kernel_s / MAC_PEAK) if macs else None,
                "note": "big-integer arithmetic: the kernel is bound by integer-multiplier issue, not by HBM — alu_frac (32x32->64 "
                        "MACs/s over the measured v_mad_u64_u32 issue peak, profiles/r01_ubench_issue.txt) is the figure that "
                        "measures kernel quality; frac is the contract's HBM fraction (DESIGN.md §4)"}
        name = ctypes.create_string_buffer(128)
        L.ptau_device_info(dev, name, 128, None, None)
        out = {
            "metric": "G1 point-updates/sec per chunk (contribute)", "value": value, "unit": "G1 point-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": cfg["workload"] + " (2^%d %s G%d points, uncompressed in -> compressed out, HBM-resident)" % (lg, curve, group),
                       "baseline_config": args.config, "points_per_step_per_gpu": n, "device": name.value.decode(),
                       "parity": "bit-exact vs CPU oracle on %d sampled points" % sample},
            "roofline": roof,
            "device_busy_frac_per_rank": busy,
            "cpu_baseline": base,
            "value_path": "BatchExpMode::Auto with CheckForCorrectness::No (the operator's contribute default), exact for ANY curve "
                          "point: " + ("batched membership probe of the vector (2^-128, inside every step), then the endomorphism "
                                       "(GLV / GLS) window kernel since every element passed" if probed else "plain signed-window multiplication"),
        }
        if direct:
            dm = macs_per_update(curve, group, endo=False)
            out["direct_path"] = {"value": direct[0], "unit": "G1 point-updates/s (this rank)", "kernel_ms": direct[1],
                                  "alu_frac": (dm * n / (direct[1] / 1e3) / MAC_PEAK) if dm else None, "macs_per_update": dm,
                                  "batch_exp_mode": "Direct (plain signed windows, no probe; what Auto falls back to when the probe fails)"}
        if endo:
            em = macs_per_update(curve, group, endo=True)
            out["endomorphism_path"] = {"value": endo[0], "unit": "G1 point-updates/s (this rank)", "kernel_ms": endo[1],
                                        "alu_frac": (em * n / (endo[1] / 1e3) / MAC_PEAK) if em else None, "macs_per_update": em,
                                        "batch_exp_mode": "Auto | PTAU_BATCH_EXP_INPUT_IN_SUBGROUP (GLV; same bytes on subgroup points)"}
        # the other groups of the config, one launch size each (kernel-only, same entry point)
        others = {}
        for (c2, g2, lg2) in cfg["groups"][1:] if world == 1 else []:
            n2 = 1 << lg2
            dc, dr, run2, u2, cs2 = make_vector(c2, g2, n2, 0)
            run2(dc, 0, dr, 1, tau)
            run2(dc, 0, dr, 1, tau)
            t2 = pkg.phase1_cli.last_timing(dev)
            p2, _ = pmc_for(KERNEL_TAG[(c2, g2)], args.config)
            # BLS12-377 G2 under default flags: Auto probes the vector and, every element being a subgroup element here, the kernel
            # that runs (and is timed) is the GLS one
            probed_endo = t2["subgroup_ms"] > 0
            m2 = macs_per_update(c2, g2, endo=probed_endo)
            entry = {"updates_per_s": n2 / ((t2["scalar_mul_ms"] + t2["normalize_ms"] + t2["subgroup_ms"]) / 1e3), "kernel_ms": t2["scalar_mul_ms"],
                     "normalize_kernel_ms": t2["normalize_ms"], "membership_probe_ms": t2["subgroup_ms"], "points": n2,
                     "algorithmic_bytes_per_update": 3 * cs2,
                     "hbm_frac": 3 * cs2 * n2 / (t2["scalar_mul_ms"] / 1e3) / 1e9 / HBM_PEAK_GBS,
                     "alu_frac": (m2 * n2 / (t2["scalar_mul_ms"] / 1e3) / MAC_PEAK) if m2 else None, "macs_per_update": m2,
                     "path": "endomorphism (GLV / GLS) kernel after the batched membership probe (BatchExpMode::Auto)" if probed_endo else "plain windows",
                     "traffic_ratio": (p2["hbm_bytes_per_launch"] / (3 * cs2 * n2)) if p2 else None}
            if (not c2.startswith("mnt") or g2 == 2) and not args.exact_only:      # MNT: GLS on the twists (G2) only
                run2(dc, 0, dr, 1, tau, ENDO)
                entry["endomorphism_updates_per_s"] = n2 / (pkg.phase1_cli.last_timing(dev)["kernel_ms"] / 1e3)
                entry["endomorphism_kernel_ms"] = pkg.phase1_cli.last_timing(dev)["scalar_mul_ms"]
                entry["endomorphism_alu_frac"] = macs_per_update(c2, g2, endo=True) * n2 / (entry["endomorphism_kernel_ms"] / 1e3) / MAC_PEAK
            if not args.exact_only:
                run2(dc, 0, dr, 1, tau, DIRECT)
                entry["direct_updates_per_s"] = n2 / (pkg.phase1_cli.last_timing(dev)["kernel_ms"] / 1e3)
                entry["direct_kernel_ms"] = pkg.phase1_cli.last_timing(dev)["scalar_mul_ms"]
                entry["direct_alu_frac"] = macs_per_update(c2, g2, endo=False) * n2 / (entry["direct_kernel_ms"] / 1e3) / MAC_PEAK
            others["%s_G%d" % (c2, g2)] = entry
            del dc, dr
        if others:
            out["other_groups"] = others
        if world == 1 and not args.no_extras:
            out["extras"] = whole_chunk_extras(pkg, curve, cfg["power"] if args.log2_points is None else lg,
                                               cfg["log2_chunk"] if args.log2_points is None else lg, dev)
            # the denominator of "x host-CPU chunk-contribution throughput": the same whole-chunk work on this box's host cores
            cb = cpu_chunk_baseline(curve, 14 if curve == "bls12_377" else (12 if curve == "bw6" else 10))
            nidx = 1 << (cfg["log2_chunk"] if args.log2_points is None else lg)
            cb["projected_chunk_seconds"] = nidx / cb["value"]
            ex = out["extras"]
            for k in ("contribute_chunk", "contribute_chunk_exact", "contribute_chunk_direct"):
                if k in ex:
                    ex[k]["vs_cpu_all_cores"] = cb["projected_chunk_seconds"] / ex[k]["seconds"]
                    ex[k]["vs_cpu_all_cores_device_time"] = cb["projected_chunk_seconds"] / (ex[k]["device_ms"] / 1e3)
            out["cpu_baseline_whole_chunk"] = cb
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def sharded_ratios_line(args, pkg, dev, rank, world, dist, f, prm, lg, barrier):
    """The one exchange step of the path, executed over the job's process group: the whole-transcript ratio
    check of verify_transcript (phase1_cli::transform_ratios, control.rs:587-591) with every vector's power_pairs reduction
    sliced over the ranks and ONE all_gather of the partial pairs per vector — RCCL over xGMI when the backend is nccl.
    Rank 0 builds the combined accumulator of the round (chunk 0 + the tauG1-only chunk 1 -> combine, untimed) in a
    directory every rank of the node can map; all ranks then run the sharded check on it."""
    import mmap
    import shutil
    n = 1 << lg
    shared = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp",
                          "ptau_bench_combined_%s" % os.environ.get("MASTER_PORT", str(os.getpid())))
    full = pkg.Phase1Parameters.new_full("bls12_377", 0, lg, n)
    try:
        if rank == 0:
            os.makedirs(shared, exist_ok=True)
            prm1 = pkg.Phase1Parameters.new_chunk("bls12_377", 0, 1, n, 0, lg, n)
            pkg.new_challenge(f("d0"), f("d0.hash"), prm1)
            pkg.phase1_cli.contribute_seeded(f("d0"), f("d0.h2"), f("s0"), f("s0.hash"), pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto,
                                             prm1, b"bench verify seed".ljust(64, b"."), device=dev)
            open(os.path.join(shared, "list"), "w").write(f("r0") + "\n" + f("s0") + "\n")
            pkg.phase1_cli.combine_file(os.path.join(shared, "list"), os.path.join(shared, "combined"), prm, device=dev)
        barrier()
        with open(os.path.join(shared, "combined"), "rb") as fh:
            mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
            try:
                pkg.phase1_cli.transform_ratios_sharded(mm, full, dist=dist, device=dev)      # warm-up (workspace growth)
                barrier()
                t0 = time.perf_counter()
                pkg.phase1_cli.transform_ratios_sharded(mm, full, dist=dist, device=dev)
                barrier()
                dt = time.perf_counter() - t0
            finally:
                mm.close()
        return {"seconds": dt, "ranks": world, "backend": (dist.get_backend() if dist is not None else "none"),
                "combined_bytes": full.accumulator_size,
                "what": "transform_ratios on the 2^%d combined accumulator: tauG1 (2^%d - 1), tauG2, alphaG1, betaG1 sliced over the "
                        "ranks (one rho stream keyed by the global index), one all_gather of the partial pairs per vector, 5 "
                        "pairing checks" % (lg, lg + 1)}
    finally:
        barrier()
        if rank == 0:
            shutil.rmtree(shared, ignore_errors=True)


def bench_lanes(args, pkg, dev, rank, world, dist, backend, barrier, verify):
    """--lanes L (with --config 2: contribute, --config 5: verify): WHOLE chunks through the file-level entry points with L
    in-process lanes per GPU — the operator's --max-in-process-lane (contribute.rs:120-121, 467-492): L threads of ONE process per
    GPU, each a full phase1_cli::contribute / transform_pok_and_correctness on its own file set (2^20 BLS12-377 chunk 0: 503 MB
    challenge, 252 MB response).  A step = every lane of this rank completes one chunk.  This is the line that shows what a
    node delivers in chunks/s: the device section of a call is 90-180 ms, the rest is the host's BLAKE2b chain, so the rate is
    bound by host threads (about 3 busy threads per lane) long before the GPUs are — `host_threads`, `host_threads_needed`
    and `device_busy_frac_per_rank` say where a given node stands."""
    import shutil
    import threading
    import torch
    lanes = args.lanes
    lg = args.log2_points or 20
    n = 1 << lg
    prm = pkg.Phase1Parameters.new_chunk("bls12_377", 0, 0, n, 0, lg, n)
    out_sz = prm.accumulator_size if verify else prm.contribution_size
    d, base = pick_workdir(2 * prm.accumulator_size + prm.contribution_size + lanes * out_sz, "ptau_bench_l%d_" % rank)
    f = lambda s: os.path.join(d, s)
    C, seed0, seed1 = pkg.CheckForCorrectness, b"bench lanes seed 0".ljust(64, b"."), b"bench lanes seed 1".ljust(64, b".")
    try:
        pkg.new_challenge(f("c0"), f("c0.hash"), prm)
        pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.h2"), f("r0"), f("r0.hash"), C.No, pkg.BatchExpMode.Auto, prm, seed0, device=dev)
        if not verify:      # contributions extend a VERIFIED challenge (genuine elements, as in a ceremony)
            pkg.transform_pok_and_correctness(f("c0"), f("v.c.hash"), C.No, f("r0"), f("v.r.hash"), C.OnlyNonZero, f("c1"), f("c1.hash"),
                                              pkg.SubgroupCheckMode.Auto, True, prm, device=dev)
        dev_ms = [0.0] * lanes
        errors = []

        def lane(k):
            try:
                if verify:
                    if os.path.exists(f("L%d.c1" % k)):
                        os.remove(f("L%d.c1" % k))
                    pkg.transform_pok_and_correctness(f("c0"), f("L%d.c.hash" % k), C.No, f("r0"), f("L%d.r.hash" % k), C.OnlyNonZero,
                                                      f("L%d.c1" % k), f("L%d.c1.hash" % k), pkg.SubgroupCheckMode.Auto, True, prm, device=dev)
                else:
                    if os.path.exists(f("L%d.r1" % k)):
                        os.remove(f("L%d.r1" % k))
                    pkg.phase1_cli.contribute_seeded(f("c1"), f("L%d.h" % k), f("L%d.r1" % k), f("L%d.r1.hash" % k), C.No,
                                                     pkg.BatchExpMode.Auto, prm, seed1, device=dev)
                dev_ms[k] += pkg.phase1_cli.last_timing(dev)["kernel_ms"]      # this lane's own call (per-thread record)
            except Exception as e:   # noqa: BLE001 - reported with the lane
                errors.append("lane %d: %r" % (k, e))

        def step():
            th = [threading.Thread(target=lane, args=(k,)) for k in range(lanes)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            if errors:
                raise SystemExit("bench.py --lanes: " + "; ".join(errors))
        for _ in range(args.warmup):
            step()
        for k in range(lanes):
            dev_ms[k] = 0.0
        barrier()
        Lb = pkg.load_library()
        busy0 = Lb.ptau_device_busy_ms(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        busy = [(Lb.ptau_device_busy_ms(dev) - busy0) / 1e3 / dt]
        if dist is not None:
            allb = [None] * world
            dist.all_gather_object(allb, busy[0])
            busy = allb
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        # every lane's output is the single-lane output (hash chain header checked here; tests/test_gpu_tail.py compares whole files)
        for k in range(lanes):
            if verify:
                assert open(f("L%d.c1" % k), "rb").read(64) == open(f("L%d.r.hash" % k), "rb").read()
            else:
                assert open(f("L%d.r1" % k), "rb").read(64) == open(f("L%d.h" % k), "rb").read()
        if rank == 0:
            chunks = world * lanes * args.steps
            elems = prm.g1_chunk_size + 3 * prm.other_chunk_size + 1
            ncpu = host_cpu_share()
            per_lane_threads = 3.0      # a lane's launching thread + two BLAKE2b / page-population workers busy at a time
            print(json.dumps({
                "metric": "chunks verified/sec" if verify else "chunks contributed/sec",
                "value": chunks / dt, "unit": "chunks %s/s" % ("verified" if verify else "contributed"),
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                "config": {"workload": "whole 2^%d BLS12-377 chunks (chunk 0: %d elements, challenge %d B, response %d B) through the "
                                       "file-level %s, %d in-process lanes per GPU x %d GPU(s), files on %s; BLAKE2b chain, key "
                                       "generation / proofs of knowledge, PCIe inside the step"
                                       % (lg, elems, prm.accumulator_size, prm.contribution_size,
                                          "phase1_cli::transform_pok_and_correctness (ratio check on)" if verify else
                                          "phase1_cli::contribute (BatchExpMode::Auto, CheckForCorrectness::No: the operator's default flags)",
                                          lanes, world, base),
                           "baseline_config": 5 if verify else 2, "lanes_per_gpu": lanes, "chunks_per_step": world * lanes},
                "chunks_per_s": chunks / dt, "chunks_per_s_per_gpu": lanes * args.steps / dt,
                "g1_equiv_updates_per_s": chunks * elems / dt,
                "device_ms_per_chunk_rank0": sum(dev_ms) / (lanes * args.steps),
                "device_busy_frac_per_rank": busy,
                "host_threads": ncpu, "host_threads_needed": world * lanes * per_lane_threads,
                "host_bound": world * lanes * per_lane_threads > ncpu,
                "note": "the rate of whole chunks is bound by the host (one sequential BLAKE2b stream per file, ~1.3 GB/s per core) "
                        "long before the GPU: device_busy_frac says how much of the wall time the GPU had kernels running; "
                        "host_threads_needed = lanes x ranks x ~3 busy threads — on a node with fewer host threads more GPUs do not "
                        "add chunks/s (INTEGRATION.md §4)"}))
    finally:
        barrier()
        shutil.rmtree(d, ignore_errors=True)
    if dist is not None:
        dist.destroy_process_group()


def bench_verify(args, pkg, dev, rank, world, dist, backend, barrier):
    """--config 5: verification of one 2^20 BLS12-377 chunk per GPU (BASELINE configs[4], second half of the metric).  A step =
    phase1_cli::transform_pok_and_correctness on chunk 0 (files on tmpfs, ratio check on): the boundary of this path is the
    file pair, so `value` includes PCIe and the BLAKE2b chain; the device-resident share is reported beside it."""
    import shutil
    import tempfile
    import torch
    lg = args.log2_points or 20
    n = 1 << lg
    prm = pkg.Phase1Parameters.new_chunk("bls12_377", 0, 0, n, 0, lg, n)
    if args.lanes:
        return bench_lanes(args, pkg, dev, rank, world, dist, backend, barrier, verify=True)
    # this rank's files (challenge, response, new challenge, the tauG1-only chunk of the tail) + rank 0's combined accumulator
    full_sz = pkg.Phase1Parameters.new_full("bls12_377", 0, lg, n).accumulator_size
    d, base = pick_workdir(3 * prm.accumulator_size + 2 * prm.contribution_size + (full_sz if rank == 0 else 0), "ptau_bench_v%d_" % rank)
    f = lambda s: os.path.join(d, s)
    try:
        pkg.new_challenge(f("c0"), f("c0.hash"), prm)
        pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.h2"), f("r0"), f("r0.hash"), pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto,
                                         prm, b"bench verify seed".ljust(64, b"."), device=dev)

        def step(k):
            if os.path.exists(f("c1")):
                os.remove(f("c1"))
            pkg.transform_pok_and_correctness(f("c0"), f("v.c.hash"), pkg.CheckForCorrectness.No, f("r0"), f("v.r.hash"),
                                              pkg.CheckForCorrectness.OnlyNonZero, f("c1"), f("c1.hash"), pkg.SubgroupCheckMode.Auto,
                                              True, prm, device=dev)
        for k in range(args.warmup):
            step(k)
        barrier()
        Lb = pkg.load_library()
        busy0 = Lb.ptau_device_busy_ms(dev)
        t0 = time.perf_counter()
        dev_ms = sub_ms = dec_ms = msm_ms = tail_ms = 0.0
        for k in range(args.steps):
            step(k)
            tm = pkg.phase1_cli.last_timing(dev)
            dev_ms += tm["kernel_ms"]
            sub_ms += tm["subgroup_ms"]
            dec_ms += tm["decompress_ms"]
            msm_ms += tm["msm_ms"]
            tail_ms += tm["msm_tail_ms"]
        barrier()
        dt = time.perf_counter() - t0
        busy = [(Lb.ptau_device_busy_ms(dev) - busy0) / 1e3 / dt]
        if dist is not None:
            allb = [None] * world
            dist.all_gather_object(allb, busy[0])
            busy = allb
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        sharded = sharded_ratios_line(args, pkg, dev, rank, world, dist, f, prm, lg, barrier)
        if rank == 0:
            assert open(f("c1"), "rb").read(64) == open(f("v.r.hash"), "rb").read()
            elems = 4 * n + 1
            dec_s = dec_ms / args.steps / 1e3
            algo = 3 * n * (48 + 96) + (n + 1) * (96 + 192)      # decompression: compressed element in, uncompressed element out
            dmacs = decompress_macs_per_chunk(n)
            p1, src1 = pmc_for("BlsG1", 5, "decompress_kernel")
            p2, _ = pmc_for("BlsG2", 5, "decompress_kernel")
            # PMC traffic of the four decompression launches of one chunk (3 G1 vectors + the G2 vector; the lone beta_g2 launch is noise)
            dtraffic = (3 * p1["hbm_bytes_per_launch"] + p2["hbm_bytes_per_launch"]) if (p1 and p2 and lg == 20) else None
            print(json.dumps({
                "metric": "chunks verified/sec", "value": world * args.steps / dt, "unit": "chunks verified/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                "config": {"workload": "verify_transcript step on a 2^%d BLS12-377 ceremony: transform_pok_and_correctness of chunk 0 "
                                       "(%d elements: challenge %d B, response %d B), ratio check + proofs of knowledge, files on %s; "
                                       "PCIe and the BLAKE2b chain are inside the step (the path's boundary is the file pair)"
                                       % (lg, elems, prm.accumulator_size, prm.contribution_size, base or "tmp"),
                           "baseline_config": 5, "elements_per_step_per_gpu": elems},
                "device_resident": {"kernel_ms_per_chunk": dev_ms / args.steps, "chunks_per_s": 1e3 / (dev_ms / args.steps),
                                    "decompress_ms": dec_ms / args.steps, "subgroup_ms": sub_ms / args.steps, "msm_ms": msm_ms / args.steps,
                                    "msm_tail_latency_ms_hidden_on_second_stream": tail_ms / args.steps,
                                    "note": "SubgroupCheckMode::Auto on 2^20 elements = the batched membership test through the buckets of "
                                            "the ratio check's MSM (inside msm_ms); PTAU_BATCHED_SUBGROUP=0 runs the exact per-element test"},
                "device_busy_frac_per_rank": busy,
                "roofline": {"bound": "hbm", "kernel": "decompress_kernel (G1 x3 vectors + G2)", "kernel_ms": dec_ms / args.steps,
                             "achieved": algo / dec_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo / dec_s / 1e9 / HBM_PEAK_GBS,
                             "traffic": dtraffic, "traffic_source": src1, "traffic_ratio": (dtraffic / algo) if dtraffic else None,
                             "algorithmic_bytes_per_chunk": algo, "macs_per_chunk": dmacs, "alu_peak_mac_per_s": MAC_PEAK,
                             "alu_frac": dmacs / dec_s / MAC_PEAK,
                             "note": "a square root is ~470 field products per coordinate (330-bit exponentiation + Pohlig-Hellman in the "
                                     "2^46 torsion on one squaring chain): integer-ALU-bound; alu_frac = modelled 32x32->64 MACs (bench.py "
                                     "sqrt_macs / decompress_macs_per_chunk) over the measured v_mad_u64_u32 issue peak; traffic = PMC bytes of "
                                     "the chunk's four decompression launches (the 80 KB torsion table is gathered per lane), DESIGN.md §4.4"},
                "msm": {"kernel": "msm_accumulate_kernel (3 G1 vectors + G2, two sums each) inside msm_ms", "msm_ms": msm_ms / args.steps,
                        "accumulate_macs_per_chunk": msm_accumulate_macs_per_chunk(n),
                        "alu_frac_lower_bound": msm_accumulate_macs_per_chunk(n) / (msm_ms / args.steps / 1e3) / MAC_PEAK,
                        "note": "msm_ms also holds msm_prepare, the counting sort, msm_combine and what of the tails was not hidden: the "
                                "accumulate launches alone are 29 of its 44 ms (profiles/r04_verify_timeline.txt), i.e. alu_frac ~0.59 — "
                                "random gathers of 116-byte points, one exposed latency per addition"},
                "sharded_transform_ratios": sharded,
                "cpu_baseline": cpu_verify_baseline(lg) if world == 1 else None}))
    finally:
        shutil.rmtree(d, ignore_errors=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
