#!/usr/bin/env python3
"""bench.py — G1 point-updates/s of the chunk tau-update (BASELINE.json metric), one rank per GPU.

Step = one pass of the hot path over one chunk vector: the tau-update of the 2^20-point tauG1
vector of a BLS12-377 power-20 chunk (BASELINE configs[1]), device-resident in and out
(uncompressed challenge points -> compressed response points).  N > 1: every rank updates its own
chunk (independent chunks, no data-path collective) -> weak scaling.
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
ALGO_BYTES_PER_UPDATE = 96 + 48      # uncompressed read + compressed write per BLS12-377 G1 update
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


def pmc_traffic():
    """HBM bytes per scalar_mul_kernel launch from the committed rocprofv3 PMC summary (profiles/)."""
    try:
        import glob
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))[-1]
        d = json.load(open(f))
        k = d["kernels"]
        for name in ("window_scalar_mul_kernel<BlsG1>", "window_scalar_mul_kernel", "scalar_mul_kernel"):
            if name in k and "hbm_bytes_per_launch" in k[name]:
                return k[name]["hbm_bytes_per_launch"], os.path.basename(f)
        return None, None
    except Exception:
        return None, None


def cpu_baseline(sample_points, chal_bytes, start, tau):
    """The C oracle ("port") timed on this host's cores over a bounded sample of the same workload."""
    from oracle import cref
    cores = usable_cores()
    cref.lib()
    t0 = time.time()
    out = cref.apply_powers("bls12_377", 1, chal_bytes, sample_points, start, tau, None, False, True, nthreads=cores)
    dt = time.time() - t0
    return out, {"value": sample_points / dt, "unit": "G1 point-updates/s", "cores": cores, "kind": "port",
                 "sample": "%d of the 2^20 tauG1 points of the same chunk, C oracle (oracle/c), %d threads, %.1f s wall"
                           % (sample_points, cores, dt)}


def whole_chunk_extras(pkg, log2n, dev):
    """Second half of the BASELINE metric ("chunks verified/sec") and the whole-chunk contribution, on chunk 0 of a
    BLS12-377 power-`log2n` ceremony with chunk_size 2^log2n (3*2^n G1 + 2^n+1 G2 elements), through the FILE-level
    entry points the operator calls (phase1_cli::contribute / transform_pok_and_correctness: mmap'd files on tmpfs,
    BLAKE2b of every file, `.hash` side files, PCIe both ways, pairing tail).  Untimed set-up: new_challenge -> one
    contribution; timed: verify (ratio_check on / off) and a second contribution of the verified new challenge."""
    import shutil
    import tempfile
    n = 1 << log2n
    prm = pkg.Phase1Parameters.new_chunk("bls12_377", 0, 0, n, 0, log2n, n)
    seed0, seed1 = b"bench seed 0".ljust(64, b"."), b"bench seed 1".ljust(64, b".")
    # the coordinator hands a contributor a challenge that came out of a verifier's subgroup checks: the operator-side
    # integration asserts that and gets the endomorphism path; plain Auto is exact on any curve point
    verified = int(pkg.BatchExpMode.Auto) | pkg.phase1_cli.INPUT_IN_SUBGROUP
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="ptau_bench_", dir=base)
    f = lambda name: os.path.join(d, name)
    out = {"workload": "BLS12-377 power %d chunk 0, chunk_size 2^%d: %d G1 + %d G2 elements; challenge %d B, response %d B; "
                       "file-level entry points, files on %s" % (log2n, log2n, 3 * n, n + 1, prm.accumulator_size,
                                                                  prm.contribution_size, base or "the default tmp dir")}
    try:
        pkg.new_challenge(f("c0"), f("c0.hash"), prm)
        pkg.phase1_cli.contribute_seeded(f("c0"), f("c0.hash2"), f("r0"), f("r0.hash"), pkg.CheckForCorrectness.No, pkg.BatchExpMode.Auto,
                                         prm, seed0, device=dev)
        t0 = time.perf_counter()
        pkg.transform_pok_and_correctness(f("c0"), f("v.c.hash"), pkg.CheckForCorrectness.No, f("r0"), f("v.r.hash"),
                                          pkg.CheckForCorrectness.OnlyNonZero, f("c1"), f("c1.hash"), pkg.SubgroupCheckMode.Auto,
                                          True, prm, device=dev)
        dt = time.perf_counter() - t0
        tm = pkg.phase1_cli.last_timing(dev)
        out["verify_chunk"] = {"seconds": dt, "chunks_verified_per_s": 1.0 / dt, "elements_per_s": (4 * n + 1) / dt,
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
        for tag, mode, what in (("contribute_chunk", verified, "Auto | INPUT_IN_SUBGROUP (challenge produced by the verification above: endomorphism path)"),
                                ("contribute_chunk_exact", pkg.BatchExpMode.Auto, "Auto, CheckForCorrectness::No (exact on any curve point: plain windows)")):
            if os.path.exists(f("r1")):
                os.remove(f("r1"))
            t0 = time.perf_counter()
            pkg.phase1_cli.contribute_seeded(f("c1"), f("c1.hash2"), f("r1"), f("r1.hash"), pkg.CheckForCorrectness.No, mode, prm, seed1, device=dev)
            dt3 = time.perf_counter() - t0
            tm = pkg.phase1_cli.last_timing(dev)
            out[tag] = {"seconds": dt3, "chunks_per_s": 1.0 / dt3, "g1_equiv_updates_per_s": (4 * n + 1) / dt3, "batch_exp_mode": what,
                        "includes": "phase1_cli::contribute on files from a seed: key generation, BLAKE2b(challenge) overlapped, 5 vectors "
                                    "updated (PCIe both ways), public key + BLAKE2b(response) + msync",
                        "device_ms": tm["kernel_ms"], "h2d_ms": tm["h2d_ms"], "d2h_ms": tm["d2h_ms"]}
        assert open(f("c1"), "rb").read(64) == open(f("v.r.hash"), "rb").read()   # new challenge extends the response it verified
        # several in-process lanes on ONE GPU (the operator's --max-in-process-lane, contribute.rs:120-121,467-492): the
        # library serialises only the device section of a call, so one lane's BLAKE2b overlaps another lane's kernels
        import threading
        for lanes in (2, 3):
            def lane_verify(k):
                pkg.transform_pok_and_correctness(f("c0"), f("L%d.c.hash" % k), pkg.CheckForCorrectness.No, f("r0"), f("L%d.r.hash" % k),
                                                  pkg.CheckForCorrectness.OnlyNonZero, f("L%d.c1" % k), f("L%d.c1.hash" % k),
                                                  pkg.SubgroupCheckMode.Auto, True, prm, device=dev)
            def lane_contribute(k):
                pkg.phase1_cli.contribute_seeded(f("c1"), f("L%d.h" % k), f("L%d.r1" % k), f("L%d.r1.hash" % k), pkg.CheckForCorrectness.No,
                                                 verified, prm, seed1, device=dev)
            for name, fn, outs in (("verify", lane_verify, ("L%d.c1",)), ("contribute", lane_contribute, ("L%d.r1",))):
                th = [threading.Thread(target=fn, args=(k,)) for k in range(lanes)]
                t0 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                dtl = time.perf_counter() - t0
                for k in range(lanes):
                    for o in outs:
                        assert os.path.getsize(f(o % k)) > 0
                        os.remove(f(o % k))
                out.setdefault("lanes", {})["%s_%d_lanes" % (name, lanes)] = {"seconds": dtl, "chunks_per_s": lanes / dtl}
        # whole-transcript tail (BASELINE configs[4]: verify_transcript's combine + transform_ratios): the round's second
        # chunk (tauG1 only, 2^n - 1 elements) contributed with the same key, both responses combined, ratio checks run
        prm1 = pkg.Phase1Parameters.new_chunk("bls12_377", 0, 1, n, 0, log2n, n)
        full = pkg.Phase1Parameters.new_full("bls12_377", 0, log2n, n)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2-points", type=int, default=20)
    ap.add_argument("--cpu-sample", type=int, default=1 << 20, help="points the CPU oracle updates (parity check + cpu_baseline; ~13 s at 16 threads)")
    ap.add_argument("--no-extras", action="store_true", help="skip the whole-chunk contribute / verify measurements")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # rehearsal knobs for a one-GPU box (the driver's N > 1 runs use neither): PTAU_BENCH_BACKEND=gloo lets several ranks
    # share one device (RCCL refuses duplicate GPUs), PTAU_BENCH_DEVICE pins every rank to that device
    backend = os.environ.get("PTAU_BENCH_BACKEND", "nccl")
    if "PTAU_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["PTAU_BENCH_DEVICE"])
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
    n = 1 << args.log2_points
    nl = L.ptau_fr_limbs(0)
    eb = ctypes.create_string_buffer(256)

    ENDO = int(pkg.BatchExpMode.Auto) | pkg.phase1_cli.INPUT_IN_SUBGROUP

    def run(d_in, cin, d_out, cout, count, start, tau, mode=0):
        rc = L.ptau_apply_powers_dev(0, 1, ctypes.c_void_p(d_in.data_ptr()), cin, ctypes.c_void_p(d_out.data_ptr()), cout,
                                     count, start, pkg._lib.limbs(tau, nl), None, 0, mode, dev, eb, len(eb))
        if rc:
            raise RuntimeError("ptau_apply_powers_dev rc=%d: %s" % (rc, eb.value.decode()))

    # synthetic chunk (rank r owns chunk index r of a power-20 ceremony, chunk_size 2^20):
    # a genuine accumulator vector [s^(start+i) G1] made on the device from the generator (untimed).
    start = rank * n
    s, tau = det_scalar("bench/s"), det_scalar("bench/tau")
    gen = G1_GEN_X.to_bytes(48, "little") + G1_GEN_Y.to_bytes(48, "little")
    d_gen = torch.frombuffer(bytearray(gen * n), dtype=torch.uint8).cuda()
    d_chal = torch.empty(n * 96, dtype=torch.uint8, device="cuda")
    d_resp = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    run(d_gen, 0, d_chal, 0, n, start, s, ENDO)
    del d_gen

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run(d_chal, 0, d_resp, 1, n, start, tau)
    barrier()
    t0 = time.perf_counter()
    smul_ms = norm_ms = 0.0
    for _ in range(args.steps):
        run(d_chal, 0, d_resp, 1, n, start, tau)
        tm = pkg.phase1_cli.last_timing(dev)      # HIP events on the library's own stream
        smul_ms += tm["scalar_mul_ms"]
        norm_ms += tm["normalize_ms"]
    barrier()
    dt = time.perf_counter() - t0
    # the same step on the endomorphism path (GLV; membership asserted by the caller — what an operator that trusts its
    # coordinator's verifier runs): reported next to `value`, never as `value`
    run(d_chal, 0, d_resp, 1, n, start, tau, ENDO)
    endo_ms = 0.0
    te = time.perf_counter()
    for _ in range(args.steps):
        run(d_chal, 0, d_resp, 1, n, start, tau, ENDO)
        endo_ms += pkg.phase1_cli.last_timing(dev)["scalar_mul_ms"]
    torch.cuda.synchronize()
    dte = time.perf_counter() - te
    run(d_chal, 0, d_resp, 1, n, start, tau)      # d_resp again holds the exact path's output (parity sample below)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        value = world * n * args.steps / dt
        kernel_s = (smul_ms / args.steps) / 1e3
        achieved = ALGO_BYTES_PER_UPDATE * n / kernel_s / 1e9
        # parity check + CPU baseline on a bounded sample of the same chunk (the timed baseline only at N = 1: at
        # N > 1 the other ranks would idle behind rank 0's host cores, so only a small parity sample is checked)
        sample = min(args.cpu_sample if world == 1 else 4096, n)
        chal_host = d_chal[:sample * 96].cpu().numpy().tobytes()
        ref, base = cpu_baseline(sample, chal_host, start, tau)
        if world > 1:
            base = None
        got = d_resp[:sample * 48].cpu().numpy().tobytes()
        if got != ref:
            raise SystemExit("PARITY FAILURE: GPU response differs from the CPU oracle on the sampled points")
        traffic, traffic_src = pmc_traffic()
        name = ctypes.create_string_buffer(128)
        L.ptau_device_info(dev, name, 128, None, None)
        extras = None
        if world == 1 and not args.no_extras:
            extras = whole_chunk_extras(pkg, args.log2_points, dev)
        print(json.dumps({
            "metric": "G1 point-updates/sec per chunk (contribute)", "value": value, "unit": "G1 point-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "BLS12-377 phase-1 power 20, chunk_size 2^20: tauG1 vector of one chunk per GPU "
                                   "(2^%d points, uncompressed in -> compressed out, HBM-resident)" % args.log2_points,
                       "points_per_step_per_gpu": n, "device": name.value.decode(), "parity": "bit-exact vs CPU oracle on %d sampled points" % sample},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "window_scalar_mul_kernel<BLS12-377 G1>", "kernel_ms": smul_ms / args.steps,
                         "normalize_kernel_ms": norm_ms / args.steps,
                         "note": "integer-ALU-bound by construction (~0.6 M 32x32->64 MACs per update vs 144 B; 58 % of the measured v_mad_u64_u32 issue peak): see DESIGN.md"},
            "cpu_baseline": base,
            "value_path": "BatchExpMode::Auto with CheckForCorrectness::No (the operator's contribute default): plain signed-window "
                          "multiplication, exact for ANY curve point",
            "endomorphism_path": {"value": n * args.steps / dte, "unit": "G1 point-updates/s (this rank)", "kernel_ms": endo_ms / args.steps,
                                  "batch_exp_mode": "Auto | PTAU_BATCH_EXP_INPUT_IN_SUBGROUP (GLV; same bytes on subgroup points)"},
            "extras": extras,
        }))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
