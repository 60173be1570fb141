"""Chunk farm: independent chunks of one ceremony round spread over the ranks of a node.

Chunks share (tau, alpha, beta) and differ only by their global exponent offset
chunk_index*chunk_size (/root/reference/src/bin/contribute.rs:783-789: the same seed -> the same key in
every chunk), so the contribute path shards with NO data-path collective: rank r runs its chunks on its
own GPU and only the 64-byte response hashes are exchanged at the end (all_gather over RCCL on GPUs,
gloo in the CPU tests).  Second-half chunks (start >= 2^power) carry only tauG1 (about 1/5 of the work
of a full chunk), so assignment is longest-processing-time-first rather than round-robin.
"""
import os
from typing import Callable, Dict, List, Sequence, Tuple


# Cost of one G2 update in G1 updates, measured on MI355X this round (profiles/r02_*: kernel time per element of the two
# groups' tau-update, exact path; the endomorphism path of BLS12-377 gives 4.3): Fq2 arithmetic on BLS12-377 and MNT4-753,
# Fq3 on MNT6-753, the same base field on BW6-761.
G2_COST = {"bls12_377": 5.2, "bw6": 1.0, "mnt4_753": 4.4, "mnt6_753": 9.7}


def chunk_weight(g1_chunk_size: int, other_chunk_size: int, curve: str = "bls12_377", g2_cost: float = None) -> float:
    """Relative cost of one chunk: tauG1 + (tauG2 + alphaG1 + betaG1) elements; a G2 update costs g2_cost G1 updates
    (per-curve measured ratio unless given)."""
    if g2_cost is None:
        g2_cost = G2_COST[curve]
    return g1_chunk_size + other_chunk_size * (2.0 + g2_cost)


def assign_chunks(weights: Sequence[float], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first assignment; deterministic, identical on every rank."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    loads = [0.0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += weights[i]
    return [sorted(x) for x in out]


def run_farm(n_chunks: int, weights: Sequence[float], process_chunk: Callable[[int], bytes], dist=None) -> Dict[int, bytes]:
    """Every rank processes its share with `process_chunk(chunk_index) -> response_hash` and all ranks
    receive the full {chunk_index: response_hash} map.  `dist` is torch.distributed (initialised) or None."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    mine = assign_chunks(weights, world)[rank]
    local: List[Tuple[int, bytes]] = [(i, process_chunk(i)) for i in mine]
    if dist is None:
        gathered = [local]
    else:
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
    result: Dict[int, bytes] = {}
    for part in gathered:
        for i, h in part:
            result[i] = h
    if sorted(result) != list(range(n_chunks)):      # not an assert: must hold under python -O as well
        raise RuntimeError("a chunk was dropped or duplicated by the farm: got %r of %d" % (sorted(result), n_chunks))
    return result


# ---- final transcript ratio check, sharded (SURVEY §8(e) collective C1) ---------------------------------------
# Phase1::aggregate_verification (phase1_cli::transform_ratios; /root/reference/src/bin/control.rs:587-591,
# src/bin/verify_transcript.rs:646-653, 812-822) reduces each full-length vector to ONE pair
# (sum rho_i v_i, sum rho_i v_{i+1}) and tests it with two pairings.  The reduction is linear, so each rank reduces
# a contiguous slice of pair indices with the SAME coefficient stream (the generator is keyed by a shared 16-byte
# seed and the GLOBAL index, ptau_power_pairs' index_base); the per-rank partial pairs are exchanged with ONE
# all_gather per vector (2 points x world_size, RCCL over xGMI on GPUs / gloo in the CPU tests), summed, and the
# O(1) pairing tail runs once.  Slices overlap by one element: pair (i, i+1) straddles the boundary.

def slice_bounds(n_pairs: int, world: int, rank: int) -> Tuple[int, int]:
    """Pair-index range [lo, hi) of `rank`: contiguous, balanced to within one pair, empty when n_pairs < world."""
    base, extra = divmod(n_pairs, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def accumulator_vectors(point_sizes: Dict[int, int], powers_g1_length: int, powers_length: int):
    """(name, group, byte offset, element count) of the four power vectors in a Full-mode uncompressed accumulator,
    plus the offset of betaG2.  point_sizes = {1: g1 uncompressed bytes, 2: g2 uncompressed bytes}."""
    u1, u2 = point_sizes[1], point_sizes[2]
    o = 64
    out = []
    for name, grp, cnt in (("tau_g1", 1, powers_g1_length), ("tau_g2", 2, powers_length), ("alpha_g1", 1, powers_length),
                           ("beta_g1", 1, powers_length)):
        out.append((name, grp, o, cnt))
        o += cnt * point_sizes[grp]
    return out, o


def sharded_transform_ratios(combined, point_sizes: Dict[int, int], powers_g1_length: int, powers_length: int,
                             power_pairs: Callable, sum_points: Callable, check_same_ratio: Callable, dist=None,
                             seed: bytes = None, device: int = 0) -> bool:
    """transform_ratios with the four vector reductions split over the ranks of `dist`.

    power_pairs(group, data, n, seed, index_base) -> (a, b)   reduces n consecutive points (n - 1 pairs) of one slice;
    sum_points(group, [points])                 -> point     adds serialized points;
    check_same_ratio((a, b), (c, d))            -> bool      the pairing test, G1 pair vs G2 pair.
    On GPUs these are phase1_cli.power_pairs / dense_multiexp with unit scalars / check_same_ratio; the CPU tests
    inject oracle stand-ins.  Every rank returns the same verdict.
    `combined` is any buffer (bytes, or a memory-mapped file): a rank copies only ITS slice of each vector out of it.
    `device`: this rank's GPU ordinal (the tensors of the RCCL collectives live there)."""
    import torch
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    dev = torch.device("cuda", device) if (dist is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    combined = memoryview(combined)
    if seed is None:
        seed = os.urandom(16)
    if dist is not None:   # one coefficient stream for the whole vector: rank 0's seed
        t = torch.frombuffer(bytearray(seed), dtype=torch.uint8).to(dev)
        dist.broadcast(t, src=0)
        seed = bytes(t.cpu().numpy().tobytes())
    vectors, beta_g2_off = accumulator_vectors(point_sizes, powers_g1_length, powers_length)
    reduced = {}
    for name, grp, off, cnt in vectors:
        usz = point_sizes[grp]
        lo, hi = slice_bounds(cnt - 1, world, rank)
        if hi > lo:      # points lo .. hi inclusive
            a, b = power_pairs(grp, bytes(combined[off + lo * usz: off + (hi + 1) * usz]), hi - lo + 1, seed, lo)
            part = a + b + b"\x01"
        else:
            part = bytes(2 * usz) + b"\x00"      # no pairs on this rank
        if dist is not None:
            mine = torch.frombuffer(bytearray(part), dtype=torch.uint8).to(dev)
            gathered = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
            parts = [bytes(g.cpu().numpy().tobytes()) for g in gathered]
        else:
            parts = [part]
        parts = [p for p in parts if p[-1]]
        reduced[name] = (sum_points(grp, [p[:usz] for p in parts]), sum_points(grp, [p[usz:2 * usz] for p in parts]))
    u1, u2 = point_sizes[1], point_sizes[2]
    v = {name: off for name, _, off, _ in vectors}
    cut = lambda a, b: bytes(combined[a:b])
    tau_g1_01 = (cut(v["tau_g1"], v["tau_g1"] + u1), cut(v["tau_g1"] + u1, v["tau_g1"] + 2 * u1))
    tau_g2_01 = (cut(v["tau_g2"], v["tau_g2"] + u2), cut(v["tau_g2"] + u2, v["tau_g2"] + 2 * u2))
    ok = True
    for name in ("tau_g1", "alpha_g1", "beta_g1"):
        ok = ok and check_same_ratio(reduced[name], tau_g2_01)
    ok = ok and check_same_ratio(tau_g1_01, reduced["tau_g2"])
    beta_g1_0 = cut(v["beta_g1"], v["beta_g1"] + u1)
    beta_g2 = cut(beta_g2_off, beta_g2_off + u2)
    ok = ok and check_same_ratio((tau_g1_01[0], beta_g1_0), (tau_g2_01[0], beta_g2))
    return bool(ok)
