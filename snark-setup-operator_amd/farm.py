"""Chunk farm: independent chunks of one ceremony round spread over the ranks of a node.

Chunks share (tau, alpha, beta) and differ only by their global exponent offset
chunk_index*chunk_size (/root/reference/src/bin/contribute.rs:783-789: the same seed -> the same key in
every chunk), so the contribute path shards with NO data-path collective: rank r runs its chunks on its
own GPU and only the 64-byte response hashes are exchanged at the end (all_gather over RCCL on GPUs,
gloo in the CPU tests).  Second-half chunks (start >= 2^power) carry only tauG1 (about 1/5 of the work
of a full chunk), so assignment is longest-processing-time-first rather than round-robin.
"""
from typing import Callable, Dict, List, Sequence, Tuple


def chunk_weight(g1_chunk_size: int, other_chunk_size: int, g2_cost: float = 3.0) -> float:
    """Relative cost of one chunk: tauG1 + (tauG2 + alphaG1 + betaG1) elements; a G2 update costs about
    g2_cost G1 updates (Fq2 arithmetic)."""
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
    assert sorted(result) == list(range(n_chunks)), "a chunk was dropped or duplicated by the farm"
    return result
