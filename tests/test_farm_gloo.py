"""world_size-2 gloo test of the N>1 path (chunk farm) on CPU.  The per-chunk compute is injected:
here it is the C oracle (test infrastructure); on GPUs it is phase1_cli.contribute."""
import os
import subprocess
import sys

from conftest import ROOT, load_package

WORKER = r'''
import os, sys, json
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
from conftest import load_package
from oracle import cref
from oracle.pyref import phase1 as P
from oracle.pyref.curves import CURVES
pkg = load_package()
from snark_setup_operator_amd import farm
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
c = CURVES["bls12_377"]
power, cs = 3, 4                      # G1L = 15 -> 4 chunks: 2 full, 2 tauG1-only
prm = [P.Params(c, P.CHUNKED, i, cs, power, cs) for i in range(4)]
weights = [farm.chunk_weight(p.g1_chunk_size, p.other_chunk_size) for p in prm]
tau, alpha, beta = 0x1234567, 0x7654321, 0xabcdef
done = []
def process(i):
    done.append(i)
    ch = P.initialization(prm[i])
    out = cref.computation("bls12_377", 0, i, cs, power, ch, tau, alpha, beta)
    return cref.blake2b512(out)
res = farm.run_farm(4, weights, process, dist)
print(json.dumps({{"rank": dist.get_rank(), "done": done, "res": {{str(k): v.hex() for k, v in res.items()}}}}))
dist.destroy_process_group()
'''


def test_assignment_is_balanced_and_complete():
    pkg = load_package()
    from snark_setup_operator_amd import farm
    # BASELINE config #3 shape: 8 chunks, 4 full + 4 tauG1-only, 8 ranks -> one chunk each
    w = [farm.chunk_weight(1 << 18, 1 << 18)] * 4 + [farm.chunk_weight(1 << 18, 0)] * 4
    a = farm.assign_chunks(w, 8)
    assert sorted(i for r in a for i in r) == list(range(8)) and all(len(r) == 1 for r in a)
    # 2 ranks: each gets two heavy and two light chunks
    a = farm.assign_chunks(w, 2)
    assert [sum(w[i] for i in r) for r in a][0] == [sum(w[i] for i in r) for r in a][1]
    # more ranks than chunks
    a = farm.assign_chunks(w[:3], 4)
    assert sorted(i for r in a for i in r) == [0, 1, 2] and a[3] == []
    # single process path
    assert farm.run_farm(3, [1, 1, 1], lambda i: bytes([i])) == {0: b"\0", 1: b"\1", 2: b"\2"}


def test_farm_world_size_2_gloo(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-2000:]
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    # disjoint, complete sharding; identical result map on both ranks
    d0, d1 = set(outs[0]["done"]), set(outs[1]["done"])
    assert d0 | d1 == {0, 1, 2, 3} and not (d0 & d1) and d0 and d1
    assert outs[0]["res"] == outs[1]["res"] and len(outs[0]["res"]) == 4
    # each rank got one full and one tauG1-only chunk (balanced)
    for d in (d0, d1):
        assert len(d & {0, 1}) == 1 and len(d & {2, 3}) == 1


RATIO_WORKER = r'''
import os, sys, json, random
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
from conftest import load_package
from oracle.pyref import phase1 as P, serde
from oracle.pyref.curves import CURVES
from oracle.pyref.ec import Group
from oracle.pyref.pairing import Pairing
pkg = load_package()
from snark_setup_operator_amd import farm
rank, world = int(sys.argv[1]), int(sys.argv[2])
dist_mod = None
if world > 1:
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dist_mod = dist
c = CURVES["bls12_377"]
Pg = Pairing(c)
G = {{1: Pg.G1, 2: Pg.G2}}
power = 2
full = P.Params(c, P.FULL, 0, 4, power, 4)
acc = bytes(64) + P.computation(full, P.initialization(full), 0x1234567, 0x7654321, 0xabcdef, False, False)
sizes = {{1: Pg.G1.uncomp_size, 2: Pg.G2.uncomp_size}}
calls = []
def power_pairs(grp, data, n, seed, index_base):
    calls.append((grp, n, index_base))
    pts = serde.read_batch(G[grp], data, False)
    a = b = None
    for i in range(n - 1):
        rho = pkg.phase1_cli.power_pairs_coefficient(seed, index_base + i, c.r.bit_length())
        a = G[grp].add(a, G[grp].mul(pts[i], rho))
        b = G[grp].add(b, G[grp].mul(pts[i + 1], rho))
    return serde.write_point(G[grp], a, False), serde.write_point(G[grp], b, False)
def sum_points(grp, pts):
    s = None
    for p in pts:
        s = G[grp].add(s, serde.read_point(G[grp], p, False))
    return serde.write_point(G[grp], s, False)
def check(g1p, g2p):
    return Pg.check_same_ratio(tuple(serde.read_point(Pg.G1, x, False) for x in g1p), tuple(serde.read_point(Pg.G2, x, False) for x in g2p))
args = (sizes, full.powers_g1_length, full.powers_length, power_pairs, sum_points, check)
ok = farm.sharded_transform_ratios(acc, *args, dist=dist_mod, seed=bytes(range(16)))
# two alpha_g1 elements swapped: still subgroup points, no longer geometric
off = 64 + full.powers_g1_length * sizes[1] + full.powers_length * sizes[2]
bad = bytearray(acc)
u = sizes[1]
bad[off + u:off + 2 * u], bad[off + 2 * u:off + 3 * u] = bad[off + 2 * u:off + 3 * u], bad[off + u:off + 2 * u]
bad_ok = farm.sharded_transform_ratios(bytes(bad), *args, dist=dist_mod, seed=bytes(range(16)))
print(json.dumps({{"rank": rank, "ok": ok, "bad_ok": bad_ok, "calls": calls[:4]}}))
if dist_mod is not None:
    dist.destroy_process_group()
'''


def test_slice_bounds_cover_every_pair_once():
    load_package()
    from snark_setup_operator_amd import farm
    for n_pairs in (0, 1, 6, 7, (1 << 21) - 2):
        for world in (1, 2, 3, 8):
            b = [farm.slice_bounds(n_pairs, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n_pairs
            assert all(b[r][1] == b[r + 1][0] for r in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_sharded_transform_ratios_world_size_2_gloo(tmp_path):
    """The one exchange step of the path (final transcript ratio check): each rank reduces its slice with the shared
    coefficient stream, all_gather of the partial pairs, one pairing tail; verdict equals the unsharded run."""
    import json
    port = 31500 + (os.getpid() % 2000)
    script = tmp_path / "ratio_worker.py"
    script.write_text(RATIO_WORKER.format(root=ROOT, port=port))
    single = subprocess.run([sys.executable, str(script), "0", "1"], capture_output=True, text=True, timeout=600)
    assert single.returncode == 0, single.stderr[-2000:]
    one = json.loads(single.stdout.strip().splitlines()[-1])
    assert one["ok"] is True and one["bad_ok"] is False
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    assert all(o["ok"] is True and o["bad_ok"] is False for o in outs)
    # tau_g1 has 7 elements = 6 pairs: rank 0 reduces pairs 0..2 (4 points), rank 1 pairs 3..5 from global index 3
    by_rank = {o["rank"]: o["calls"] for o in outs}
    assert by_rank[0][0] == [1, 4, 0] and by_rank[1][0] == [1, 4, 3]
