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
