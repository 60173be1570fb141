"""GPU test of the N>1 exchange step: the final transcript ratio check sharded over two processes (gloo between
them — two ranks cannot share one device under RCCL — each rank's reductions running on cuda:0 through the C ABI)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, json, random
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
from conftest import load_package
pkg = load_package()
rank, world = int(sys.argv[1]), int(sys.argv[2])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=world)
name, power, cs = "bls12_377", 9, 256
rng = random.Random(5)
keys = [rng.randrange(1, 1 << 250) for _ in range(3)]
full = pkg.Phase1Parameters.new_full(name, 0, power, cs)
n_chunks = -(-full.powers_g1_length // cs)
responses = []
for i in range(n_chunks):
    prm = pkg.Phase1Parameters.new_chunk(name, 0, i, cs, 0, power, cs)
    ch = pkg.phase1_cli.initialization(prm)
    resp = bytearray(pkg.phase1_cli.computation(ch, prm, pkg.PrivateKey(*keys, bytes(prm.public_key_size))))
    resp[:64] = pkg.calculate_hash(ch)
    responses.append(bytes(resp))
combined = pkg.combine(responses, cs, full)
pkg.phase1_cli.transform_ratios_sharded(combined, full, dist=dist)          # honest: passes on every rank
pkg.transform_ratios(combined, pkg.CheckForCorrectness.No, full)            # and agrees with the unsharded check
u1 = 96
off = 64 + (full.powers_g1_length - 3) * u1        # swap two tau_g1 elements in the LAST rank's slice
bad = bytearray(combined)
bad[off:off + u1], bad[off + u1:off + 2 * u1] = bad[off + u1:off + 2 * u1], bad[off:off + u1]
verdict = "accepted"
try:
    pkg.phase1_cli.transform_ratios_sharded(bytes(bad), full, dist=dist)
except pkg.PtauError as e:
    verdict = "rejected %d" % e.code
print(json.dumps({{"rank": rank, "tampered": verdict}}))
dist.destroy_process_group()
'''


def test_sharded_transform_ratios_two_processes_on_gpu(tmp_path):
    port = 33500 + (os.getpid() % 2000)
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    assert [o["tampered"] for o in outs] == ["rejected -6", "rejected -6"]
