"""bench.py --gpus N starts its own N ranks (no torchrun): environment of the children, relay of rank 0's line, failure
propagation.  CPU-only: the children here are a stand-in script, not the GPU workload."""
import importlib.util
import json
import os
import subprocess
import sys

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


CHILD = r'''
import json, os, sys
rank = int(os.environ["RANK"])
if "--fail-rank" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-rank") + 1]):
    sys.exit(7)
if "--hang-others" in sys.argv and rank != 0:
    import time
    time.sleep(600)
print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                  "HSA_ENABLE_IPC_MODE_LEGACY", "PTAU_BENCH_CHILD")} | {"argv": sys.argv[1:]}))
'''


def test_rank_env_is_what_torchrun_would_export():
    b = _bench()
    e = b.rank_env(3, 8, 29999, base_env={"PATH": "/bin"})
    assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"], e["MASTER_PORT"]) == ("3", "3", "8", "127.0.0.1", "29999")
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/bin"


def test_launcher_relays_rank0_and_sets_env(tmp_path, capfd):
    b = _bench()
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    rc = b.launch_ranks(3, ["--gpus", "3", "--steps", "2"], script=str(child))
    assert rc == 0
    line = capfd.readouterr().out.strip().splitlines()
    assert len(line) == 1          # ONE line: rank 0's
    d = json.loads(line[0])
    assert d["RANK"] == "0" and d["WORLD_SIZE"] == "3" and d["MASTER_ADDR"] == "127.0.0.1" and d["argv"] == ["--gpus", "3", "--steps", "2"]


def test_launcher_propagates_failure_and_stops_survivors(tmp_path, capfd):
    b = _bench()
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    rc = b.launch_ranks(3, ["--fail-rank", "1", "--hang-others"], script=str(child))
    assert rc == 7
    assert "rank 1 exited with code 7" in capfd.readouterr().err


def test_parent_does_not_import_torch_before_spawning(tmp_path):
    """`python bench.py --gpus 2` on a box without GPUs: the parent must spawn (and report the children's failure) without
    importing torch or loading the HIP library itself — a process that initialised the GPU must never fork."""
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    rc = e.code\n"
            "print('TORCH_IN_PARENT', 'torch' in sys.modules, 'RC', rc)\n" % os.path.join(ROOT, "bench.py"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "TORCH_IN_PARENT False" in p.stdout, p.stdout + p.stderr
    assert "RC 0" not in p.stdout          # no GPU here: the ranks fail and the parent says so


def test_only_json_lines_of_rank0_reach_stdout(tmp_path, capfd):
    """A library of rank 0 may print to stdout before the line (gloo: "[Gloo] Rank 0 is connected ..."): the parent passes such
    lines on through stderr, stdout stays the ONE JSON line the contract promises."""
    b = _bench()
    script = tmp_path / "child.py"
    script.write_text('import json, os\nif os.environ["RANK"] == "0":\n    print("[Gloo] Rank 0 is connected to 1 peer ranks.")\n'
                      '    print(json.dumps({"metric": "m", "value": 1}))\n')
    rc = b.launch_ranks(2, [], script=str(script), timeout=60)
    out, err = capfd.readouterr()
    assert rc == 0
    assert out.strip().splitlines() == ['{"metric": "m", "value": 1}'] and "[Gloo]" in err


def test_workdir_with_room_or_a_clean_error(tmp_path, monkeypatch):
    """The file-level measurements map hundreds of megabytes per lane: pick_workdir returns a directory on a file system with
    the room (tmpfs first), and stops with a message — not a SIGBUS inside an mmap'd write — when nothing has it."""
    import shutil
    b = _bench()
    d, base = b.pick_workdir(1 << 20, "ptau_test_")
    try:
        assert os.path.isdir(d) and os.path.dirname(d) == base
    finally:
        shutil.rmtree(d, ignore_errors=True)
    import pytest
    with pytest.raises(SystemExit) as e:
        b.pick_workdir(1 << 62, "ptau_test_")
    assert "scratch space" in str(e.value) and "GB free" in str(e.value)
    assert 1 <= b.host_cpu_share() <= (os.cpu_count() or 1)
