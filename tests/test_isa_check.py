"""tools/isa_exec_prologue_check.py: the build gate against vector instructions ahead of the EXEC restore at a join.

ROCm 7.2's clang placed register-allocator copies of a table entry's x coordinate BEFORE `s_or_b64 exec, exec, s[0:1]` at
the join of `if (negate) y = -y` in the BLS12-377 G2 plain-window kernel; with a positive digit (branch skipped, EXEC = 0)
the copies did not execute and the addition used stale registers.  The Makefile disassembles every device
object and refuses one with that pattern; these tests pin the detector on both listing formats and run it over the objects
of the library that is in the tree.
"""
import glob
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_check", os.path.join(ROOT, "tools", "isa_exec_prologue_check.py"))
isa_check = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa_check)

BAD_S = """
_Z6kernelv:
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_sub_u32_e32 v43, 0x21a3fffe, v43
.LBB0_2:
	v_mov_b64_e32 v[102:103], v[2:3]
	s_mov_b32 s81, 0x29185f12
	s_or_b64 exec, exec, s[0:1]
	v_add_u32_e32 v1, v2, v3
	s_endpgm
.Lfunc_end0:
"""

GOOD_S = """
_Z6kernelv:
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_sub_u32_e32 v43, 0x21a3fffe, v43
.LBB0_2:
	s_mov_b32 s81, 0x29185f12
	v_readlane_b32 s4, v248, 3
	s_or_b64 exec, exec, s[0:1]
	v_mov_b64_e32 v[102:103], v[2:3]
	s_and_saveexec_b64 s[2:3], vcc
; a short body without a skip branch: the restore follows the body in the same block
	v_add_u32_e32 v1, v2, v3
	s_or_b64 exec, exec, s[2:3]
	s_endpgm
.Lfunc_end0:
"""

BAD_OBJDUMP = """
0000000000001000 <_Z6kernelv>:
	s_and_saveexec_b64 s[0:1], vcc                             // 000000001000: BE80206A
	s_cbranch_execz L7                                         // 000000001004: BF880002
	v_sub_u32_e32 v43, 0x21a3fffe, v43                         // 000000001008: 6A5656FF 21A3FFFE

0000000000001010 <L7>:
	v_mov_b64_e32 v[102:103], v[2:3]                           // 000000001010: 7ECC7102
	s_or_b64 exec, exec, s[0:1]                                // 000000001014: 87FE007E
	s_endpgm                                                   // 000000001018: BF810000
"""


FALLTHROUGH_BAD_S = """
_Z6kernelv:
	s_and_saveexec_b64 s[0:1], vcc
; %bb.1:
	v_sub_u32_e32 v43, 0x21a3fffe, v43
; %bb.2:
	s_mov_b32 s81, 0x29185f12
	v_mov_b64_e32 v[102:103], v[2:3]
	s_or_b64 exec, exec, s[0:1]
	s_endpgm
.Lfunc_end0:
"""


def run_on(tmp_path, text):
    p = tmp_path / "k.s"
    p.write_text(text)
    return isa_check.check(str(p))


def test_vector_copy_ahead_of_the_restore_is_found(tmp_path):
    hits = run_on(tmp_path, BAD_S)
    assert len(hits) == 1
    func, label, _, ins = hits[0]
    assert func == "_Z6kernelv" and label == ".LBB0_2"
    assert [t.split()[0] for _, t in ins] == ["v_mov_b64_e32"]
    assert isa_check.main([str(tmp_path / "k.s")]) == 1


def test_scalar_prologue_and_fallthrough_bodies_are_not_flagged(tmp_path):
    assert run_on(tmp_path, GOOD_S) == []
    assert isa_check.main([str(tmp_path / "k.s")]) == 0


def test_join_without_a_skip_branch_is_checked_in_compiler_listings(tmp_path):
    """A short body loses its `s_cbranch_execz`: the join has no label, only the listing's `; %bb.N:` shows it."""
    p = tmp_path / "k.s"
    p.write_text(FALLTHROUGH_BAD_S)
    assert isa_check.check(str(p)) == []
    hits = isa_check.check_fallthrough(str(p))
    assert len(hits) == 1 and hits[0][3][0][1].startswith("v_mov_b64_e32")
    assert isa_check.main([str(p)]) == 1
    p.write_text(FALLTHROUGH_BAD_S.replace("\tv_mov_b64_e32 v[102:103], v[2:3]\n", ""))
    assert isa_check.main([str(p)]) == 0
    # body and join printed as one block (nothing can have been inserted between them): not flagged
    p.write_text(FALLTHROUGH_BAD_S.replace("; %bb.2:\n", ""))
    assert isa_check.main([str(p)]) == 0


ELSE_HEAD_S = """
_Z6kernelv:
	s_and_saveexec_b64 s[0:1], s[36:37]
	s_xor_b64 s[36:37], exec, s[0:1]
	s_cbranch_execz .LBB0_4
; %bb.1:
	v_add_u32_e32 v1, v2, v3
.LBB0_4:
	HEAD
	s_or_saveexec_b64 s[0:1], s[36:37]
	v_mov_b64_e32 v[0:1], s[2:3]
	s_xor_b64 exec, exec, s[0:1]
	s_cbranch_execz .LBB0_9
; %bb.5:
	v_add_u32_e32 v1, v2, v3
.LBB0_9:
	s_or_b64 exec, exec, s[0:1]
	s_endpgm
.Lfunc_end0:
"""


def test_else_block_head(tmp_path):
    """if / else: the skip branch lands on the else block, whose prologue is `s_or_saveexec_b64 T, <flipped mask>` ...
    `s_xor_b64 exec, exec, T`.  Copies BETWEEN the two are the backend's design (they run for the lanes of both sides);
    a vector instruction ahead of the `s_or_saveexec_b64` runs under the then-side's mask only."""
    assert run_on(tmp_path, ELSE_HEAD_S.replace("HEAD", "s_nop 0")) == []
    hits = run_on(tmp_path, ELSE_HEAD_S.replace("HEAD", "v_mov_b32_e32 v7, v9"))
    assert len(hits) == 1 and hits[0][1] == ".LBB0_4" and hits[0][3][0][1] == "v_mov_b32_e32 v7, v9"


LOOP_S = """
_Z6kernelv:
	s_mov_b64 s[14:15], 0
.LBB0_1:
	v_add_u32_e32 v1, v2, v3
	v_cmp_gt_u32_e32 vcc, 2, v1
	s_or_b64 s[14:15], vcc, s[14:15]
	s_andn2_b64 exec, exec, s[14:15]
	BRANCH
	EXITHEAD
	s_or_b64 exec, exec, s[14:15]
	v_add_u32_e32 v1, v2, v3
	s_endpgm
.Lfunc_end0:
"""


@pytest.mark.parametrize("branch,exit_label", [("s_cbranch_execnz .LBB0_1\n; %bb.2:", ""),
                                               ("s_cbranch_execz .LBB0_3\n\ts_branch .LBB0_1\n.LBB0_3:", ".LBB0_3")])
def test_exit_of_a_divergent_loop(tmp_path, branch, exit_label):
    """Lanes leave a divergent loop by clearing their EXEC bit: the exit block runs with EXEC = 0 until it restores them
    from the break mask, so a vector copy ahead of that restore never executes."""
    text = LOOP_S.replace("BRANCH", branch)
    assert run_on(tmp_path, text.replace("EXITHEAD", "s_mov_b32 s3, 0")) == []
    hits = run_on(tmp_path, text.replace("EXITHEAD", "v_mov_b32_e32 v9, v1"))
    assert len(hits) == 1 and hits[0][3][0][1] == "v_mov_b32_e32 v9, v1"
    assert hits[0][1] == (exit_label or "(loop exit)")


def test_objdump_listing_with_per_function_labels(tmp_path):
    hits = run_on(tmp_path, BAD_OBJDUMP)
    assert len(hits) == 1 and hits[0][1] == "L7"
    # the same label name in another function is a different block
    two = BAD_OBJDUMP + BAD_OBJDUMP.replace("_Z6kernelv", "_Z5otherv").replace("v_mov_b64_e32 v[102:103], v[2:3]", "s_nop 0")
    hits = run_on(tmp_path, two)
    assert [h[0] for h in hits] == ["_Z6kernelv"]


def test_objects_of_the_library_in_the_tree_are_clean():
    objs = sorted(glob.glob(os.path.join(ROOT, "snark-setup-operator_amd", "csrc", "*.o")))
    tools = [os.path.join(isa_check.LLVM_BIN, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not objs or not all(os.path.exists(t) for t in tools):
        pytest.skip("no built objects or no LLVM tools here")
    assert isa_check.main(["--object"] + objs) == 0
