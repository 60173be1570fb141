#!/usr/bin/env python3
"""Finds vector instructions placed ahead of the EXEC restore at a control-flow join.

The AMDGPU backend ends a divergent `if` with `s_or_b64 exec, exec, <saved>` as the first instruction of the join block.
Register-allocator copies (live-range splits) are inserted at the top of a block; when scalar moves already sit there the
"block prologue" scan stops early and VECTOR copies land BEFORE the restore, so they execute under the mask of the branch
that is ending - with a uniformly false condition (`s_cbranch_execz` taken) under EXEC = 0, i.e. not at all.  That is a
miscompile (ROCm 7.2's clang did it to the BLS12-377 G2 plain-window kernel: DESIGN.md 4.4b) and it is invisible in the
source, so the build is checked instead:

    isa_exec_prologue_check.py file.s [...]         # hipcc -S --cuda-device-only output
    isa_exec_prologue_check.py --object file.o [...] # a built object: the gfx950 code object is unbundled and disassembled
                                                    # (clang-offload-bundler + llvm-objdump --symbolize-operands), so the
                                                    # check sees the very code that ships (the Makefile does this per object)

Reports every join block whose restore is preceded by VALU/VMEM instructions; exit status 1 when any is found.

Two forms of the join are recognised.  (1) The labelled one: `s_and_saveexec_b64 S` (+ `s_xor_b64 S2, exec, S` of an
if / else) + `s_cbranch_execz L`; block L sets EXEC from S (`s_or_b64 exec, exec, S`) or, as the head of the else block, from
S2 (`s_or_saveexec_b64 T, S2` - what stands between that and the following `s_xor_b64 exec, exec, T` runs for the lanes of
both sides by the backend's design and is not flagged) - in both kinds of listing.  The exit of a divergent loop
(`s_andn2_b64 exec, exec, B` + a branch on EXEC) is the same thing with the break mask B: the block the loop leaves to -
the target of a forward `s_cbranch_execz`, or what follows the backward `s_cbranch_execnz` - runs with EXEC = 0 until its
`s_or_b64 exec, exec, B`.  (2) The fall-through one: the backend drops the skip branch of a short body, the join then
has no label and only the compiler's own listing still shows where it starts (`; %bb.N:`), so this form is checked in .s
files only: the Makefile keeps the device assembly of every object (-save-temps; same code, compared once) for it.
"""
import os
import re
import subprocess
import sys
import tempfile

LABEL = re.compile(r"^(?:[0-9a-f]{8,16} <([^>]+)>|(\.LBB\d+_\d+|_Z\w+|[A-Za-z_]\w*)):")   # objdump form | .s form
SAVE = re.compile(r"\bs_and_saveexec_b64\s+(s\[\d+:\d+\])")
SKIP = re.compile(r"\bs_cbranch_execz\s+(\.LBB\d+_\d+|L\d+)\b")
BLOCK = re.compile(r"^(\.L\w+|L\d+)$")
LOOP_MASK = re.compile(r"\bs_andn2_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\])")
LOOP_BACK = re.compile(r"\bs_cbranch_execnz\s+(\.LBB\d+_\d+|L\d+)\b")
FLIP = re.compile(r"\bs_xor_b64\s+(s\[\d+:\d+\]),\s*exec,\s*(s\[\d+:\d+\])")
VECTOR = re.compile(r"^\s*(v_|global_|flat_|scratch_|buffer_|ds_)")
ENDS = re.compile(r"^\s*(s_cbranch|s_branch|s_endpgm|s_setpc|s_swappc)")
# vector instructions that do not depend on EXEC are harmless ahead of the restore
EXEC_FREE = re.compile(r"^\s*(v_readlane_b32|v_readfirstlane_b32|v_writelane_b32)")


def instructions(path):
    with open(path, errors="replace") as f:
        for ln, raw in enumerate(f, 1):
            line = raw.split("//")[0].split(";")[0].rstrip()
            if line.strip():
                yield ln, line


def label_of(line):
    m = LABEL.match(line)
    return (m.group(1) or m.group(2)) if m else None


EXEC_WRITE = re.compile(r"^\s*s_(?:or|andn2|xor|mov|and)(_saveexec)?_b64\s+(.*)$")


def writes_exec_from(line, saved):
    """The instruction that ends a join's prologue: it sets EXEC from one of the saved masks - `s_or_b64 exec, exec, S` at
    the end of an if, `s_or_saveexec_b64 T, S` / `s_andn2_saveexec_b64 T, S` at the head of an else block."""
    m = EXEC_WRITE.match(line)
    if not m:
        return False
    ops = [o.strip() for o in m.group(2).split(",")]
    if not (m.group(1) or ops[0] == "exec"):
        return False
    return any(o in saved for o in ops[1:])


def check(path):
    """-> [(function, label, line, [(line, text), ...])]"""
    lines = list(instructions(path))
    # the skip branch of every divergent if: (function, label) -> the SGPR pairs holding the mask to restore there
    joins = {}
    fallouts = []          # (function, index of the backward branch, break mask)
    func = "?"
    for i, (ln, line) in enumerate(lines):
        name = label_of(line)
        if name is not None and not BLOCK.match(name):
            func = name
        lp = LOOP_MASK.search(line)
        if lp:
            # a divergent loop: lanes leave by clearing their EXEC bit; the exit block restores them all from the break mask
            for j, (_, nxt) in enumerate(lines[i + 1:i + 4]):
                k = SKIP.search(nxt)
                if k:                                   # exit by a forward skip: the labelled block restores
                    joins.setdefault((func, k.group(1)), set()).add(lp.group(1))
                    break
                if LOOP_BACK.search(nxt):               # exit by falling out of the backward branch
                    fallouts.append((func, i + 1 + j, lp.group(1)))
                    break
                if label_of(nxt) is not None or VECTOR.match(nxt):
                    break
            continue
        m = SAVE.search(line)
        if not m:
            continue
        masks = {m.group(1)}
        for _, nxt in lines[i + 1:i + 4]:
            x = FLIP.search(nxt)
            if x and x.group(2) in masks:
                masks.add(x.group(1))     # if / else: the else block restores from the flipped mask
                continue
            k = SKIP.search(nxt)
            if k:
                joins.setdefault((func, k.group(1)), set()).update(masks)
                break
            if label_of(nxt) is not None or VECTOR.match(nxt):
                break
    found = []
    for func, at, mask in fallouts:
        pending = []
        for l2, t in lines[at + 1:]:
            if label_of(t) is not None and not BLOCK.match(label_of(t)):
                break
            if label_of(t) is not None:
                continue                 # the exit block may carry a label of its own
            if ENDS.match(t):
                break
            if writes_exec_from(t, {mask}):
                if pending:
                    found.append((func, "(loop exit)", lines[at][0], pending))
                break
            if VECTOR.match(t) and not EXEC_FREE.match(t):
                pending.append((l2, t.strip()))
            elif re.search(r"\bexec\b", t):
                break
    func = "?"
    for i, (ln, line) in enumerate(lines):
        name = label_of(line)
        if name is None:
            continue
        if not BLOCK.match(name):
            func = name
            continue
        if (func, name) not in joins:
            continue
        saved = joins[(func, name)]
        pending = []
        for l2, t in lines[i + 1:]:
            if label_of(t) is not None or ENDS.match(t):
                break                    # no restore in this block: the mask is restored further on (nested joins)
            if writes_exec_from(t, saved):
                if pending:
                    found.append((func, name, ln, pending))
                break
            if VECTOR.match(t) and not EXEC_FREE.match(t):
                pending.append((l2, t.strip()))
    return found


MBB = re.compile(r"^; %bb\.\d+:")
BRANCH_ON_EXEC = re.compile(r"^\s*s_cbranch_exec(z|nz)\b")
RESTORE_ANY = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\])")


def check_fallthrough(path):
    """Form (2), compiler listings only: `s_and_saveexec_b64 S`, a body block without a skip branch or labels, then the join
    block `; %bb.J:` that restores S.  -> the same tuples as check()."""
    found = []
    func = "?"
    stage, sreg, join_at, pending = None, None, None, []
    with open(path, errors="replace") as f:
        for ln, raw in enumerate(f, 1):
            if MBB.match(raw):
                if stage == "saved":
                    stage = "body"
                elif stage == "body":
                    stage, join_at, pending = "join", ln, []
                else:
                    stage = None
                continue
            line = raw.split(";")[0].rstrip()
            if not line.strip() or line.lstrip().startswith("."):
                name = label_of(line)
                if name is not None:
                    stage = None          # a label: not the simple fall-through shape
                continue
            name = label_of(line)
            if name is not None:
                if not BLOCK.match(name):
                    func = name
                stage = None
                continue
            m = SAVE.search(line)
            if m:
                stage, sreg = "saved", m.group(1)
                continue
            if stage == "saved":
                stage = None              # a skip branch (form 1) or anything else between the save and the body
                continue
            if stage == "body":
                if ENDS.match(line) or RESTORE_ANY.match(line):
                    stage = None          # body and join in one block, or control flow inside the body
                continue
            if stage == "join":
                r = RESTORE_ANY.match(line)
                if r:
                    if r.group(1) == sreg and pending:
                        found.append((func, "(fall-through join)", join_at, pending))
                    stage = None
                elif VECTOR.match(line) and not EXEC_FREE.match(line):
                    pending.append((ln, line.strip()))
                elif ENDS.match(line) or re.search(r"\bexec\b", line):
                    stage = None
    return found


LLVM_BIN = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def disassemble(obj, out):
    """The gfx950 code object bundled in a hipcc object -> out (text); False when the object has no device code."""
    fat, co = out + ".fatbin", out + ".co"
    subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.devnull],
                   capture_output=True, text=True)
    if not os.path.exists(fat):
        return False                     # host-only object: nothing to check
    r = subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co],
                       capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
        raise RuntimeError(f"{obj}: no gfx950 code object in it ({r.stderr.strip()})")
    with open(out, "w") as f:
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--symbolize-operands", co], stdout=f,
                       stderr=subprocess.DEVNULL, check=True)
    os.unlink(co)
    os.unlink(fat)
    return True


def main(argv):
    total = 0
    objects = bool(argv) and argv[0] == "--object"
    if objects:
        argv = argv[1:]
    if not argv:
        print(__doc__)
        return 2
    for path in argv:
        if objects:
            with tempfile.TemporaryDirectory() as d:
                text = os.path.join(d, "isa.s")
                hits = check(text) if disassemble(path, text) else []
        else:
            hits = check(path) + check_fallthrough(path)
        for func, label, ln, ins in hits:
            total += 1
            print(f"{path}:{ln}: {func[:90]} block {label}: {len(ins)} vector instruction(s) before the EXEC restore")
            for l, t in ins[:4]:
                print(f"    {l}: {t}")
    print(f"{total} suspicious join block(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
