#!/usr/bin/env python3
"""ISA lint of the built libscaldpc.so: the gfx940-family hazard "a VALU instruction reads an SGPR that a VALU
instruction wrote fewer than 2 wait states earlier" (LLVM GCNHazardRecognizer: VALUWriteSGPRVALUReadWaitstates = 2).

The compiler pads its OWN instructions; what it cannot see is the inside of an `asm` statement, so a hand-written
`v_writelane_b32 v, sN, k` that the register allocator happens to feed from a `v_readlane_b32` spill reload (or a
`v_cmp`) is exactly where this hazard hides ("What's weak" #1: two such sites in k_check_minsum_rec<64,...>).
The product no longer has inline-asm VALU instructions that read SGPRs; this lint is what keeps it so: it walks the
disassembly of every kernel in the gfx950 code objects of the .so and reports each (writer, reader) pair that is too
close -- whoever emitted it.

Usage:  python3 profiles/isa_lint.py [path/to/libscaldpc.so]      exit code 1 when a site is found
Used by tests/test_isa_lint.py (CPU suite).  Needs only llvm-objdump from /opt/rocm/lib/llvm/bin.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
WAIT_STATES = 2  # VALU write of an SGPR -> VALU read of it

# second operand is an SGPR destination too (carry-out / scale flag)
_TWO_DST = ("v_add_co_", "v_sub_co_", "v_subrev_co_", "v_addc_co_", "v_subb_co_", "v_subbrev_co_", "v_div_scale_",
            "v_mad_u64_u32", "v_mad_i64_i32")
_SREG = re.compile(r"^(?:s(\d+)|s\[(\d+):(\d+)\]|(vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0|scc))$")
_LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_FUNC = re.compile(r"^[0-9a-f]+ <(.+)>:$")
VCC = (106, 107)  # SGPR numbers of vcc_lo / vcc_hi on GFX9
EXEC = (126, 127)


def _sregs(op):
    """SGPR numbers an operand names (empty for VGPRs, literals, modifiers)."""
    op = op.strip()
    if op.startswith("|") and op.endswith("|"):
        op = op[1:-1]
    if op.startswith("-"):
        op = op[1:]
    m = _SREG.match(op)
    if not m:
        return ()
    if m.group(1) is not None:
        return (int(m.group(1)),)
    if m.group(2) is not None:
        return tuple(range(int(m.group(2)), int(m.group(3)) + 1))
    name = m.group(4)
    return {"vcc": VCC, "vcc_lo": VCC[:1], "vcc_hi": VCC[1:], "exec": EXEC, "exec_lo": EXEC[:1], "exec_hi": EXEC[1:]}.get(name, ())


def _split_ops(text):
    ops, depth, cur = [], 0, ""
    for ch in text:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur)
    # trailing modifiers ("op_sel:[..]", "clamp", "sc1") ride on the last operand after a space
    return [o.strip().split(" ")[0] for o in ops]


def _is_valu(mn):
    return mn.startswith("v_")


def parse(path_s):
    """-> {kernel: [(addr, mnemonic, [operands], text)]}"""
    funcs, cur = {}, None
    with open(path_s) as f:
        for line in f:
            m = _FUNC.match(line)
            if m:
                cur = funcs.setdefault(m.group(1), [])
                continue
            if cur is None:
                continue
            m = _LINE.match(line)
            if not m:
                continue
            mn, rest, addr = m.group(1), m.group(2), int(m.group(3), 16)
            cur.append((addr, mn, _split_ops(rest), (mn + " " + rest).strip()))
    return funcs


def _writes_reads(mn, ops):
    """SGPRs a VALU instruction writes / reads."""
    ndst = 2 if mn.startswith(_TWO_DST) else 1
    if mn.startswith("v_cmpx"):
        w = set(EXEC)
        r = set(x for o in ops for x in _sregs(o))
        return w, r
    w = set(x for o in ops[:ndst] for x in _sregs(o))
    r = set(x for o in ops[ndst:] for x in _sregs(o))
    return w, r


def _wait_states(mn, ops):
    if mn == "s_nop":
        try:
            return int(ops[0], 0) + 1
        except (ValueError, IndexError):
            return 1
    return 1


def lint_function(insts):
    """-> [(writer_text, reader_text, wait_states_between, reader_addr)]"""
    sites = []
    by_addr = {a: i for i, (a, _, _, _) in enumerate(insts)}
    # recent VALU SGPR writes: list of (set of sgprs, wait states elapsed since, text)
    recent = []

    def check(reader, pending):
        _, mn, ops, text = reader
        if not _is_valu(mn):
            return
        _, rd = _writes_reads(mn, ops)
        for regs, elapsed, wtext in pending:
            if elapsed < WAIT_STATES and regs & rd:
                sites.append((wtext, text, elapsed, reader[0]))

    for i, inst in enumerate(insts):
        addr, mn, ops, text = inst
        check(inst, recent)
        # a branch hands its pending writes to the first instruction of its target (the fall-through is the linear order)
        if mn.startswith(("s_cbranch", "s_branch")) and ops:
            try:
                off = int(ops[0], 0)
                if off >= 0x8000:
                    off -= 0x10000
                tgt = by_addr.get(addr + 4 + 4 * off)
            except ValueError:
                tgt = None
            if tgt is not None:
                handed = [(r, e + 1, t) for r, e, t in recent]
                check(insts[tgt], handed)
        ws = _wait_states(mn, ops)
        # any write of these registers by a non-VALU instruction ends the hazard for them
        if not _is_valu(mn):
            wr = set(x for x in _sregs(ops[0])) if ops and mn.startswith(("s_", "v_")) else set()
            recent = [(r - wr, e, t) for r, e, t in recent]
        recent = [(r, e + ws, t) for r, e, t in recent if e + ws < WAIT_STATES + 2 and r]
        if _is_valu(mn):
            w, _ = _writes_reads(mn, ops)
            if w:
                recent.append((w, 0, text))
    return sites


def extract_code_objects(so_path, workdir):
    objdump = os.path.join(LLVM_BIN, "llvm-objdump")
    local = os.path.join(workdir, os.path.basename(so_path))
    shutil.copy(so_path, local)  # (--offloading writes the bundles next to its input)
    subprocess.run([objdump, "--offloading", local], check=True, stdout=subprocess.DEVNULL)
    out = []
    for f in sorted(os.listdir(workdir)):
        if "amdgcn" in f and "gfx950" in f:
            s = os.path.join(workdir, f + ".s")
            with open(s, "w") as fh:
                subprocess.run([objdump, "-d", os.path.join(workdir, f)], check=True, stdout=fh)
            out.append(s)
    return out


def lint(so_path):
    """-> (sites, stats); a site = (kernel, writer, reader, wait_states, address)."""
    sites, stats = [], {"kernels": 0, "instructions": 0, "v_writelane": 0, "v_readlane": 0}
    with tempfile.TemporaryDirectory() as wd:
        for s in extract_code_objects(so_path, wd):
            for name, insts in parse(s).items():
                stats["kernels"] += 1
                stats["instructions"] += len(insts)
                stats["v_writelane"] += sum(1 for x in insts if x[1] == "v_writelane_b32")
                stats["v_readlane"] += sum(1 for x in insts if x[1] == "v_readlane_b32")
                for w, r, ws, addr in lint_function(insts):
                    sites.append((name, w, r, ws, addr))
    return sites, stats


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "sca-ldpc_amd", "libscaldpc.so")
    sites, stats = lint(so)
    print("isa_lint: %(kernels)d kernels, %(instructions)d instructions, %(v_writelane)d v_writelane, %(v_readlane)d v_readlane" % stats)
    for name, w, r, ws, addr in sites:
        filt = shutil.which("c++filt")
        demangled = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() if filt else name
        print("HAZARD %s @%x: `%s` -> `%s` (%d wait state%s, needs %d)" % (demangled[:100], addr, w, r, ws, "" if ws == 1 else "s", WAIT_STATES))
    print("isa_lint: %d site(s)" % len(sites))
    return 1 if sites else 0


if __name__ == "__main__":
    sys.exit(main())
