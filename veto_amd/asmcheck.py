"""Checks on the GENERATED code of veto_amd/csrc/ffn_fused.hip (gfx950).

The panel kernel's MFMAs are inline asm with tied accumulators (192 of 256 registers are accumulators; the compiler's own MFMA
forms rename and spill them), so the compiler pads no MFMA hazard around them and counts none of the kernel's LDS-DMA.  What keeps
the kernel correct is instruction placement, and this module is what checks it after every build:

  * hazards()      every compiler instruction that reads or writes a register an MFMA wrote fewer than 18 wait states earlier,
                   and every vector write of an MFMA operand fewer than 2 states ahead of the MFMA;
  * unpadded()     every inline-asm MFMA that is not opened by its own `s_nop 1` (the pad that makes the second rule hold by
                   construction, whatever the compiler puts in front of the statement);
  * m0_users()     compiler-generated instructions that touch M0 (the LDS-DMA statements write it without being able to declare it);
  * stats()        per kernel: registers, scratch (spill) instructions, compiler-inserted vector-memory waits per barrier interval.

`check()` runs all of them and raises; `__graft_entry__.build()` and tests/test_ffn_asm.py call it, tools/audit_ffn_asm.py and
tools/ffn_asm_stats.py print the details.  Pure text processing: no GPU, no torch.
"""
import os
import re
import shutil
import subprocess
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
MIN_STATES = 18      # MFMA result -> any non-MFMA reader / writer of the register (16-pass forms: 18 wait states)
MAX_VGPRS = 256      # two waves per SIMD


def compile_asm(out_dir, extra_flags=(), source="ffn_fused.hip"):
    """hipcc -S of one source for gfx950 (device side only); returns the path of the .s file."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found")
    out = os.path.join(out_dir, os.path.splitext(source)[0] + ".s")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-Wno-unused-result", "-Wno-unused-value"]
    cmd += list(extra_flags) + [source, "-o", out]
    subprocess.run(cmd, cwd=CSRC, check=True)
    return out


def _regs(tok):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


def _split_ops(text):
    parts = text.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    return parts[0], [p.strip() for p in parts[1].split(",")]


def _instructions(path):
    """(text, inside_inline_asm) for every instruction line of the file, in order."""
    out, inasm = [], False
    for raw in open(path):
        s = raw.strip()
        if s.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if s.startswith(";;#ASMEND"):
            inasm = False
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        out.append((s.split(";")[0].strip(), inasm))
    return out


def hazards(path):
    """List of findings (strings); empty when the generated code keeps both MFMA rules."""
    written_at = {}     # register -> wait-state clock of the last MFMA write
    valu_write_at = {}  # register -> clock of the last non-MFMA vector write
    clock, found = 0, []
    for text, _ in _instructions(path):
        op, ops = _split_ops(text)
        if op == "s_nop":
            clock += int(ops[0]) + 1
            continue
        if op.startswith("v_mfma"):
            dst = _regs(ops[0])
            srcs = set()
            for o in ops[1:]:
                srcs |= _regs(o)
            for r in srcs - dst:
                if r in valu_write_at and clock - valu_write_at[r] < 2:
                    found.append("vector write of v%d %d states ahead of: %s" % (r, clock - valu_write_at[r], text))
            for r in (_regs(ops[3]) if len(ops) > 3 else set()):
                if r in valu_write_at and clock - valu_write_at[r] < 2:
                    found.append("vector write of accumulator v%d %d states ahead of: %s" % (r, clock - valu_write_at[r], text))
            for r in dst:
                written_at[r] = clock
            clock += 1
            continue
        touched = set()
        for o in ops:
            touched |= _regs(o)
        for r in touched:
            if r in written_at and clock - written_at[r] < MIN_STATES:
                found.append("%s touches v%d %d states behind an MFMA write" % (text, r, clock - written_at[r]))
                break
        if op.startswith(("v_", "ds_read", "ds_bpermute", "global_load_dword", "scratch_load")) and ops:
            for r in _regs(ops[0]):
                valu_write_at[r] = clock
                written_at.pop(r, None)
        clock += 1
    return found


def unpadded(path):
    """Inline-asm MFMAs whose statement does not open with `s_nop 1` (or longer)."""
    found, prev, prev_inasm = [], None, False
    n_mfma = 0
    for text, inasm in _instructions(path):
        op, ops = _split_ops(text)
        if op.startswith("v_mfma") and inasm:
            n_mfma += 1
            p_op, p_ops = _split_ops(prev) if prev else ("", [])
            if not (prev_inasm and p_op == "s_nop" and int(p_ops[0]) >= 1):
                found.append("no s_nop 1 in front of: %s" % text)
        prev, prev_inasm = text, inasm
    if n_mfma == 0:
        found.append("no inline-asm MFMA found at all (wrong file?)")
    return found


def m0_users(path):
    """Compiler-generated instructions (outside the inline-asm statements) that name M0."""
    return [text for text, inasm in _instructions(path) if not inasm and re.search(r"\bm0\b", text)]


def stats(path):
    """{mode: dict(barriers, scratch_ops, compiler_vmcnt_waits, vgprs, scratch_bytes, by_interval)} for the three panel kernels."""
    text = open(path).read().split("\n")
    starts = [(i, re.search(r"ffn_fused_kernelILi(\d)E", l).group(1)) for i, l in enumerate(text)
              if re.match(r"^_ZN4veto.*ffn_fused_kernelILi\dE.*:", l)]
    out = {}
    for i0, mode in starts:
        i1 = next(j for j in range(i0, len(text)) if "s_endpgm" in text[j])
        inasm, nbar, rows = False, 0, []
        for j in range(i0, i1):
            s = text[j].strip()
            if s.startswith(";;#ASMSTART"):
                inasm = True
                continue
            if s.startswith(";;#ASMEND"):
                inasm = False
                continue
            if s == "s_barrier":
                nbar += 1
            if not inasm and (re.search(r"s_waitcnt.*vmcnt", s) or s.startswith("scratch_")):
                rows.append((nbar, s.split(";")[0].strip()))
        vgprs = scratch = None
        for j in range(i1, min(i1 + 400, len(text))):
            m = re.search(r"; NumVgprs: (\d+)", text[j])
            if m and vgprs is None:
                vgprs = int(m.group(1))
            m = re.search(r"; ScratchSize: (\d+)", text[j])
            if m and scratch is None:
                scratch = int(m.group(1))
        by = {}
        for nb, s in rows:
            k = by.setdefault(nb, [0, 0])
            k[0] += s.startswith("scratch_")
            k[1] += "vmcnt" in s
        out[int(mode)] = dict(barriers=nbar, scratch_ops=sum(r[1].startswith("scratch_") for r in rows),
                              compiler_vmcnt_waits=sum("vmcnt" in r[1] for r in rows), vgprs=vgprs, scratch_bytes=scratch, by_interval=by)
    return out


def problems(path):
    """Every finding of every check, as strings."""
    out = list(hazards(path)) + list(unpadded(path))
    out += ["compiler instruction touches m0: %s" % t for t in m0_users(path)]
    st = stats(path)
    for mode in (0, 1, 2):
        if mode not in st:
            out.append("kernel ffn_fused_kernel<%d> not found in %s" % (mode, path))
            continue
        k = st[mode]
        if k["scratch_ops"] or k["scratch_bytes"]:
            out.append("MODE %d: %d scratch instructions, %s bytes of scratch (a spill shares vmcnt with the LDS-DMA)" % (mode, k["scratch_ops"], k["scratch_bytes"]))
        if k["vgprs"] is None or k["vgprs"] > MAX_VGPRS:
            out.append("MODE %d: %s VGPRs (two waves per SIMD need <= %d)" % (mode, k["vgprs"], MAX_VGPRS))
    return out


def check(extra_flags=(), keep_dir=None):
    """Compiles ffn_fused.hip to assembly and raises RuntimeError on any finding.  Returns the stats."""
    d = keep_dir or tempfile.mkdtemp(prefix="veto_asm_")
    try:
        path = compile_asm(d, extra_flags)
        bad = problems(path)
        if bad:
            raise RuntimeError("generated code of ffn_fused.hip fails its audit (%d findings):\n  %s" % (len(bad), "\n  ".join(bad[:20])))
        return stats(path)
    finally:
        if keep_dir is None:
            shutil.rmtree(d, ignore_errors=True)
