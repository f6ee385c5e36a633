"""Checks on the GENERATED code of veto_amd/csrc/ffn_fused.hip and veto_amd/csrc/qkv_attn_fused.hip (gfx950).

Both kernels' GEMM MFMAs are inline asm with tied accumulators (140 - 192 of 256 registers are accumulators; the compiler's own MFMA
forms rename and spill them), so the compiler pads no MFMA hazard around them and counts none of the kernels' LDS-DMA.  What keeps
the kernel correct is instruction placement, and this module is what checks it after every build:

  * hazards()      every compiler instruction that reads or writes a register an INLINE-ASM MFMA wrote fewer than 18 wait states
                   earlier, and every vector write of an MFMA operand fewer than 2 states ahead of the MFMA (MFMAs the compiler
                   emitted itself -- the attention products of qkv_attn_fused.hip -- get their wait states from the compiler);
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
    from .build import FLAGS      # the flags the shipped objects are compiled with: the audited code is the shipped code
    cmd = [hipcc] + list(FLAGS) + ["-S", "--cuda-device-only"]
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


def _instructions(path, labels=False):
    """(text, inside_inline_asm) for every instruction line of the file, in order; with labels=True also ("label:", False) rows."""
    out, inasm = [], False
    for raw in open(path):
        s = raw.strip()
        if s.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if s.startswith(";;#ASMEND"):
            inasm = False
            continue
        if labels and re.match(r"^[.\w$]+:", s) and not s.startswith(".amdhsa"):
            out.append((s.split(":")[0] + ":", False))
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        out.append((s.split(";")[0].strip(), inasm))
    return out


def hazards(path):
    """List of findings (strings); empty when the generated code keeps the MFMA rules.

    Control flow: the scan is linear, but at every label the "states since the last MFMA write" of a register is merged (minimum)
    with what a first pass recorded at each branch to that label -- forward branches and loop back-edges alike -- so a result read
    too early across a back-edge or behind a branch target is seen.  (One merge round, not a fixpoint: a hazard that only exists
    after two trips through different back-edges in a row would need a second round; every branch distance is an under-estimate of
    the real one because taken branches cost cycles that are not counted.)"""
    instrs = _instructions(path, labels=True)

    def scan(snapshots, collect):
        written_at = {}     # register -> wait-state clock of the last MFMA write
        valu_write_at = {}  # register -> clock of the last non-MFMA vector write
        clock, found = 0, []
        for text, inasm in instrs:
            if text.endswith(":"):
                for dist in snapshots.get(text[:-1], ()):
                    for r, d in dist.items():
                        written_at[r] = max(written_at.get(r, -10 ** 9), clock - d)
                continue
            op, ops = _split_ops(text)
            if op == "s_nop":
                clock += int(ops[0]) + 1
                continue
            if op in ("s_branch",) or op.startswith("s_cbranch"):
                if collect and ops:
                    snapshots.setdefault(ops[-1], []).append({r: clock + 1 - t for r, t in written_at.items() if clock + 1 - t < MIN_STATES})
                clock += 1
                continue
            if op.startswith("v_mfma"):
                dst = _regs(ops[0])
                srcs = set()
                for o in ops[1:]:
                    srcs |= _regs(o)
                for r in srcs - dst:
                    if r in valu_write_at and clock - valu_write_at[r] < 2:
                        found.append("vector write of v%d %d states ahead of: %s" % (r, clock - valu_write_at[r], text))
                    # an MFMA result used as the A / B operand of a later MFMA (not the tied accumulator) needs the full distance too
                    if r in written_at and clock - written_at[r] < MIN_STATES:
                        found.append("MFMA operand v%d %d states behind an MFMA write: %s" % (r, clock - written_at[r], text))
                for r in (_regs(ops[3]) if len(ops) > 3 else set()):
                    if r in valu_write_at and clock - valu_write_at[r] < 2:
                        found.append("vector write of accumulator v%d %d states ahead of: %s" % (r, clock - valu_write_at[r], text))
                if inasm:       # (an MFMA the compiler emitted itself gets its wait states from the compiler: only the inline-asm ones are tracked)
                    for r in dst:
                        written_at[r] = clock
                else:
                    for r in dst:
                        written_at.pop(r, None)
                clock += 1
                continue
            touched = set()
            for o in ops:
                touched |= _regs(o)
            for r in touched:
                if r in written_at and clock - written_at[r] < MIN_STATES:
                    found.append("%s touches v%d %d states behind an MFMA write" % (text, r, clock - written_at[r]))
                    break
            # (v_cmp* / v_readfirstlane / v_readlane write a scalar destination: their first operand names no vector register)
            if op.startswith(("v_", "ds_read", "ds_bpermute", "global_load_dword", "scratch_load")) and ops and not op.startswith(("v_cmp", "v_readfirstlane", "v_readlane")):
                for r in _regs(ops[0]):
                    valu_write_at[r] = clock
                    written_at.pop(r, None)
            clock += 1
        return found

    snapshots = {}
    scan(snapshots, True)
    return scan(snapshots, False)


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


# source -> (kernel name pattern: the integer template argument as group 1, the boolean ones behind it as group 2; the instantiations that
# must be present).  An instantiation's key is its integer argument, with the boolean arguments appended as digits when any of them is set:
# ffn_fused_kernel<2, true, true, false> (the layer tail on 3-byte residual rows) is 2110, <2, false, false, true> (its single-pass form) 2001.
KERNELS = {"ffn_fused.hip": (r"ffn_fused_kernelILi(\d+)E((?:Lb[01]E)*)", (0, 1, 2, 2001, 2100, 2110)),
           "qkv_attn_fused.hip": (r"qkv_attn_fused_kernelILi(\d+)E((?:Lb[01]E)*)", (72, 96, 721, 961))}


def _inst_key(m):
    bools = re.findall(r"Lb([01])E", m.group(2) or "")
    return int(m.group(1) + "".join(bools)) if "1" in bools else int(m.group(1))


def stats(path, pattern=KERNELS["ffn_fused.hip"][0]):
    """{template argument: dict(barriers, scratch_ops, compiler_vmcnt_waits, vgprs, scratch_bytes, by_interval)} per kernel instantiation."""
    text = open(path).read().split("\n")
    starts = [(i, _inst_key(re.search(pattern, l))) for i, l in enumerate(text)
              if re.match(r"^_ZN4veto.*" + pattern + r".*:", l)]
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


def problems(path, parse_failures=None, source="ffn_fused.hip", scratch_ok=()):
    """Every finding of every check, as strings.  parse_failures: a list that receives "could not find / read" findings instead of the
    result (a change in the compiler's output format is not a hazard: __graft_entry__.build() prints those and goes on).
    scratch_ok: instantiations whose scratch use is reported by stats() but is not a finding."""
    out = list(hazards(path)) + list(unpadded(path))
    out += ["compiler instruction touches m0: %s" % t for t in m0_users(path)]
    pattern, wanted = KERNELS[source]
    st = stats(path, pattern)
    for mode in wanted:
        if mode not in st:
            (out if parse_failures is None else parse_failures).append("kernel instantiation <%d> of %s not found in %s" % (mode, source, path))
            continue
        k = st[mode]
        if (k["scratch_ops"] or k["scratch_bytes"]) and mode not in scratch_ok:
            out.append("MODE %d: %d scratch instructions, %s bytes of scratch (a spill shares vmcnt with the LDS-DMA)" % (mode, k["scratch_ops"], k["scratch_bytes"]))
        if k["vgprs"] is None:
            (out if parse_failures is None else parse_failures).append("MODE %d: register count not found in %s" % (mode, path))
        elif k["vgprs"] > MAX_VGPRS:
            out.append("MODE %d: %s VGPRs (two waves per SIMD need <= %d)" % (mode, k["vgprs"], MAX_VGPRS))
    return out


def check(extra_flags=(), keep_dir=None, parse_failures=None, source="ffn_fused.hip", scratch_ok=()):
    """Compiles one of the audited sources (KERNELS) to assembly and raises RuntimeError on any finding.  Returns the stats.
    What the audit covers: hazards around the inline-asm MFMAs (linear scan + one merge round at labels, see hazards()), their
    s_nop pads, M0, scratch, register count.  It does not model taken-branch timing or LDS / memory ordering."""
    d = keep_dir or tempfile.mkdtemp(prefix="veto_asm_")
    try:
        path = compile_asm(d, extra_flags, source)
        bad = problems(path, parse_failures, source, scratch_ok)
        if bad:
            raise RuntimeError("generated code of %s fails its audit (%d findings):\n  %s" % (source, len(bad), "\n  ".join(bad[:20])))
        return stats(path, KERNELS[source][0])
    finally:
        if keep_dir is None:
            shutil.rmtree(d, ignore_errors=True)
