"""Audit of the generated code of ffn_fused.hip (its MFMAs are inline asm, so the compiler pads no MFMA hazard): lists every
compiler instruction that reads or writes a register an MFMA wrote fewer than MIN_STATES wait states earlier, and every VALU
write of an MFMA operand fewer than 2 states ahead of it.  usage: python tools/audit_ffn_asm.py <file.s>
(hipcc -O3 --offload-arch=gfx950 -c veto_amd/csrc/ffn_fused.hip --save-temps=obj writes the .s)"""
import re
import sys

MIN_STATES = 18
lines = [l.strip() for l in open(sys.argv[1]) if l.strip() and not l.strip().startswith((";", ".", "//"))]
ins = [l.split(";")[0].strip() for l in lines if not l.endswith(":")]


def regs(tok):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", tok):
        out.add(int(a))
    return out


def split_ops(text):
    parts = text.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    return parts[0], [p.strip() for p in parts[1].split(",")]


written_at = {}     # register -> wait-state clock of the last MFMA write
valu_write_at = {}  # register -> clock of the last non-MFMA vector write
clock = 0
problems = 0
for n, text in enumerate(ins):
    op, ops = split_ops(text)
    if op == "s_nop":
        clock += int(ops[0]) + 1
        continue
    if op.startswith("v_mfma"):
        dst = regs(ops[0])
        srcs = set()
        for o in ops[1:]:
            srcs |= regs(o)
        for r in srcs - dst:
            if r in valu_write_at and clock - valu_write_at[r] < 2:
                print("VALU write of v%d %d states ahead of: %s" % (r, clock - valu_write_at[r], text))
                problems += 1
        for r in (regs(ops[3]) if len(ops) > 3 else set()):
            if r in valu_write_at and clock - valu_write_at[r] < 2:
                print("VALU write of accumulator v%d %d states ahead of: %s" % (r, clock - valu_write_at[r], text))
                problems += 1
        for r in dst:
            written_at[r] = clock
        clock += 1
        continue
    touched = set()
    for o in ops:
        touched |= regs(o)
    for r in touched:
        if r in written_at and clock - written_at[r] < MIN_STATES:
            print("%s touches v%d %d states behind an MFMA write" % (text, r, clock - written_at[r]))
            problems += 1
            break
    if op.startswith(("v_", "ds_read", "ds_bpermute", "global_load_dword", "scratch_load")) and ops:
        for r in regs(ops[0]):
            valu_write_at[r] = clock
            written_at.pop(r, None)
    clock += 1
print("%d instructions, %d findings" % (len(ins), problems))
sys.exit(1 if problems else 0)
