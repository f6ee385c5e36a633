"""Per-kernel summary of the generated code of ffn_fused.hip (veto_amd/asmcheck.py): registers, scratch traffic and compiler-generated
vector-memory waits by barrier interval.  usage: python tools/ffn_asm_stats.py [file.s] [mode]"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veto_amd import asmcheck

path = sys.argv[1] if len(sys.argv) > 1 else asmcheck.compile_asm(tempfile.mkdtemp(prefix="veto_asm_"))
want = int(sys.argv[2]) if len(sys.argv) > 2 else None
for mode, k in sorted(asmcheck.stats(path).items()):
    if want is not None and mode != want:
        continue
    print("== MODE %d: %d barriers, %d scratch ops, %d compiler vmcnt waits, %s VGPRs, scratch %s B" % (
        mode, k["barriers"], k["scratch_ops"], k["compiler_vmcnt_waits"], k["vgprs"], k["scratch_bytes"]))
    print("   interval: scratch ops / vmcnt waits   " + "  ".join("%d: %d/%d" % (nb, v[0], v[1]) for nb, v in sorted(k["by_interval"].items())))
