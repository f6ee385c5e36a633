"""Audit of the generated code of ffn_fused.hip (veto_amd/asmcheck.py does the work): hazards around the inline-asm MFMAs, missing
pads, compiler uses of M0.  usage: python tools/audit_ffn_asm.py [file.s]      (without a file: compiles ffn_fused.hip first)"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veto_amd import asmcheck

path = sys.argv[1] if len(sys.argv) > 1 else asmcheck.compile_asm(tempfile.mkdtemp(prefix="veto_asm_"))
found = asmcheck.hazards(path) + asmcheck.unpadded(path) + ["compiler instruction touches m0: %s" % t for t in asmcheck.m0_users(path)]
for f in found:
    print(f)
print("%s: %d findings" % (path, len(found)))
sys.exit(1 if found else 0)
