"""Per-kernel summary of the generated code of ffn_fused.hip: compiler-generated vector-memory waits and scratch (spill) traffic,
by barrier interval.  usage: python tools/ffn_asm_stats.py build/ffn_fused.s [mode]
(hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only veto_amd/csrc/ffn_fused.hip -o build/ffn_fused.s)"""
import re
import sys

text = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else None
starts = [(i, re.search(r"ffn_fused_kernelILi(\d)E", l).group(1)) for i, l in enumerate(text) if re.match(r"^_ZN4veto.*ffn_fused_kernelILi\dE.*:", l)]
for (i0, mode) in starts:
    if want and mode != want:
        continue
    i1 = next(j for j in range(i0, len(text)) if "s_endpgm" in text[j])
    inasm, nbar, rows = False, 0, []
    vg = None
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
            rows.append((nbar, j - i0 + 1, s.split(";")[0].strip()))
    for j in range(i1, min(i1 + 400, len(text))):
        m = re.search(r"\.vgpr_count:\s+(\d+)|NumVgprs: (\d+)|ScratchSize: (\d+)", text[j])
        if m:
            vg = (vg or "") + " " + text[j].strip("; \t")
    print("== MODE %s: %d barriers, %d scratch ops, %d compiler vmcnt waits %s" % (
        mode, nbar, sum(r[2].startswith("scratch_") for r in rows), sum("vmcnt" in r[2] for r in rows), vg or ""))
    by = {}
    for nb, ln, s in rows:
        k = by.setdefault(nb, [0, 0, []])
        k[0] += s.startswith("scratch_")
        k[1] += "vmcnt" in s
    print("   interval: scratch ops / vmcnt waits   " + "  ".join("%d: %d/%d" % (nb, k[0], k[1]) for nb, k in sorted(by.items())))
