"""Builds profiles/traffic.json (what bench.py puts into `roofline.traffic` and `attention`) from the rocprofv3 --pmc CSVs of
ONE profile tag, so that the bench line and the committed profile always come from the same run set.

    python tools/pmc_traffic.py <tag> <pmc_fetch_dir> <pmc_write_dir> <pmc_mfma_dir> [--workload 12 36 4 8 mixed] [--out traffic.json]

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md, section HBM: FETCH_SIZE / WRITE_SIZE are in KiB and come from
separate passes; on gfx950 FETCH_SIZE reports half of the bytes of a wide (16 B per lane) coalesced read stream, so
bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.  Launches are attributed by their position inside a forward step (steps start
at pair_indices_kernel), the same rule as tools/trace_launches.py: one kernel template runs every Linear."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# launches of gemm_split_ps_kernel inside one forward step of the L4 / mixed workload, in order (round 3: the FeedForward
# Linears of the three full layers are the ffn_fused_kernel launches, labelled by kernel name below)
GEMM_ORDER_L4_TWO_LAUNCH = ["gemm_patch", "gemm_qkv0_tab", "gemm_qkv0_tab", "gemm_qkv0_lc", "gemm_qkv0_lc", "gemm_qkv", "gemm_qkv",
                            "gemm_q_cls", "gemm_u_cls", "gemm_v_cls", "gemm_out_cls", "gemm_fc1_cls", "gemm_fc2_cls"]
# round 5: the QKV projections of the middle layers run inside qkv_attn_fused_kernel (labelled by kernel name below)
GEMM_ORDER_L4 = ["gemm_patch", "gemm_qkv0_tab", "gemm_qkv0_tab", "gemm_qkv0_lc", "gemm_qkv0_lc",
                 "gemm_q_cls", "gemm_u_cls", "gemm_v_cls", "gemm_out_cls", "gemm_fc1_cls", "gemm_fc2_cls"]


def load(d):
    """{counter: [per forward step: [(kernel name, value, duration_us), ... in launch order]]}"""
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    by_counter = collections.defaultdict(list)
    for r in rows:
        by_counter[r["Counter_Name"]].append(r)
    out = {}
    for c, rs in by_counter.items():
        rs.sort(key=lambda r: int(r["Start_Timestamp"]))
        steps, cur = [], None
        for r in rs:
            if "pair_indices_kernel" in r["Kernel_Name"]:
                cur = []
                steps.append(cur)
            if cur is not None:
                cur.append((r["Kernel_Name"], float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        n = max(len(s) for s in steps)
        out[c] = [s for s in steps if len(s) == n]
    return out


def per_label(steps):
    """mean counter value and duration per labelled launch: GEMM launches by position, other kernels by name"""
    acc = collections.defaultdict(list)
    for s in steps:
        gi = 0
        order = GEMM_ORDER_L4 if any("qkv_attn_fused_kernel" in name for name, _, _ in s) else GEMM_ORDER_L4_TWO_LAUNCH
        for name, v, us in s:
            if "gemm_split_ps_kernel" in name:
                label = order[gi] if gi < len(order) else "gemm_%d" % gi
                gi += 1
            else:
                label = name.replace("void ", "").replace("veto::", "").replace("(anonymous namespace)::", "")
                label = re.split(r"[<(]", label)[0].strip() or name[:40]
                if label == "ffn_fused_kernel":  # the names bench.py's per-kernel timers use: MODE 0 FeedForward, MODE 1 out projection
                    label = {"0": "ffn_fused", "1": "out_ln_fused", "2": "layer_tail_fused"}[re.search(r"ffn_fused_kernel<(\d)", name).group(1)]
                if label == "qkv_attn_fused_kernel":
                    label = "qkv_attn_fused"
                if "attention_mfma_kernel" in name:          # the table form of layer 0 is its own instantiation
                    label += "_tab" if re.search(r"attention_mfma_kernel<\d+, *(true|1)", name) else ""
            acc[label].append((v, us))
    return {k: (sum(x for x, _ in v) / len(v), sum(u for _, u in v) / len(v), len(v) // max(len(steps), 1)) for k, v in acc.items()}


def main():
    tag, dfetch, dwrite, dmfma = sys.argv[1:5]
    workload = [12, 36, 4, 8, "mixed"]
    if "--workload" in sys.argv:
        i = sys.argv.index("--workload")
        workload = [int(x) for x in sys.argv[i + 1:i + 5]] + [sys.argv[i + 5]]
    fetch = per_label(load(dfetch)["FETCH_SIZE"])
    write = per_label(load(dwrite)["WRITE_SIZE"])
    m = load(dmfma)
    busy, gui = per_label(m["SQ_VALU_MFMA_BUSY_CYCLES"]), per_label(m["GRBM_GUI_ACTIVE"])
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, (0, 0, 0)), write.get(k, (0, 0, 0))
        kernels[k] = {"FETCH_SIZE_KiB": round(f[0], 1), "WRITE_SIZE_KiB": round(w[0], 1),
                      "hbm_bytes_per_launch": int(2 * f[0] * 1024 + w[0] * 1024), "launches_per_step": f[2] or w[2]}
        if k in busy and gui.get(k, (0,))[0] > 0:
            # SQ_VALU_MFMA_BUSY_CYCLES sums the 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs' clocks over the dispatch
            kernels[k]["mfma_busy"] = round(busy[k][0] / 1024.0 / (gui[k][0] / 8.0), 4)
    att = {k: v for k, v in kernels.items() if "attention" in k or "attn" in k}
    doc = {"_comment": "per-launch HBM bytes (2 * FETCH_SIZE + WRITE_SIZE, KiB -> bytes; gfx950 FETCH correction of the micro-architecture "
                       "guide) and MFMA-busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / kernel cycles) from the rocprofv3 --pmc passes "
                       "of profile tag %s (profiles/%s_pmc_*.txt); written by tools/pmc_traffic.py" % (tag, tag),
           "tag": tag, "workload": workload, "kernels": kernels,
           "attention": {"note": "MFMA utilisation of the attention contractions (north_star): busy fraction of the attention kernels, and of "
                                 "the dominant GEMM for comparison", "kernels": {k: v.get("mfma_busy") for k, v in att.items()},
                         "gemm_qkv_mfma_busy": kernels.get("gemm_qkv", {}).get("mfma_busy"),
                         "layer_tail_mfma_busy": kernels.get("layer_tail_fused", {}).get("mfma_busy")}}
    out_name = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else "traffic.json"      # (other workloads: their own file)
    json.dump(doc, open(os.path.join(ROOT, "profiles", out_name), "w"), indent=1)
    for k, v in kernels.items():
        print("%-34s x%d  hbm %8.1f MB  mfma_busy %s" % (k, v["launches_per_step"], v["hbm_bytes_per_launch"] / 1e6, v.get("mfma_busy")))


if __name__ == "__main__":
    main()
