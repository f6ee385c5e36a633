"""Per-launch view of a rocprofv3 kernel trace: the split-GEMM kernel runs every Linear of the step, so its `--stats`
average mixes shapes.  This prints, for each GEMM dispatch position inside a forward step (steps start at
pair_indices_kernel), the instantiation and the mean duration over the steps of the trace -- the rocprof-side numbers
that bench.py's per-launch hipEvent times (kernels_ms_per_step / launches) are to be compared with.
usage: python tools/trace_launches.py <kernel_trace.csv>   (launch order = the run_gemm calls of veto_forward, veto_abi.hip)"""
import csv
import re
import sys
from collections import defaultdict

ORDER_L4_UNFUSED = ["gemm_patch", "gemm_qkv0_tab (S)", "gemm_qkv0_tab (O)", "gemm_qkv0_lc (t17)", "gemm_qkv0_lc (t18)", "gemm_out", "gemm_fc1",
                    "gemm_fc2", "gemm_qkv", "gemm_out", "gemm_fc1", "gemm_fc2", "gemm_qkv", "gemm_out", "gemm_fc1", "gemm_fc2", "gemm_u_cls",
                    "gemm_out_cls", "gemm_fc1_cls", "gemm_fc2_cls"]
# round 3 (VETO_MIXED): the FeedForward Linears of the three full layers run in ffn_fused_kernel (listed separately below)
ORDER_L4_PRODUCTS = ["gemm_patch", "gemm_qkv0_tab (S)", "gemm_qkv0_tab (O)", "gemm_qkv0_lc (t17)", "gemm_qkv0_lc (t18)", "gemm_qkv", "gemm_qkv",
                     "gemm_u_cls", "gemm_out_cls", "gemm_fc1_cls", "gemm_fc2_cls"]
# ... and the folded last layer runs its two products as four block-structured GEMMs (VETO_FOLD_BLOCKS=0: the list above)
ORDER_L4_TWO_LAUNCH = ["gemm_patch", "gemm_qkv0_tab (S)", "gemm_qkv0_tab (O)", "gemm_qkv0_lc (t17)", "gemm_qkv0_lc (t18)", "gemm_qkv", "gemm_qkv",
                       "gemm_q_cls", "gemm_u_cls", "gemm_v_cls", "gemm_out_cls", "gemm_fc1_cls", "gemm_fc2_cls"]
# round 5: the QKV projections of the middle layers run inside qkv_attn_fused_kernel (listed separately below; VETO_QKV_ATTN_FUSED=0: the list above)
ORDER_L4 = ["gemm_patch", "gemm_qkv0_tab (S)", "gemm_qkv0_tab (O)", "gemm_qkv0_lc (t17)", "gemm_qkv0_lc (t18)",
            "gemm_q_cls", "gemm_u_cls", "gemm_v_cls", "gemm_out_cls", "gemm_fc1_cls", "gemm_fc2_cls"]


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps, cur, ffn, outp, tail, qa = [], None, [], [], [], []
    for r in rows:
        name = r["Kernel_Name"]
        if "pair_indices_kernel" in name:
            cur = []
            steps.append(cur)
        elif cur is not None and "gemm_split_ps_kernel" in name:
            inst = re.search(r"gemm_split_ps_kernel<([^>]*)>", name).group(1)
            cur.append((inst, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        elif cur is not None and "qkv_attn_fused_kernel" in name:
            qa.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        elif cur is not None and "ffn_fused_kernel<0" in name:
            ffn.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        elif cur is not None and "ffn_fused_kernel<1" in name:
            outp.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        elif cur is not None and "ffn_fused_kernel<2" in name:
            tail.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    n = max(len(s) for s in steps)
    steps = [s for s in steps if len(s) == n]
    acc = defaultdict(list)
    for s in steps:
        for i, (inst, us) in enumerate(s):
            acc[(i, inst)].append(us)
    order = next((o for o in (ORDER_L4, ORDER_L4_TWO_LAUNCH, ORDER_L4_PRODUCTS, ORDER_L4_UNFUSED) if n == len(o) and (o is not ORDER_L4_PRODUCTS or not qa)), None)
    print("%d forward steps, %d GEMM launches each (4-layer order: %s)" % (len(steps), n, "yes" if order else "n/a"))
    if qa:
        print("  qkv_attn_fused_kernel (QKV projection + attention of a middle layer): %d launches, mean %8.1f us  (min %8.1f, max %8.1f)" % (len(qa), sum(qa) / len(qa), min(qa), max(qa)))
    if ffn:
        print("  ffn_fused_kernel<0> (FeedForward): %d launches, mean %8.1f us  (min %8.1f, max %8.1f)" % (len(ffn), sum(ffn) / len(ffn), min(ffn), max(ffn)))
    if tail:
        print("  ffn_fused_kernel<2> (layer tail: out projection + LayerNorm2 + FeedForward + LayerNorm1): %d launches, mean %8.1f us  (min %8.1f, max %8.1f)" % (len(tail), sum(tail) / len(tail), min(tail), max(tail)))
    if outp:
        print("  ffn_fused_kernel<1> (out projection + LayerNorm2): %d launches, mean %8.1f us  (min %8.1f, max %8.1f)" % (len(outp), sum(outp) / len(outp), min(outp), max(outp)))
    for (i, inst), v in sorted(acc.items()):
        label = order[i] if order else ""
        print("  launch %2d  gemm_split_ps_kernel<%s>  mean %8.1f us  (min %8.1f, max %8.1f)  %s" % (i, inst, sum(v) / len(v), min(v), max(v), label))


if __name__ == "__main__":
    main(sys.argv[1])
