#!/usr/bin/env python3
"""Benchmark of the VETO pairwise relation-prediction hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no torch.distributed environment launches the N ranks itself
(a child `python -m torch.distributed.run`, started BEFORE the parent touches a GPU; the parent only relays
rank 0's JSON line and the exit code).  The line carries `ranks_seen` (the RCCL world size) and every
rank's device ordinal.

One step = one eval forward of the predictor (through the C ABI) over one resident synthetic batch:
BASELINE.json configs[1] -- 12 images x 36 objects = 15120 ordered pairs, d=576 tokens, 8 heads,
4 layers, 51 predicates.  Weak scaling: every rank processes its own 12-image batch; for N > 1 each
step ends with the RCCL all-gather of the [P, 51] logits (eval aggregation).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0
# precision modes of the library (include/veto_amd.h): what the Linears compute in, and how many bf16-rate MFMA
# passes one algorithmic FLOP costs in each
DTYPE_OF = {"precise": "bf16x3 (split-bf16 MFMA, fp32 accumulate)",
            "fast": "fp16 single pass (the mixed mode's launches with the correction stages of the fused token-row launches skipped)",
            "mixed": "fp16 + e4m3 correction terms (fp16 MFMA main product, e4m3 K=128 MFMA cross terms, fp32 accumulate)"}
MFMA_PASSES = {"precise": 3.0, "fast": 1.0, "mixed": 2.0}
DEFAULT_PRECISION = "mixed"


def load_traffic(name, workload):
    """profiles/<name> (written by tools/pmc_traffic.py from the --pmc passes of one profile tag) if it describes `workload`, else None."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", name)))
        return tj if tj.get("workload") == workload and "kernels" in tj and "tag" in tj else None
    except (OSError, ValueError):
        return None


def flops_per_pair(layers, heads):
    """Reference-equivalent dense FLOPs (2*M*N*K) per pair, SURVEY.md section 8(d)."""
    patch = 16 * 2048 * 576 * 2
    loc, cls = 256 * 576 * 2, 400 * 576 * 2
    qkv = 19 * 576 * 1728 * 2
    att = 2 * (19 * 19 * 576 * 2)
    out = 19 * 576 * 576 * 2
    mlp = 2 * (19 * 576 * 1152 * 2)
    head = 576 * 51 * 2
    return patch + loc + cls + layers * (qkv + att + out + mlp) + head


def count_gpus_sysfs():
    """GPUs of this node from the KFD topology (nodes with simd_count > 0), WITHOUT touching the HIP runtime: the parent of a
    multi-rank launch must not open the device (a child started from a process that holds it is the exec this pool forbids).
    None when the topology is not readable (then the ranks themselves report what they find)."""
    import glob
    if not os.path.isdir("/sys/class/kfd/kfd/topology/nodes"):
        return 0            # no KFD driver on this host: no GPU
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for path in nodes:
        try:
            for line in open(path):
                parts = line.split()
                if len(parts) == 2 and parts[0] == "simd_count" and int(parts[1]) > 0:
                    n += 1
        except OSError:
            return None
    return n


def self_launch(args):
    """--gpus N > 1 outside torch.distributed.run: start the N ranks as a child process group (subprocess: fresh children, never
    an exec of this process) and relay.  The parent makes NO torch.cuda call: the GPU count comes from sysfs."""
    dry = os.environ.get("VETO_BENCH_DRYRUN") == "1"
    have = count_gpus_sysfs()
    if not dry and have is not None and have < args.gpus:
        print("bench.py --gpus %d: this node exposes %d GPU(s); nothing was run" % (args.gpus, have), file=sys.stderr)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    result = [ln for ln in lines if ln.startswith('{"metric"')]
    for ln in lines:
        if ln not in result:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or not result:
        print("bench.py: the %d-rank child exited with code %d%s" % (args.gpus, proc.returncode, "" if result else " and printed no result line"),
              file=sys.stderr)
        return proc.returncode or 1
    print(result[-1], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--images", type=int, default=12)
    ap.add_argument("--objs", type=int, default=36)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--precision", default=DEFAULT_PRECISION, choices=sorted(DTYPE_OF))
    ap.add_argument("--chunk", type=int, default=0, help="VETO_AMD.MAX_CHUNK_PAIRS (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra L6/H6 and other-precision lines (N = 1 only)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d" % (args.gpus, world, args.gpus))
    dry = os.environ.get("VETO_BENCH_DRYRUN") == "1"   # CPU test of the launch / rendezvous / gather plumbing: gloo, no model
    if not dry:
        assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    if not dry:
        torch.cuda.set_device(dev)
    dist = None
    force_dist = os.environ.get("VETO_BENCH_FORCE_DIST") == "1"  # exercise the RCCL path with one rank (testing)
    if world > 1 or force_dist:
        import torch.distributed as dist
        if force_dist and "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from veto_amd import distributed as vdist

    if dry:
        n_pairs = args.images * args.objs * (args.objs - 1)
        fake = torch.full((n_pairs, 51), float(rank))

        def step():
            return vdist.all_gather_logits(fake, equal_counts=True, force=force_dist) if dist is not None else fake

        def fence():
            if dist is not None:
                dist.barrier()
    else:
        from veto_amd import synth, testing
        from veto_amd.pairs import prepare_test_pairs
        sd = synth.predictor_state_dict(0, layers=args.layers)
        model = testing.make_predictor(testing.make_config(args.layers, args.heads, precision=args.precision,
                                                           max_chunk_pairs=args.chunk), sd, dev)
        batch = synth.synthetic_batch(7 + rank, args.images, args.objs)
        props = testing.make_proposals(batch, "predcls", dev)
        pairs = prepare_test_pairs(dev, props)
        rgb = torch.from_numpy(batch["roi_features"]).to(dev)
        dep = torch.from_numpy(batch["roi_depth_features"]).to(dev)
        n_pairs = sum(int(p.shape[0]) for p in pairs)

        def step():
            with torch.no_grad():
                out = model(props, pairs, None, None, roi_features=rgb, roi_depth_features=dep)
            logits = torch.cat(list(out[1]), 0) if dist is not None else out[1]
            if dist is not None:
                logits = vdist.all_gather_logits(logits, equal_counts=True, force=force_dist)
            return logits

        def fence():
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    fence()
    elapsed = time.perf_counter() - t0
    ranks_seen, devices = 1, [0 if dry else torch.cuda.current_device()]
    rank_elapsed = [elapsed]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allt = torch.empty(dist.get_world_size(), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allt, t)          # every rank's own clock: a straggler is visible in the line
        rank_elapsed = [float(x) for x in allt.tolist()]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ranks_seen = dist.get_world_size()
        d = torch.tensor([-1 if dry else torch.cuda.current_device()], dtype=torch.int64, device=dev)
        alld = torch.empty(ranks_seen, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(alld, d)
        devices = [int(x) for x in alld.tolist()]
        gathered_rows = int(last.shape[0])
        assert gathered_rows == n_pairs * ranks_seen, (gathered_rows, n_pairs, ranks_seen)

    if dry:
        if rank == 0:
            print(json.dumps({"metric": "DRY RUN of the launch path (gloo, CPU, no model): not a measurement", "value": 0.0,
                              "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": elapsed / args.steps * 1e3, "ranks_seen": ranks_seen, "devices": devices,
                              "per_rank_units_per_s": [round(n_pairs * args.steps / e, 1) for e in rank_elapsed],
                              "gathered_rows": int(last.shape[0]), "dry_run": True}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- per-kernel device time (hipEvents on the launch stream, inside the library) -------------
    eng = model._engine
    eng.profile_reset()
    eng.profile_enable(True)
    prof_steps = min(5, args.steps)
    for _ in range(prof_steps):
        with torch.no_grad():
            model(props, pairs, None, None, roi_features=rgb, roi_depth_features=dep)
    torch.cuda.synchronize(dev)
    prof = eng.profile()
    eng.profile_enable(False)

    if rank == 0:
        total_pairs = n_pairs * world
        ms_per_step = elapsed / args.steps * 1e3
        value = total_pairs * args.steps / elapsed
        kern = {k: dict(ms_per_step=v["total_ms"] / prof_steps, launches_per_step=v["launches"] // prof_steps,
                        avg_ms=v["total_ms"] / v["launches"],
                        tflops=(v["flops_per_launch"] / (v["total_ms"] / v["launches"] * 1e-3) / 1e12)
                        if v["flops_per_launch"] else None)
                for k, v in prof.items()}
        gemms = {k: v for k, v in kern.items() if k.startswith("gemm_") or k in ("ffn_fused", "out_ln_fused", "layer_tail_fused")}
        # the dominant kernel = the matrix launch with the longest duration (round 3: the layer-tail launch -- out projection +
        # residual + LayerNorm2 + FeedForward + residual + next LayerNorm1 of a full layer, 953 GF; before that the QKV projection);
        # `gemm_all` below is the rate over all matrix launches
        dom = max(gemms, key=lambda k: gemms[k]["avg_ms"])
        achieved = gemms[dom]["tflops"]
        gemm_ms = sum(v["ms_per_step"] for v in gemms.values())
        gemm_flops = sum(prof[k]["flops_per_launch"] * kern[k]["launches_per_step"] for k in gemms)
        # FLOPs the step actually executes (algorithmic 2*M*N*K of every launch, attention contractions included)
        exec_flops = sum(prof[k]["flops_per_launch"] * kern[k]["launches_per_step"] for k in kern if prof[k]["flops_per_launch"])
        f_ref = flops_per_pair(args.layers, args.heads)
        passes = MFMA_PASSES[args.precision]
        hbm_gbps = {k: round(prof[k]["bytes_per_launch"] / (kern[k]["avg_ms"] * 1e-3) / 1e9, 1)
                    for k in ("attention", "attention_cls", "layernorm") if k in prof and prof[k]["bytes_per_launch"]}
        gather_bytes = prof["assemble_tokens"]["bytes_per_launch"]      # bytes the kernel writes, as the library accounts them
        gather_gbps = gather_bytes / (kern["assemble_tokens"]["avg_ms"] * 1e-3) / 1e9
        # PMC-derived numbers for THIS workload (HBM bytes per launch, MFMA-busy fractions): counters cannot be collected inside a timed
        # run, so they come from the committed rocprofv3 --pmc passes of the profile tag named in `traffic_source` (tools/pmc_traffic.py)
        tj = load_traffic("traffic.json", [args.images, args.objs, args.layers, args.heads, args.precision])
        traffic = tj["kernels"].get(dom, {}).get("hbm_bytes_per_launch") if tj else None
        attention_pmc = tj.get("attention") if tj else None
        tj_kernels = tj["kernels"] if tj else None
        traffic_source = "profiles/%s_pmc_*.txt (rocprofv3 --pmc passes of profile tag %s)" % (tj["tag"], tj["tag"]) if tj else None
        res = {
            "metric": "relation-pairs/sec (PredCls, 36 obj/img)", "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_OF[args.precision], "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: synthetic PredCls, %d img x %d obj = %d pairs/GPU, d=576, "
                                   "%d heads, %d layers, 51 predicates" % (args.images, args.objs, n_pairs, args.heads, args.layers),
                       "images_per_gpu": args.images, "objects_per_image": args.objs, "pairs_per_gpu": n_pairs,
                       "layers": args.layers, "heads": args.heads, "precision": args.precision,
                       "parallelism": "image-sharded x%d, RCCL all-gather of logits" % world if world > 1 else "single GPU"},
            "ranks_seen": ranks_seen, "devices": devices,
            "per_rank_pairs_per_s": [round(n_pairs * args.steps / e, 1) for e in rank_elapsed],
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                         "note": "achieved = algorithmic 2*M*N*K of one launch / its mean hipEvent duration; this precision mode "
                                 "issues %.4g bf16-equivalent MFMA passes per algorithmic FLOP" % passes},
            # the four floors of the dominant launch (rates measured on MI355X in round 4): it runs above all of
            # them because an in-order wave overlaps its matrix, LDS-DMA, L2-miss and HBM streams only partly
            "floors": layer_tail_floors(n_pairs * 19, prof[dom], traffic, gemms[dom]["avg_ms"]) if dom == "layer_tail_fused" else None,
            # north_star: "HBM GB/s on the gather and MFMA utilisation on the attention GEMMs"
            "gather": {"bound": "hbm", "kernel": "assemble_tokens", "achieved": gather_gbps, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                       "frac": gather_gbps / PEAK_HBM_GBS,
                       "bytes": gather_bytes,
                       "gathered_bytes": float(n_pairs) * 18 * 2 * 576 * 4,
                       "moved_gbps": (gather_bytes + float(n_pairs) * 18 * 2 * 576 * 4) / (kern["assemble_tokens"]["avg_ms"] * 1e-3) / 1e9,
                       "note": "pair gather + token assembly: bytes = what the kernel writes (the [pairs, 19, 576] fp32 token rows plus, "
                               "with layer 0 in the per-object form, the row statistics and the split rows of two of the 19 tokens); "
                               "the per-object rows it gathers (gathered_bytes; moved_gbps counts them too) come from L2 / Infinity Cache"},
            "attention": attention_pmc,
            # the fused QKV projection + attention launch of the middle layers (round 5): algorithmic FLOPs of the projection and of the
            # attention contractions / its mean hipEvent duration, counted HBM bytes per launch from the same PMC passes
            "qkv_attn": ({"kernel": "qkv_attn_fused", "ms_per_launch": round(kern["qkv_attn_fused"]["avg_ms"], 4),
                          "achieved": kern["qkv_attn_fused"]["tflops"], "unit": "TFLOP/s", "peak": PEAK_BF16_TFLOPS,
                          "frac": kern["qkv_attn_fused"]["tflops"] / PEAK_BF16_TFLOPS,
                          "traffic": (tj_kernels or {}).get("qkv_attn_fused", {}).get("hbm_bytes_per_launch"),
                          "algorithmic_bytes": prof["qkv_attn_fused"]["bytes_per_launch"],
                          "mfma_busy": (tj_kernels or {}).get("qkv_attn_fused", {}).get("mfma_busy")}
                         if "qkv_attn_fused" in kern else None),
            "gemm_all": {"kernel": "all GEMM launches of a step", "ms_per_step": round(gemm_ms, 4),
                         "achieved": gemm_flops / (gemm_ms * 1e-3) / 1e12, "unit": "TFLOP/s",
                         "frac": gemm_flops / (gemm_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS},
            "hbm_kernels_gbps": hbm_gbps,
            "mfma_issued": {"kernel": dom, "passes_per_flop": passes, "issued_tflops": achieved * passes,
                            "frac_of_bf16_peak": achieved * passes / PEAK_BF16_TFLOPS,
                            "note": "matrix-pipe work actually issued by the dominant GEMM, in bf16-rate units (an e4m3 K=128 MFMA "
                                    "counts half of its FLOPs: it runs at twice the bf16 rate)"},
            "whole_path": {"executed_flops_per_pair": exec_flops / n_pairs, "tflops_executed": value * exec_flops / n_pairs / 1e12,
                           "frac_of_bf16_peak": value * exec_flops / n_pairs / 1e12 / PEAK_BF16_TFLOPS / world,
                           "ref_flops_per_pair": f_ref, "tflops_ref_equivalent": value * f_ref / 1e12,
                           "note": "frac uses the FLOPs the restructured path executes (SURVEY.md 8d); the reference-equivalent "
                                   "figure (what the unrestructured formulation would need for the same pairs) is the labelled extra"},
            "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms_per_step"])},
            "gemm_tflops": {k: round(v["tflops"], 1) for k, v in gemms.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            gl = list(last.split([int(p.shape[0]) for p in pairs])) if torch.is_tensor(last) else last
            res["cpu_baseline"], res["logit_max_abs_err"] = cpu_baseline(sd, args, batch, gl, pairs)
        if world == 1 and not args.no_extra:
            res["extra"] = extra_lines(args, dev, batch, sd)
        # RCCL writes its version banner through C stdio, which (when stdout is a file or pipe) only
        # drains at exit, i.e. AFTER a Python print: drain it first so that the JSON is the last line.
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def extra_lines(args, dev, batch, sd):
    """Context next to the headline, N = 1 only, short runs: (1) the architecture the reference ships
    (configs/VETO_final.yaml: 6 layers x 6 heads) on the same 12 x 36 batch; (2) the other precision modes on the
    headline workload with their logit error against the headline mode's CPU-checked logits."""
    from oracle import veto_oracle as vo
    from veto_amd import synth, testing
    from veto_amd.pairs import prepare_test_pairs
    out = {}
    props = testing.make_proposals(batch, "predcls", dev)
    pairs = prepare_test_pairs(dev, props)
    rgb = torch.from_numpy(batch["roi_features"]).to(dev)
    dep = torch.from_numpy(batch["roi_depth_features"]).to(dev)
    n_pairs = sum(int(p.shape[0]) for p in pairs)

    def timed(model, steps=30):
        with torch.no_grad():
            for _ in range(6):
                o = model(props, pairs, None, None, roi_features=rgb, roi_depth_features=dep)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                o = model(props, pairs, None, None, roi_features=rgb, roi_depth_features=dep)
            torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        return dt, torch.cat(list(o[1]), 0).cpu()

    # one image of the batch through the CPU oracle = the error yardstick of every line below
    n = args.objs
    img0 = {"num_objs": [n], "boxes": batch["boxes"][:n], "labels": batch["labels"][:n],
            "roi_features": batch["roi_features"][:n], "roi_depth_features": batch["roi_depth_features"][:n]}
    ppi = n * (n - 1)
    sd6 = synth.predictor_state_dict(0, layers=6)
    m6 = testing.make_predictor(testing.make_config(6, 6, precision=args.precision), sd6, dev)
    dt, lg = timed(m6)
    ref6, _, _ = vo.forward(sd6, vo.OracleConfig(layers=6, heads=6), img0)
    out["l6h6"] = {"workload": "the reference's shipped architecture (6 layers x 6 heads) on the same 12 x %d batch" % n,
                   "pairs_per_s": n_pairs / dt, "ms_per_step": dt * 1e3, "precision": args.precision,
                   "logit_max_abs_err_image0": float((lg[:ppi] - ref6).abs().max())}
    # its own roofline object: the dominant launch by the library's hipEvent timers, counted traffic from this architecture's own PMC passes
    eng6 = m6._engine
    eng6.profile_reset()
    eng6.profile_enable(True)
    with torch.no_grad():
        for _ in range(3):
            m6(props, pairs, None, None, roi_features=rgb, roi_depth_features=dep)
    torch.cuda.synchronize(dev)
    prof6 = eng6.profile()
    eng6.profile_enable(False)
    mat6 = {k: v for k, v in prof6.items() if (k.startswith("gemm_") or k in ("layer_tail_fused", "qkv_attn_fused")) and v["flops_per_launch"]}
    if mat6:
        dom6 = max(mat6, key=lambda k: mat6[k]["total_ms"] / mat6[k]["launches"])
        avg6 = mat6[dom6]["total_ms"] / mat6[dom6]["launches"]
        tf6 = mat6[dom6]["flops_per_launch"] / (avg6 * 1e-3) / 1e12
        tj6 = load_traffic("traffic_l6h6.json", [args.images, n, 6, 6, args.precision])
        out["l6h6"]["roofline"] = {"bound": "mfma", "kernel": dom6, "ms_per_launch": avg6, "launches_per_step": mat6[dom6]["launches"] // 3,
                                   "achieved": tf6, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": tf6 / PEAK_BF16_TFLOPS,
                                   "traffic": tj6["kernels"].get(dom6, {}).get("hbm_bytes_per_launch") if tj6 else None,
                                   "traffic_source": "profiles/%s_pmc_*.txt" % tj6["tag"] if tj6 else None}
        out["l6h6"]["kernels_ms_per_step"] = {k: round(v["total_ms"] / 3, 4) for k, v in sorted(prof6.items(), key=lambda kv: -kv[1]["total_ms"])[:8]}
    del m6, eng6, prof6      # (the engine holds the handle and its workspace)
    ref, _, _ = vo.forward(sd, vo.OracleConfig(layers=args.layers, heads=args.heads), img0)
    for prec in sorted(DTYPE_OF):
        if prec == args.precision:
            continue
        m = testing.make_predictor(testing.make_config(args.layers, args.heads, precision=prec), sd, dev)
        dt, lg = timed(m)
        out["precision_" + prec] = {"dtype": DTYPE_OF[prec], "pairs_per_s": n_pairs / dt, "ms_per_step": dt * 1e3,
                                    "logit_max_abs_err_image0": float((lg[:ppi] - ref).abs().max())}
        del m
    import gc
    gc.collect()
    out["train"] = train_line(args, dev, batch, sd)
    out["train_recompute"] = train_recompute_line()
    return out


def train_recompute_line():
    """The same training step with VETO_TRAIN_RECOMPUTE=1 (LayerNorm / GELU rows recomputed in the backward instead of kept; the knob is read
    once per process, hence a child process running tools/train_bench.py).  None when the child fails: context, not the headline."""
    import re
    import subprocess
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_bench.py"), "3"], env=dict(os.environ, VETO_TRAIN_RECOMPUTE="1"),
                           capture_output=True, text=True, timeout=300)
        m = re.search(r"training step: ([0-9.]+) ms \(forward\+loss ([0-9.]+) ms, backward ([0-9.]+) ms.*peak memory ([0-9.]+) GB", p.stdout)
        if not m:
            return None
        return {"workload": "the training step above with VETO_TRAIN_RECOMPUTE=1, in a child process (its peak memory is the child's own)",
                "ms_per_step": float(m.group(1)), "forward_ms": float(m.group(2)), "backward_ms": float(m.group(3)), "peak_memory_gb": float(m.group(4))}
    except (OSError, subprocess.SubprocessError, ValueError):
        return None


def train_line(args, dev, batch, sd, steps=3):
    """One training step of the same architecture on the same batch (BASELINE configs 3-5 are training configs): veto_forward_train +
    veto_ce_loss + veto_backward through autograd (every layer on all 19 tokens, activations kept, 3-term split-bf16 operands -- none of
    the inference path's fused kernels), SGD step outside the timed forward / backward spans.  peak_memory_gb is the process's peak
    (resident_before_gb of it were held before the training model was built); train_workspace_gb is the path's own workspace.  Reference: roi_relation_predictors.py:4129-4136, tools/relation_train_net.py:372-380."""
    from veto_amd import synth, testing
    from veto_amd.pairs import prepare_test_pairs
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    resident = torch.cuda.memory_allocated(dev)      # what the headline model of this process still holds (its inference workspace)
    model = testing.make_predictor(testing.make_config(args.layers, args.heads), sd, dev).train()
    props = testing.make_proposals(batch, "predcls", dev)
    pairs = prepare_test_pairs(dev, props)
    n = sum(int(p.shape[0]) for p in pairs)
    labels = torch.from_numpy(synth.integers(5, "bench.labels", (n,), 0, 51)).to(dev)
    rel_labels = list(labels.split([int(p.shape[0]) for p in pairs]))
    kw = dict(roi_features=torch.from_numpy(batch["roi_features"]).to(dev), roi_depth_features=torch.from_numpy(batch["roi_depth_features"]).to(dev))
    opt = torch.optim.SGD(model.parameters(), lr=1e-4)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_f = t_b = wall = 0.0
    loss = None
    for i in range(steps + 1):            # one warm-up step (allocates the cached 36 GB activation workspace)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        ev[0].record()
        loss = model(props, pairs, rel_labels, None, **kw)[2]["rel_loss"]
        ev[1].record()
        loss.backward()
        ev[2].record()
        opt.step()
        torch.cuda.synchronize(dev)
        if i > 0:
            wall += time.perf_counter() - t0
            t_f += ev[0].elapsed_time(ev[1])
            t_b += ev[1].elapsed_time(ev[2])
    peak = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    ws = model.__dict__.get("_train_ws")      # the cached activation + scratch workspace of veto_forward_train / veto_backward
    ws_gb = ws.numel() * ws.element_size() / 2 ** 30 if torch.is_tensor(ws) else None
    del model, opt, ws
    torch.cuda.empty_cache()
    return {"workload": "one training step (forward + weighted-CE loss + backward + SGD) of the headline architecture on the same 12 x %d batch, "
                        "dropout at the reference's rates" % args.objs, "steps": steps, "ms_per_step": wall / steps * 1e3,
            "pairs_per_s": n / (wall / steps), "forward_ms": t_f / steps, "backward_ms": t_b / steps, "loss": float(loss.detach()),
            "peak_memory_gb": round(peak, 1), "resident_before_gb": round(resident / 2 ** 30, 1),
            "train_workspace_gb": round(ws_gb, 1) if ws_gb is not None else None,
            "dtype": "bf16x3 (split-bf16 MFMA, fp32 accumulate)"}


def usable_cores():
    """CPUs this process may actually use: the cgroup quota when there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def layer_tail_floors(rows, rec, l2_miss_bytes, avg_ms):
    """Lower bounds (ms) of one layer-tail launch from the rates measured in round 4 (profiles/r04_ldsdma_scale.txt,
    profiles/r04_tail_variants_b.txt): matrix pipe at the 1.44 PFLOP/s (algorithmic, both precision terms) of the MFMA-only ablation;
    L2 -> LDS at 91 GB/s per CU x 256 for the 8.7 MB of operand stages a 128-row panel streams; the L2 misses (PMC) at the ~8 TB/s
    of the path behind the L2; the algorithmic HBM bytes at the ~5 TB/s a chip-wide store burst reaches."""
    panels = (rows + 127) // 128
    lds_bytes = panels * (576 * 576 + 2 * 576 * 1152) * 4.0 + panels * 128 * 576 * 4.0 * 7      # weights + activation panel x (1 + 6)
    out = {"matrix_ms": rec["flops_per_launch"] / 1.44e15 * 1e3, "l2_to_lds_ms": lds_bytes / (256 * 91e9) * 1e3,
           "l2_miss_ms": (l2_miss_bytes / 8e12 * 1e3) if l2_miss_bytes else None,
           "hbm_ms": rec["bytes_per_launch"] / 5e12 * 1e3, "measured_ms": avg_ms}
    tight = max(v for k, v in out.items() if v is not None and k != "measured_ms")
    out["frac_of_tightest_floor"] = tight / avg_ms
    return {k: (round(v, 4) if v is not None else None) for k, v in out.items()}


def cpu_baseline(sd, args, batch, gpu_logits, pairs):
    """Times the CPU oracle (a port of the reference formulation: materialised gathers, per-pair patch
    embedding, all layers on all 19 tokens, fp32 torch-CPU) image by image on the host cores until
    ~cpu-seconds have elapsed; also returns the max-abs logit difference GPU vs oracle on those images."""
    from oracle import veto_oracle as vo
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = vo.OracleConfig(layers=args.layers, heads=args.heads)
    n = args.objs

    def image(i):
        sl = slice(i * n, (i + 1) * n)
        return {"num_objs": [n], "boxes": batch["boxes"][sl], "labels": batch["labels"][sl],
                "roi_features": batch["roi_features"][sl], "roi_depth_features": batch["roi_depth_features"][sl]}

    vo.forward(sd, cfg, image(0))  # warm-up
    done, spent, err = 0, 0.0, 0.0
    gl = [g.cpu() for g in gpu_logits]
    while done < args.images and spent < args.cpu_seconds:
        t0 = time.perf_counter()
        ref, _, _ = vo.forward(sd, cfg, image(done))
        spent += time.perf_counter() - t0
        err = max(err, (gl[done] - ref).abs().max().item())
        done += 1
    ppi = n * (n - 1)
    return ({"value": done * ppi / spent, "unit": "pairs/s", "cores": cores, "kind": "port",
             "sample": "%d of the %d images of the same batch (%d pairs), fp32 torch-CPU oracle in the reference "
                       "formulation, %d threads, after 1 warm-up image" % (done, args.images, done * ppi, cores)},
            err)


if __name__ == "__main__":
    main()
