#!/usr/bin/env python3
"""Benchmark of the VETO pairwise relation-prediction hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one eval forward of the predictor (through the C ABI) over one resident synthetic batch:
BASELINE.json configs[1] -- 12 images x 36 objects = 15120 ordered pairs, d=576 tokens, 8 heads,
4 layers, 51 predicates.  Weak scaling: every rank processes its own 12-image batch; for N > 1 each
step ends with the RCCL all-gather of the [P, 51] logits (eval aggregation).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--images", type=int, default=12)
    ap.add_argument("--objs", type=int, default=36)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--precision", default="precise", choices=["precise", "fast"])
    ap.add_argument("--chunk", type=int, default=0, help="VETO_AMD.MAX_CHUNK_PAIRS (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    force_dist = os.environ.get("VETO_BENCH_FORCE_DIST") == "1"  # exercise the RCCL path with one rank (testing)
    if world > 1 or force_dist:
        import torch.distributed as dist
        if force_dist and "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        dist.init_process_group("nccl", device_id=dev)

    from veto_amd import distributed as vdist
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
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

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
        gemms = {k: v for k, v in kern.items() if k.startswith("gemm_")}
        # the dominant kernel = the GEMM launch with the longest duration (the QKV projection of a full layer; the same
        # gemm_split_ps_kernel runs every Linear, `gemm_all` below is the rate over all of its launches in a step)
        dom = max(gemms, key=lambda k: gemms[k]["avg_ms"])
        achieved = gemms[dom]["tflops"]
        gemm_ms = sum(v["ms_per_step"] for v in gemms.values())
        gemm_flops = sum(prof[k]["flops_per_launch"] * kern[k]["launches_per_step"] for k in gemms)
        f_ref = flops_per_pair(args.layers, args.heads)
        passes = 3 if args.precision == "precise" else 1
        hbm_gbps = {k: round(prof[k]["bytes_per_launch"] / (kern[k]["avg_ms"] * 1e-3) / 1e9, 1)
                    for k in ("attention", "attention_cls", "layernorm") if k in prof and prof[k]["bytes_per_launch"]}
        gather_bytes = prof["assemble_tokens"]["bytes_per_launch"]      # bytes the kernel writes, as the library accounts them
        gather_gbps = gather_bytes / (kern["assemble_tokens"]["avg_ms"] * 1e-3) / 1e9
        traffic = None
        try:  # PMC-derived HBM bytes per launch of the dominant kernel at THIS workload (profiles/r01_traffic.json)
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if dom in tj and (args.images, args.objs, args.layers, args.heads, args.precision) == (12, 36, 4, 8, "precise"):
                traffic = tj[dom]["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        res = {
            "metric": "relation-pairs/sec (PredCls, 36 obj/img)", "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x3 (split-bf16 MFMA, fp32 accumulate)" if args.precision == "precise" else "bf16",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: synthetic PredCls, %d img x %d obj = %d pairs/GPU, d=576, "
                                   "%d heads, %d layers, 51 predicates" % (args.images, args.objs, n_pairs, args.heads, args.layers),
                       "images_per_gpu": args.images, "objects_per_image": args.objs, "pairs_per_gpu": n_pairs,
                       "layers": args.layers, "heads": args.heads, "precision": args.precision,
                       "parallelism": "image-sharded x%d, RCCL all-gather of logits" % world if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                         "note": "achieved = algorithmic 2*M*N*K of one launch / its mean hipEvent duration; the "
                                 "precise mode issues 3 bf16 MFMA passes per algorithmic FLOP"},
            # north_star: "HBM GB/s on the gather and MFMA utilisation on the attention GEMMs"
            "gather": {"bound": "hbm", "kernel": "assemble_tokens", "achieved": gather_gbps, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                       "frac": gather_gbps / PEAK_HBM_GBS,
                       "bytes": gather_bytes,
                       "gathered_bytes": float(n_pairs) * 18 * 2 * 576 * 4,
                       "moved_gbps": (gather_bytes + float(n_pairs) * 18 * 2 * 576 * 4) / (kern["assemble_tokens"]["avg_ms"] * 1e-3) / 1e9,
                       "note": "pair gather + token assembly: bytes = what the kernel writes (the [pairs, 19, 576] fp32 token rows plus, "
                               "with layer 0 in the per-object form, the row statistics and the split rows of two of the 19 tokens; "
                               "otherwise plus the LayerNorm'ed split copy of every row); the per-object rows it gathers (gathered_bytes; moved_gbps counts them too) come from L2 / Infinity Cache. "
                               "Pure write streams top out near 3.5 TB/s on this part (DESIGN.md section 7)"},
            "gemm_all": {"kernel": "gemm_split_ps_kernel, all launches of a step", "ms_per_step": round(gemm_ms, 4),
                         "achieved": gemm_flops / (gemm_ms * 1e-3) / 1e12, "unit": "TFLOP/s",
                         "frac": gemm_flops / (gemm_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS},
            "hbm_kernels_gbps": hbm_gbps,
            "mfma_issued": {"kernel": dom, "passes_per_flop": passes, "issued_tflops": achieved * passes,
                            "frac_of_bf16_peak": achieved * passes / PEAK_BF16_TFLOPS,
                            "note": "matrix-pipe rate actually issued by the dominant GEMM; SQ_VALU_MFMA_BUSY_CYCLES from the PMC pass "
                                    "is in profiles/ (busy fraction of kernel cycles); the vendor bf16 GEMM (hipBLASLt) given the same "
                                    "issued flops at this shape reaches 1.12 PFLOP/s on the same part (tools/vendor_gemm_probe.py, DESIGN.md section 7)"},
            "whole_path": {"ref_flops_per_pair": f_ref, "tflops_ref_equivalent": value * f_ref / 1e12,
                           "frac_of_bf16_peak": value * f_ref / 1e12 / PEAK_BF16_TFLOPS / world},
            "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms_per_step"])},
            "gemm_tflops": {k: round(v["tflops"], 1) for k, v in gemms.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            gl = list(last.split([int(p.shape[0]) for p in pairs])) if torch.is_tensor(last) else last
            res["cpu_baseline"], res["logit_max_abs_err"] = cpu_baseline(sd, args, batch, gl, pairs)
        # RCCL writes its version banner through C stdio, which (when stdout is a file or pipe) only
        # drains at exit, i.e. AFTER a Python print: drain it first so that the JSON is the last line.
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


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
