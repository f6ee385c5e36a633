"""world_size-2 tests of the multi-GPU exchanges on the gloo backend (CPU): image sharding, the variable-length
all-gather of predicate logits (eval) and the flat-bucket gradient all-reduce (training), veto_amd/distributed.py."""
import os

import pytest
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from veto_amd import distributed as vdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        num_objs = [5, 3, 4]                      # 3 images -> rank 0 gets 2, rank 1 gets 1
        mine = vdist.shard_images(len(num_objs))
        # deterministic stand-in for the per-image logits: row value encodes (image, pair)
        rows = []
        for i in mine:
            p = num_objs[i] * (num_objs[i] - 1)
            rows.append(torch.arange(p, dtype=torch.float32)[:, None] + 1000.0 * i + torch.zeros(1, 51))
        local = torch.cat(rows)
        full = vdist.all_gather_logits(local)
        same = vdist.all_gather_logits(local[:6], equal_counts=True)
        padded, counts = vdist.all_gather_logits(local, max_rows=40)      # no read-back of the counts
        if rank == 0:
            torch.save({"mine": mine, "full": full, "same": same, "padded": padded, "counts": counts}, out)
    finally:
        dist.destroy_process_group()


def test_shard_and_all_gather_world2(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["mine"] == [0, 1]
    exp = torch.cat([torch.arange(n * (n - 1), dtype=torch.float32) + 1000.0 * i for i, n in enumerate([5, 3, 4])])
    assert r["full"].shape == (20 + 6 + 12, 51) and torch.equal(r["full"][:, 0], exp)
    assert r["same"].shape == (12, 51)
    assert r["padded"].shape == (2, 40, 51) and r["counts"].tolist() == [26, 12]
    assert torch.equal(torch.cat([r["padded"][k, :n] for k, n in enumerate(r["counts"].tolist())]), r["full"])


def _grad_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
        net[2].weight.requires_grad_(False)
        gen = torch.Generator().manual_seed(100 + rank)
        ps = list(net.parameters())
        for k, p in enumerate(ps):
            if p.requires_grad and k != 1:            # parameter 1 has no gradient on either rank ("unused")
                p.grad = torch.randn(p.shape, generator=gen)
        if rank == 1:
            ps[3].grad = None                         # ... and parameter 3 only on rank 0
        before = [None if p.grad is None else p.grad.clone() for p in ps]
        one = vdist.all_reduce_gradients(ps)
        after_one = [None if p.grad is None else p.grad.clone() for p in ps]
        for p, b in zip(ps, before):
            p.grad = None if b is None else b.clone()
        many = vdist.all_reduce_gradients(ps, bucket_bytes=64, average=False)
        torch.save({"before": before, "after_one": after_one, "after_many": [None if p.grad is None else p.grad for p in ps],
                    "calls": (one, many)}, out % rank)
    finally:
        dist.destroy_process_group()


def test_all_reduce_gradients_world2(tmp_path):
    out = str(tmp_path / "g%d.pt")
    mp.spawn(_grad_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert r0["calls"][0] == 1 and r0["calls"][1] > 1 and r0["calls"] == r1["calls"]
    for k in range(len(r0["before"])):
        a, b = r0["before"][k], r1["before"][k]
        if a is None and b is None:
            assert r0["after_one"][k] is None and r1["after_one"][k] is None
            continue
        total = (a if a is not None else 0) + (b if b is not None else 0)
        for r in (r0, r1):   # DDP find_unused_parameters semantics: EVERY rank ends with the reduced gradient
            assert torch.allclose(r["after_one"][k], total / 2) and torch.allclose(r["after_many"][k], total)


def _train_worker(rank, world, port, out):
    """Three optimizer steps of two replicas whose used-parameter sets differ per rank and per step
    (tools/relation_train_net.py:372-380 wraps the model with find_unused_parameters=True for exactly this)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.ModuleList([torch.nn.Linear(6, 6) for _ in range(4)])   # branch b used when (rank + step + b) % 3 != 0
        opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
        gen = torch.Generator().manual_seed(7 + rank)
        for step in range(3):
            opt.zero_grad(set_to_none=True)
            x = torch.randn(5, 6, generator=gen)
            y = sum(net[b](x).square().mean() for b in range(3) if (rank + step + b) % 3 != 0)   # net[3] is never used
            y.backward()
            vdist.all_reduce_gradients(net.parameters())
            opt.step()
        torch.save([p.detach().clone() for p in net.parameters()], out % rank)
    finally:
        dist.destroy_process_group()


def test_replicas_stay_identical_with_rank_dependent_unused_parameters(tmp_path):
    out = str(tmp_path / "t%d.pt")
    mp.spawn(_train_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    p0, p1 = torch.load(out % 0), torch.load(out % 1)
    torch.manual_seed(0)
    init = [p.detach().clone() for p in torch.nn.ModuleList([torch.nn.Linear(6, 6) for _ in range(4)]).parameters()]
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)
    assert not torch.equal(p0[0], init[0])                                   # the used branches moved ...
    assert torch.equal(p0[6], init[6]) and torch.equal(p0[7], init[7])       # ... the never-used one did not


@pytest.mark.parametrize("ranks", [2, 8])
def test_bench_self_launch_dry_run(tmp_path, ranks):
    """`python bench.py --gpus N` outside torch.distributed.run starts its own N ranks (gloo / CPU dry run of the launch,
    rendezvous and all-gather plumbing: no model, no GPU) and relays ONE JSON line that shows every rank took part, with
    every rank's own rate (a straggler would be visible).  N = 8 is the node size the scaling bench runs at."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VETO_BENCH_DRYRUN="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["ranks_seen"] == ranks and d["dry_run"] is True
    assert d["gathered_rows"] == ranks * 12 * 36 * 35
    assert len(d["per_rank_units_per_s"]) == ranks and all(v > 0 for v in d["per_rank_units_per_s"])
    if ranks != 2:
        return
    # without the dry-run switch and without two GPUs the parent refuses cleanly before touching a device
    env.pop("VETO_BENCH_DRYRUN")
    import bench
    if (bench.count_gpus_sysfs() or 0) < 2:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, env=env, timeout=300)
        assert r.returncode == 2 and "nothing was run" in r.stderr and r.stdout.strip() == ""


def test_shard_images_partitions_exactly():
    for n in (1, 7, 12, 96):
        for world in (1, 2, 4, 8):
            got = [vdist.shard_images(n, r, world) for r in range(world)]
            assert sorted(sum(got, [])) == list(range(n))
            assert max(map(len, got)) - min(map(len, got)) <= 1


def test_single_process_is_identity():
    x = torch.randn(5, 51)
    assert vdist.all_gather_logits(x) is x
