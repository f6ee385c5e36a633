"""world_size-2 tests of the multi-GPU exchanges on the gloo backend (CPU): image sharding, the variable-length
all-gather of predicate logits (eval) and the flat-bucket gradient all-reduce (training), veto_amd/distributed.py."""
import os
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
        if rank == 0:
            torch.save({"mine": mine, "full": full, "same": same}, out)
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
        for r, mine in ((r0, a), (r1, b)):
            if mine is None:
                assert r["after_one"][k] is None        # a rank that had no gradient keeps none
            else:
                assert torch.allclose(r["after_one"][k], total / 2) and torch.allclose(r["after_many"][k], total)


def test_shard_images_partitions_exactly():
    for n in (1, 7, 12, 96):
        for world in (1, 2, 4, 8):
            got = [vdist.shard_images(n, r, world) for r in range(world)]
            assert sorted(sum(got, [])) == list(range(n))
            assert max(map(len, got)) - min(map(len, got)) <= 1


def test_single_process_is_identity():
    x = torch.randn(5, 51)
    assert vdist.all_gather_logits(x) is x
