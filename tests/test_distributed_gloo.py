"""world_size-2 test of the eval aggregation on the gloo backend (CPU): image sharding and the
variable-length all-gather of predicate logits (veto_amd/distributed.py)."""
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


def test_shard_images_partitions_exactly():
    for n in (1, 7, 12, 96):
        for world in (1, 2, 4, 8):
            got = [vdist.shard_images(n, r, world) for r in range(world)]
            assert sorted(sum(got, [])) == list(range(n))
            assert max(map(len, got)) - min(map(len, got)) <= 1


def test_single_process_is_identity():
    x = torch.randn(5, 51)
    assert vdist.all_gather_logits(x) is x
