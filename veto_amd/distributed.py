"""Data-parallel evaluation over the GPUs of one node: images are sharded across ranks (one process
per GPU, as the reference's DistributedSampler does -- data/samplers/distributed.py:10-65) and the
per-rank predicate logits are combined with ONE tensor all-gather over RCCL/xGMI, replacing the
reference's pickled-BoxList double all_gather (utils/comm.py:48-96, engine/inference.py:49-54).

The same code runs on the `gloo` backend with CPU tensors (tests/test_distributed_gloo.py)."""
import torch
import torch.distributed as dist


def shard_images(num_images, rank=None, world=None):
    """Contiguous shard of image indices for this rank; every image goes to exactly one rank and
    shard sizes differ by at most one (IMS_PER_BATCH % num_gpus == 0 in the reference, data/build.py:189-191)."""
    world = dist.get_world_size() if world is None else world
    rank = dist.get_rank() if rank is None else rank
    base, rem = divmod(num_images, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def all_gather_logits(logits, equal_counts=False, group=None, force=False, max_rows=None):
    """logits: [P_r, C] on this rank -> [sum_r P_r, C], rank-major, on every rank.

    P_r may differ between ranks (images have different object counts): the row counts are gathered
    first and the payload is padded to the maximum, so the data exchange is a single
    all_gather_into_tensor (direct, one hop per peer on the fully connected xGMI mesh).

    Host synchronisation: none with `equal_counts`; with ragged counts the default form reads the gathered counts back (one
    small copy) to size and trim the result.  `max_rows` (an upper bound of P_r known to every rank, e.g. images per rank x
    MAX_PROPOSAL_PAIR) removes that read-back: the call then returns `(padded [world, max_rows, C], counts [world] on the
    device)` and never touches the host -- the form for an eval loop that keeps everything on the GPU (the evaluators take
    device tensors)."""
    if not dist.is_available() or not dist.is_initialized():
        return logits
    world = dist.get_world_size(group)
    if world == 1 and not force:   # force: run the collective anyway (single-rank test of the RCCL path)
        return logits
    logits = logits.contiguous()
    p, c = logits.shape
    if equal_counts:
        out = torch.empty((world * p, c), dtype=logits.dtype, device=logits.device)
        dist.all_gather_into_tensor(out, logits, group=group)
        return out
    counts = torch.empty(world, dtype=torch.int64, device=logits.device)
    dist.all_gather_into_tensor(counts, torch.tensor([p], dtype=torch.int64, device=logits.device), group=group)
    if max_rows is not None:
        if p > max_rows:
            raise ValueError("all_gather_logits: %d rows on this rank exceed max_rows=%d" % (p, max_rows))
        padded = torch.zeros((max_rows, c), dtype=logits.dtype, device=logits.device)
        padded[:p] = logits
        out = torch.empty((world * max_rows, c), dtype=logits.dtype, device=logits.device)
        dist.all_gather_into_tensor(out, padded, group=group)
        return out.view(world, max_rows, c), counts
    counts = counts.tolist()
    pmax = max(counts)
    padded = logits
    if p < pmax:
        padded = torch.zeros((pmax, c), dtype=logits.dtype, device=logits.device)
        padded[:p] = logits
    out = torch.empty((world * pmax, c), dtype=logits.dtype, device=logits.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * pmax: r * pmax + counts[r]] for r in range(world)], 0)


def all_reduce_gradients(params, group=None, average=True, bucket_bytes=256 << 20):
    """Data-parallel training exchange: sum (or average) the `.grad` of `params` over the ranks.

    The reference wraps the whole detector in DistributedDataParallel (tools/relation_train_net.py:372-380,
    find_unused_parameters=True); for the predictor alone that is one all-reduce of ~70 MB fp32 per step.  On the
    point-to-point xGMI mesh a ring all-reduce is bound by one link (~153 GB/s), so the gradients are packed into
    as few, as large messages as possible: one flat bucket for the whole predictor by default (bucket_bytes only
    bounds the staging copy), ~1 ms against a ~100 ms step -- nothing to overlap.

    Same contract as DDP with find_unused_parameters=True: a parameter that received a gradient on ANY rank ends up
    with the same (summed / averaged, absent ranks counting as zero) gradient on EVERY rank -- a rank that had none
    gets one allocated -- so replicas that start equal stay equal under any optimizer.  A parameter without a gradient
    on every rank (the reference's unused `obj_embed2` / `bbox_embed`) stays without one.  One "has a gradient" flag per
    parameter travels at the tail of the same bucket, so this costs no extra collective.
    Returns the number of collectives issued."""
    if not dist.is_available() or not dist.is_initialized():
        return 0
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    calls, i = 0, 0
    while i < len(params):
        j, nbytes = i, 0
        while j < len(params) and (j == i or nbytes + params[j].numel() * 4 <= bucket_bytes):
            nbytes += params[j].numel() * 4
            j += 1
        bucket = params[i:j]
        n = sum(p.numel() for p in bucket)
        flat = torch.zeros(n + len(bucket), dtype=torch.float32, device=bucket[0].device)
        off = 0
        for p in bucket:
            if p.grad is not None:
                flat[off:off + p.numel()] = p.grad.reshape(-1)
            off += p.numel()
        # the "has a gradient" flags of the bucket in ONE copy (not one tiny device kernel per parameter)
        flat[n:] = torch.tensor([0.0 if p.grad is None else 1.0 for p in bucket], dtype=torch.float32).to(flat.device)
        dist.all_reduce(flat, group=group)
        if average:
            flat[:n] /= world
        used = flat[n:].tolist()          # one small read-back per bucket; the step is ~100 ms
        off = 0
        for k, p in enumerate(bucket):
            if used[k] > 0:
                g = flat[off:off + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = g.to(p.dtype).clone()       # (a bf16 / fp16 parameter takes a gradient of its own dtype)
                else:
                    p.grad.copy_(g)
            off += p.numel()
        calls += 1
        i = j
    return calls
