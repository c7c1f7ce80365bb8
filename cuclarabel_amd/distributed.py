"""Multi-GPU sharding for independent problems.

A single factorisation does not shard (elimination-tree dependencies would put xGMI on the
critical path), so N GPUs are used the way BASELINE.json's north_star asks: independent
problems / block-diagonal batches are dealt to ranks -- one process per GPU, no collective in
the factor/solve path -- and only a tiny per-problem record (status, IR rounds, timings) is
gathered at the end with `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def assign_problems(n_problems, world_size, rank, weights=None):
    """Static assignment of problem indices to `rank`.  Without weights: round-robin.  With
    weights (e.g. nnz(L) per problem): longest-processing-time greedy, deterministic."""
    if weights is None:
        return list(range(rank, n_problems, world_size))
    order = sorted(range(n_problems), key=lambda j: (-float(weights[j]), j))
    load = [0.0] * world_size
    mine = []
    for j in order:
        r = min(range(world_size), key=lambda q: (load[q], q))
        load[r] += float(weights[j])
        if r == rank:
            mine.append(j)
    return sorted(mine)


def reduce_max(value, device=None):
    """MAX over ranks of a scalar (the timed region's wall time: the job ends with its slowest rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_records(local_records, n_problems, width, device=None):
    """All-gather of fixed-width per-problem records (one row per problem this rank solved:
    [problem index, status, ir_rounds, t_factor_ms, t_solve_ms, ...]).  Returns an
    (n_problems, width) tensor ordered by problem index on every rank.  Message size is
    O(n_problems * width * 8 B): latency-bound, topology irrelevant."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    # rows per rank = the LARGEST share any rank holds: a weighted (LPT) assignment may give one rank more than
    # ceil(n_problems / world) problems, so the size is agreed on with one MAX all-reduce instead of assumed
    per_rank = len(local_records)
    if world > 1:
        cnt = torch.tensor([per_rank], dtype=torch.int64, device=device)
        dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
        per_rank = int(cnt.item())
    per_rank = max(per_rank, 1)
    buf = torch.full((per_rank, width), float("nan"), dtype=torch.float64, device=device)
    for k, rec in enumerate(local_records):
        buf[k, :] = torch.as_tensor(rec, dtype=torch.float64)
    if world == 1:
        allbuf = [buf]
    else:
        allbuf = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(allbuf, buf)
    out = torch.full((n_problems, width), float("nan"), dtype=torch.float64)
    for b in allbuf:
        b = b.cpu()
        for row in b:
            if not torch.isnan(row[0]):
                out[int(row[0].item())] = row
    return out


# ---- right-hand-side columns of ONE factorisation dealt to ranks ----
def shard_columns(n_columns, world_size, rank):
    """Round-robin: rank r solves columns r, r + world, r + 2 world, ..."""
    return list(range(rank, n_columns, world_size))


def gather_columns(local, n_columns, device=None):
    """All-gather of the per-rank solution blocks.  `local`: (len(shard_columns(...)), length)
    tensor, row q = this rank's q-th column.  Returns (n_columns, length), row j = column j, on
    every rank.  One collective of n_columns*length/world doubles per rank -- the only exchange of
    the batched-RHS mode (ring all-gather over xGMI: per-link bound, so it is issued once per
    batch, not per column)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    length = local.shape[1]
    if world == 1:
        return local[:n_columns]
    per = (n_columns + world - 1) // world
    send = torch.zeros(per, length, dtype=local.dtype, device=local.device if device is None else device)
    km = len(shard_columns(n_columns, world, rank))
    send[:km] = local[:km]
    out = torch.empty(world * per, length, dtype=local.dtype, device=send.device)
    dist.all_gather_into_tensor(out, send)
    # row r*per + q holds column q*world + r
    out = out.view(world, per, length).transpose(0, 1).reshape(per * world, length)
    return out[:n_columns]
