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


class BlockedColumnGather:
    """The same exchange in BLOCKS of columns, asynchronously: while block i + 1 is still being solved, block i's
    solutions are already on the links (RCCL runs the collective on its own stream; `async_op=True` orders it behind the
    work enqueued so far and returns).  A step is then bound by max(solve, exchange) instead of their sum -- at 8 ranks
    and 512 columns of cfg2 the (x, z) solutions are 1.2 GB per step, about as long on xGMI as the solves themselves.

    Columns are dealt round-robin (shard_columns): local column q of rank r is global column q * world + r, so the
    local block [q0, q1) of all ranks is the contiguous global range [q0 * world, q1 * world).  `parts` are the lengths
    of the vectors gathered per column (e.g. (n, m) for the x and z parts of a KKT solution), one collective each.
    """

    def __init__(self, n_columns, parts, block, device=None, dtype=torch.float64):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.n_columns, self.parts, self.block = int(n_columns), tuple(int(p) for p in parts), max(1, int(block))
        self.per = (self.n_columns + self.world - 1) // self.world           # local columns incl. padding
        self.km = len(shard_columns(self.n_columns, self.world, self.rank))
        self.out = [torch.zeros(self.per * self.world, p, dtype=dtype, device=device) for p in self.parts]
        self._pending = []

    def blocks(self):
        """Local column ranges [q0, q1) in the order they are to be solved and posted (the same on every rank: ranks
        with fewer columns post zero padding for the ones they do not hold)."""
        return [(q0, min(q0 + self.block, self.per)) for q0 in range(0, self.per, self.block)]

    def _undeal(self, entry):
        work, tmp, _send, p, q0, q1 = entry
        nb = q1 - q0
        self.out[p][q0 * self.world:q1 * self.world] = \
            tmp.view(self.world, nb, self.parts[p]).transpose(0, 1).reshape(nb * self.world, self.parts[p])

    def post(self, q0, q1, locals_, copy=True):
        """locals_[p]: (>= q1 - q0 rows, parts[p]) tensor whose row j is this rank's local column q0 + j (rows beyond
        this rank's share are ignored).  Starts the exchange of the block and returns at once.
        copy=True (default): the rows are copied into a buffer this object owns before the collective is started, so
        the caller may overwrite its buffer at once (the natural pattern: one block buffer re-used for the next
        solve).  copy=False sends the caller's rows in place: they must then stay untouched until finish() -- nothing
        orders a later write to them behind the collective (round-3 advisor).
        Blocks whose exchange has completed meanwhile are un-dealt into the result here, so that their receive buffers
        are freed as the step proceeds instead of all being held until finish()."""
        still = []
        for entry in self._pending:
            if entry[0].is_completed():
                self._undeal(entry)
            else:
                still.append(entry)
        self._pending = still
        nb = q1 - q0
        have = max(0, min(q1, self.km) - q0)
        for p, loc in enumerate(locals_):
            dst = self.out[p][q0 * self.world:q1 * self.world]
            if self.world == 1:
                dst[:have] = loc[:have]
                continue
            if have == nb:
                send = loc[:nb]
                if copy:
                    send = send.clone(memory_format=torch.contiguous_format)
                elif not send.is_contiguous():
                    send = send.contiguous()
            else:                                   # the last block of a rank with a short share: pad with zeros
                send = torch.zeros(nb, self.parts[p], dtype=loc.dtype, device=self.out[p].device)
                send[:have] = loc[:have]
            # (world, nb, len) rank-major; un-dealt below
            tmp = torch.empty(self.world * nb, self.parts[p], dtype=loc.dtype, device=self.out[p].device)
            work = dist.all_gather_into_tensor(tmp, send, async_op=True)
            self._pending.append((work, tmp, send, p, q0, q1))

    def finish(self):
        """Waits for every posted block; returns the gathered (n_columns, parts[p]) tensors, row j = column j."""
        for entry in self._pending:
            entry[0].wait()
            self._undeal(entry)
        self._pending = []
        return [o[:self.n_columns] for o in self.out]
