"""world_size-2 `gloo` test of the N > 1 path: problems are dealt to ranks with no data-path
collective; only the timing (MAX) and the per-problem records are exchanged."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cuclarabel_amd.distributed import assign_problems, gather_records, reduce_max


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_problems, q, weights=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cuclarabel_amd import problems
        from tests.oracle_bindings import make_oracle
        mine = assign_problems(n_problems, world, rank, weights=weights)
        recs = []
        for j in mine:
            # each rank works on its own independent SOCP (cfg4's generator, tiny): the oracle stands
            # in for the GPU here -- the sharding and the record exchange are what is under test
            pb = problems.config4(j=j, n=200)
            o = make_oracle(pb)
            ok = o.update_scaling(pb.s0, pb.z0) and o.kktsolver_update()
            o.kktsolver_setrhs(-pb.q, pb.b)
            ok2, x, _ = o.kktsolver_solve()
            recs.append([j, float(ok and ok2), o.last_ir_iters, float(np.abs(x).max())])
        table = gather_records(recs, n_problems, 4)
        tmax = reduce_max(1.0 + rank)
        q.put((rank, mine, table.numpy(), tmax))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_record_gather():
    world, n_problems = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_problems, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort(key=lambda t: t[0])
    assert out[0][1] == [0, 2, 4] and out[1][1] == [1, 3]
    # every rank sees every problem's record, identical on both ranks, all solved
    np.testing.assert_array_equal(out[0][2], out[1][2])
    table = out[0][2]
    assert list(table[:, 0]) == [0, 1, 2, 3, 4]
    assert np.all(table[:, 1] == 1.0)
    assert np.all(np.isfinite(table[:, 3]))
    assert out[0][3] == out[1][3] == 2.0          # MAX over ranks


def test_skewed_weights_with_record_gather():
    """LPT with weights [10, 1, 1, 1] on 2 ranks gives rank 1 three problems, more than ceil(4 / 2): the record
    gather must size its buffer from the largest share, not from n_problems / world."""
    world, weights = 2, [10.0, 1.0, 1.0, 1.0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, len(weights), q, weights)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort(key=lambda t: t[0])
    assert out[0][1] == [0] and out[1][1] == [1, 2, 3]
    np.testing.assert_array_equal(out[0][2], out[1][2])
    assert list(out[0][2][:, 0]) == [0, 1, 2, 3]
    assert np.all(out[0][2][:, 1] == 1.0)


def test_weighted_assignment_balances_and_is_a_partition():
    w = [9, 1, 1, 1, 8, 2, 2, 7]
    parts = [assign_problems(len(w), 3, r, weights=w) for r in range(3)]
    assert sorted(sum(parts, [])) == list(range(len(w)))
    loads = [sum(w[j] for j in p) for p in parts]
    assert max(loads) - min(loads) <= 2


def _worker_columns(rank, world, port, n_columns, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cuclarabel_amd import problems
        from cuclarabel_amd.distributed import gather_columns, shard_columns
        from tests.oracle_bindings import make_oracle
        # ONE KKT system, replicated: every rank factorises it itself and solves its share of columns
        pb = problems.config2(n=300)
        o = make_oracle(pb)
        assert o.update_scaling(pb.s0, pb.z0) and o.kktsolver_update()
        rng = np.random.default_rng(77)
        RX, RZ = rng.standard_normal((n_columns, pb.n)), rng.standard_normal((n_columns, pb.m))
        mine = shard_columns(n_columns, world, rank)
        local = torch.zeros(max(len(mine), 1), pb.n, dtype=torch.float64)
        for qi, j in enumerate(mine):
            o.kktsolver_setrhs(RX[j], RZ[j])
            ok, x, _ = o.kktsolver_solve()
            assert ok
            local[qi] = torch.from_numpy(x)
        full = gather_columns(local, n_columns)
        q.put((rank, mine, full.numpy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_rhs_column_sharding_and_solution_gather():
    world, n_columns = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_columns, args=(r, world, port, n_columns, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort(key=lambda t: t[0])
    assert out[0][1] == [0, 2, 4] and out[1][1] == [1, 3]
    np.testing.assert_array_equal(out[0][2], out[1][2])        # every rank holds every column, in order
    # and they are the solutions a single process gets
    from cuclarabel_amd import problems
    from tests.oracle_bindings import make_oracle
    pb = problems.config2(n=300)
    o = make_oracle(pb)
    assert o.update_scaling(pb.s0, pb.z0) and o.kktsolver_update()
    rng = np.random.default_rng(77)
    RX, RZ = rng.standard_normal((n_columns, pb.n)), rng.standard_normal((n_columns, pb.m))
    for j in range(n_columns):
        o.kktsolver_setrhs(RX[j], RZ[j])
        _, x, _ = o.kktsolver_solve()
        np.testing.assert_array_equal(out[0][2][j], x)


def _worker_blocked(rank, world, port, n_columns, block, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cuclarabel_amd.distributed import BlockedColumnGather, gather_columns, shard_columns
        n, m = 7, 11
        mine = shard_columns(n_columns, world, rank)
        # column j's "solution": x part = j + 0.01 * index, z part = -j - 0.01 * index
        LX = torch.zeros(max(len(mine), 1), n, dtype=torch.float64)
        LZ = torch.zeros(max(len(mine), 1), m, dtype=torch.float64)
        for qi, j in enumerate(mine):
            LX[qi] = j + 0.01 * torch.arange(n, dtype=torch.float64)
            LZ[qi] = -j - 0.01 * torch.arange(m, dtype=torch.float64)
        g = BlockedColumnGather(n_columns, (n, m), block)
        # the natural caller pattern: ONE block buffer, filled, posted, and overwritten for the next block at once -- the
        # object must have taken its own copy (post's default; round-3 advisor: the rows used to be sent in place)
        bx = torch.zeros(block, n, dtype=torch.float64)
        bz = torch.zeros(block, m, dtype=torch.float64)
        for q0, q1 in g.blocks():                  # solve block, post its exchange, go on with the next block
            k = max(0, min(q1, len(mine)) - q0)
            bx[:k] = LX[q0:q0 + k]
            bz[:k] = LZ[q0:q0 + k]
            g.post(q0, q1, [bx, bz])
            bx.fill_(float("nan"))
            bz.fill_(float("nan"))
        X, Z = g.finish()
        ref = gather_columns(LX, n_columns)        # the one-shot exchange of the same data
        q.put((rank, X.numpy(), Z.numpy(), ref.numpy(), len(g.blocks())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_columns,block", [(9, 2), (8, 4), (5, 16)])
def test_two_rank_blocked_asynchronous_column_gather(n_columns, block):
    """bench.py --mode rhs exchanges the (x, z) solutions in blocks of columns, block i on the links while block i + 1
    is solved (BlockedColumnGather): every rank must end with every column, x and z parts, in order -- also with a
    short last block and with ranks holding different numbers of columns -- and agree with the one-shot gather."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_blocked, args=(r, world, port, n_columns, block, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, X, Z, ref, nblocks in out:
        assert X.shape == (n_columns, 7) and Z.shape == (n_columns, 11)
        for j in range(n_columns):
            np.testing.assert_array_equal(X[j], j + 0.01 * np.arange(7))
            np.testing.assert_array_equal(Z[j], -j - 0.01 * np.arange(11))
        np.testing.assert_array_equal(X, ref)
        per = (n_columns + world - 1) // world
        assert nblocks == (per + block - 1) // block
