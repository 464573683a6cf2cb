"""world_size-2 gloo tests (CPU) of the multi-GPU merge: topic sharding + min-allreduce of the
packed best keys + gather of the winners' assignments."""
import os
import socket

import numpy as np

from kafka_assignment_optimizer_amd import multigpu as mg


def test_shard_topics_lpt():
    sizes = [10, 50, 20, 50, 5, 30, 1]
    sh = mg.shard_topics(sizes, 3)
    assert sorted(i for s in sh for i in s) == list(range(7))
    loads = [sum(sizes[i] for i in s) for s in sh]
    assert max(loads) - min(loads) <= max(sizes)
    assert mg.shard_topics(sizes, 1) == [list(range(7))]
    assert mg.shard_topics([7, 7], 4)[2:] == [[], []]


def test_key_packing_orders_like_device_key():
    def dev(v, obj, rho):
        return (v << 44) | ((0xFFFFFF - obj) << 20) | rho
    ks = np.array([dev(0, 366, 5), dev(0, 360, 1), dev(2, 400, 0), dev(0, 366, 4)], dtype=np.uint64)
    p = mg.pack_for_allreduce(ks, rank=3)
    assert (p > 0).all()
    assert np.argsort(p).tolist() == np.argsort(ks).tolist()
    assert mg.unpack_allreduced(p[0]) == (0, 366, 5, 3)
    assert mg.pack_for_allreduce(np.array([dev(0xFFFFF, 0, 0)], dtype=np.uint64), 15)[0] < mg.KEY_NONE


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_topics = 5
        shards = mg.shard_topics([3, 9, 4, 9, 1], world)
        owned = shards[rank]

        def dev(v, obj, rho):
            return (v << 44) | ((0xFFFFFF - obj) << 20) | rho
        # topic 4 is additionally searched by BOTH ranks (replicated): rank 1 finds the better one
        keys = [dev(0, 100 + t, t) for t in owned]
        assigns = [np.full((2, 2), 10 * rank + t) for t in owned]
        if 4 not in owned:
            owned = owned + [4]
            keys.append(dev(0, 500 + rank, 7))
            assigns.append(np.full((2, 2), 10 * rank + 4))
        else:
            i = owned.index(4)
            keys[i] = dev(0, 500 + rank, 7)
        best = mg.allreduce_best(np.array(keys, dtype=np.uint64), owned, n_topics, rank)
        got = mg.gather_assignments(best, owned, assigns, rank, world)
        # certificates: every rank's upper bound is valid, the smallest wins (rank 0 holds the tighter one for topic 4)
        bounds = mg.allreduce_bounds([120 + t if t != 4 else 510 - 5 * (1 - rank) for t in owned], owned, n_topics)
        loads = mg.allreduce_loads(np.array([rank + 1, 10 * rank, 5]))   # broker loads of this rank's topics -> cluster-wide loads
        q.put((rank, best.tolist(), [None if g is None else g.tolist() for g in got], bounds.tolist(), loads.tolist()))
    finally:
        dist.destroy_process_group()


def test_allreduce_best_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, best0, got0, bounds0, loads0), (_, best1, got1, bounds1, loads1) = res
    assert bounds0 == bounds1 == [120, 121, 122, 123, 505]
    assert loads0 == loads1 == [3, 10, 10]  # allreduce(SUM) of the per-rank broker loads (cluster-wide caps)
    assert best0 == best1  # every rank holds the same global result
    dec = [mg.unpack_allreduced(v) for v in best0]
    shards = mg.shard_topics([3, 9, 4, 9, 1], 2)
    for t in range(4):
        owner = 0 if t in shards[0] else 1
        assert dec[t] == (0, 100 + t, t, owner)
        assert got0[t] == [[10 * owner + t] * 2] * 2
    assert dec[4] == (0, 501, 7, 1)  # replicated topic: the better objective (rank 1) wins
    assert got0[4] == [[14, 14], [14, 14]] and got0 == got1


def _schur_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mc, mcp = 70, 128
        rng = np.random.default_rng(100 + rank)
        S = np.zeros((mcp, mcp))
        S[:mc, :mc] = np.tril(rng.standard_normal((mc, mc)))       # this shard's partitions' part of the lower triangle
        S[mc:, :] = 7.0 + rank                                     # padding rows: not part of the exchange
        out = mg.allreduce_schur(S, mc)
        q.put((rank, S[:mc, :mc].tolist(), out.tolist()))
    finally:
        dist.destroy_process_group()


def test_tri_packing_matches_the_kernel_index_map():
    """Row i of the lower triangle at offset i (i + 1) / 2 (k_lp_tri_pack), round trip, nothing outside the triangle travels."""
    mc, mcp = 9, 64
    S = np.arange(mcp * mcp, dtype=np.float64).reshape(mcp, mcp)
    t = mg.tri_pack(S, mc)
    assert t.shape == (mc * (mc + 1) // 2,)
    for i in range(mc):
        for j in range(i + 1):
            assert t[i * (i + 1) // 2 + j] == S[i, j]
    back = mg.tri_unpack(t, mc, mcp)
    assert np.array_equal(np.tril(back[:mc, :mc]), np.tril(S[:mc, :mc])) and back[mc:].sum() == 0 and np.triu(back, 1).sum() == 0


def test_schur_allreduce_gloo_world2():
    """KAO-LP sharded by partition range: the shards' parts of the Schur complement summed over a world of two (gloo), packed -- every
    rank ends with the dense sum of the lower triangles, bit for bit the same on both."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_schur_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, part0, out0), (_, part1, out1) = res
    want = np.array(part0) + np.array(part1)
    assert np.array_equal(np.array(out0), np.array(out1))
    assert np.array_equal(np.array(out0)[:70, :70], want) and np.array(out0)[70:].sum() == 0
