"""Topic sharding across the GPUs of one node (one process per GPU, torch.distributed).

Topics are independent sub-problems (every variable and row of the README model carries the topic
prefix, README.md:146-184), so the data path needs no collective: each rank searches its own
topics.  The one exchange step is the result merge -- a min-allreduce of the packed per-topic best
keys (RCCL `ncclMin` on int64 when the backend is "nccl"; gloo on CPU in the tests) followed by a
broadcast-free gather of the winners' assignments (each topic has exactly one owner unless topics
are replicated, in which case the allreduce picks the global best and the winning rank supplies it).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

KEY_NONE = (1 << 63) - 1  # int64 max: "this rank has no result for the topic"
RANK_BITS = 4             # up to 16 ranks per node folded into the key's low bits


def shard_topics(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time deal: topics sorted by size (B*P) descending, each to the least
    loaded rank (ties -> lowest rank).  Returns the topic indices of every rank."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(sizes[i])
    for r in range(world):
        out[r].sort()
    return out


def pack_for_allreduce(keys_u64: np.ndarray, rank: int) -> np.ndarray:
    """Device key (viol 20b | cost 24b | restart 20b) -> int64 whose minimum also identifies the
    rank: the restart id keeps its 20 bits, the rank takes the low RANK_BITS of a left-shifted key.
    The top bits stay clear (20+24+20+4 = 68 > 63), so the violation field is saturated to 15 bits."""
    k = np.asarray(keys_u64, dtype=np.uint64)
    viol = np.minimum(k >> np.uint64(44), np.uint64(0x7FFF))
    rest = k & np.uint64((1 << 44) - 1)
    packed = (viol << np.uint64(44 + RANK_BITS)) | (rest << np.uint64(RANK_BITS)) | np.uint64(rank)
    return packed.astype(np.int64)


def unpack_allreduced(v: int):
    """-> (violation, objective, restart, rank)"""
    v = int(v)
    rank = v & ((1 << RANK_BITS) - 1)
    v >>= RANK_BITS
    return v >> 44, 0xFFFFFF - ((v >> 20) & 0xFFFFFF), v & 0xFFFFF, rank


def allreduce_best(local_keys_u64: np.ndarray, owned: Sequence[int], n_topics: int, rank: int, device=None):
    """Min-allreduce of the packed best keys over all ranks.

    local_keys_u64[i] is this rank's key for topic owned[i].  Returns an int64 numpy array
    [n_topics] holding, for every topic, the globally best (violation, cost, restart, rank)."""
    import torch
    import torch.distributed as dist

    full = np.full(n_topics, KEY_NONE, dtype=np.int64)
    if len(owned):
        full[np.asarray(owned, dtype=np.int64)] = pack_for_allreduce(local_keys_u64, rank)
    t = torch.from_numpy(full)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return t.cpu().numpy()


class _DevKeys:
    """Exposes a session's resident key buffer to torch (no copy) through __cuda_array_interface__."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 3}


def allreduce_best_resident(sess, owned: Sequence[int], n_topics: int, rank: int):
    """The same min-allreduce without leaving the GPU: the session's packed keys are read where K-eval wrote them (zero-copy
    view), re-packed with the rank on the device, scattered into a [n_topics] int64 tensor and all-reduced (RCCL ncclMin over
    xGMI when the backend is "nccl").  Returns the DEVICE tensor (same encoding as allreduce_best)."""
    import torch
    import torch.distributed as dist

    sess.sync()                                   # K-eval wrote the keys on the session's own stream
    dev = torch.device("cuda", torch.cuda.current_device())
    full = torch.full((n_topics,), KEY_NONE, dtype=torch.int64, device=dev)
    if len(owned):
        k = torch.as_tensor(_DevKeys(sess.device_keys_ptr(), len(owned)), device=dev)
        viol = torch.clamp((k >> 44) & 0xFFFFF, max=0x7FFF)
        rest = k & ((1 << 44) - 1)
        full[torch.as_tensor(list(owned), dtype=torch.int64, device=dev)] = (viol << (44 + RANK_BITS)) | (rest << RANK_BITS) | rank
    dist.all_reduce(full, op=dist.ReduceOp.MIN)
    return full


def allreduce_bounds(local_bounds: Sequence[int], owned: Sequence[int], n_topics: int, device=None) -> np.ndarray:
    """Min-allreduce of the per-topic certificates (kao_result.upper_bound).  Every rank's value is a valid upper bound
    on the topic's optimum (closed-form bound or a Lagrangian dual value), so the smallest one is the best certificate --
    this matters when a topic is replicated over several GPUs, whose K-bound runs aim at different incumbents.
    Returns int64 [n_topics] (KEY_NONE where no rank owns the topic)."""
    import torch
    import torch.distributed as dist

    full = np.full(n_topics, KEY_NONE, dtype=np.int64)
    if len(owned):
        full[np.asarray(owned, dtype=np.int64)] = np.asarray(local_bounds, dtype=np.int64)
    t = torch.from_numpy(full)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return t.cpu().numpy()


def allreduce_loads(local_loads, device=None) -> np.ndarray:
    """Cluster-wide per-broker caps couple the topics: when topics are sharded over processes, one price
    round of kao_solve_capped's scheme needs the broker loads summed over ALL ranks -- allreduce(SUM) of int64[n_brokers]
    (4-8 KB at 1000 brokers; RCCL ncclSum when the backend is "nccl").  Returns the global loads on every rank."""
    import torch
    import torch.distributed as dist

    t = torch.as_tensor(np.asarray(local_loads, dtype=np.int64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def gather_assignments(best: np.ndarray, owned: Sequence[int], local_assignments: Sequence[np.ndarray],
                       rank: int, world: int) -> List[np.ndarray]:
    """Every rank contributes the assignments of the topics it won; rank 0 receives all of them."""
    import torch.distributed as dist

    mine = {}
    for i, t in enumerate(owned):
        if unpack_allreduced(best[t])[3] == rank and best[t] != KEY_NONE:
            mine[int(t)] = np.asarray(local_assignments[i])
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    merged = {}
    for d in gathered:
        merged.update(d)
    return [merged.get(t) for t in range(len(best))]


# ---- KAO-LP sharded by partition range (round 6): what the shards exchange per factorisation -------------------------------------
def tri_pack(S: np.ndarray, mc: int) -> np.ndarray:
    """A shard's part of the Schur complement for the all-reduce: rows 0 .. mc-1 of the lower triangle of S [mcp, mcp], row i at offset
    i (i + 1) / 2 -- the index map of k_lp_tri_pack (kao_lp.hip): half the bytes of the square."""
    i, j = np.tril_indices(mc)
    return np.ascontiguousarray(S[i, j], dtype=np.float64)


def tri_unpack(tri: np.ndarray, mc: int, mcp: int, out: np.ndarray = None) -> np.ndarray:
    """Inverse of tri_pack into the lower triangle of an [mcp, mcp] matrix (everything else untouched / zero)."""
    S = np.zeros((mcp, mcp)) if out is None else out
    i, j = np.tril_indices(mc)
    S[i, j] = tri
    return S


def allreduce_schur(S_local: np.ndarray, mc: int, device=None) -> np.ndarray:
    """One process per GPU: the shards' lower triangles summed (f64) over the process group, packed; every rank gets the same matrix.
    (Inside one process the library does this itself: kao_lp_sharded_test, LpGroup in kao_solve.cpp.)"""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(tri_pack(S_local, mc))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tri_unpack(t.cpu().numpy(), mc, S_local.shape[0])
