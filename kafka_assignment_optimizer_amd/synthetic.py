"""Synthetic clusters for the BASELINE.json configs, product side.

Old cluster: brokers 0..B0-1 with rack(b) = b mod R (generalises README.md:28-29); every topic gets
a deterministic rack-aware balanced assignment; then a seeded change set (splitmix64, seed
0x4B414F00 + config number) removes and/or adds brokers.  tests/test_synthetic.py checks these
arrays against the oracle's independent generator.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from .model import DEFAULT_WEIGHTS, NONE, Topic

CONFIG_SEED = 0x4B414F00
_M64 = 0xFFFFFFFFFFFFFFFF

WORKLOADS = {
    1: "cfg1: README example, 20 brokers / 2 AZ, 1 topic x 10 partitions RF 2, remove broker 19",
    2: "cfg2: 100 brokers / 4 racks, 1 topic x 256 partitions RF 3, remove 1 broker",
    3: "cfg3: 200 brokers / 6 racks, 50 topics x 64 partitions RF 3, add 20 brokers",
    4: "cfg4: 500 brokers / 10 racks, 200 topics x 50 partitions (10k partitions) RF 3, rolling replace 50 brokers",
    5: "cfg5: 1000 brokers / 20 racks, 1000 topics x 100 partitions (100k partitions) RF 3, remove 50 + add 50, per-broker cap ceil(avg)+1",
}


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & _M64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def below(self, n: int) -> int:
        return self.next() % n

    def sample(self, items: Sequence[int], k: int) -> List[int]:
        pool = list(items)
        return [pool.pop(self.below(len(pool))) for _ in range(k)]


def balanced_fill(n_brokers: int, n_racks: int, n_partitions: int, rf: int, t: int, rack_of: np.ndarray) -> np.ndarray:
    """Slot k of partition p -> rack (t+p+k) mod R, least-loaded broker of that rack not already in
    p; leaders ordered by (leaders, replicas, id), followers by (replicas, id)."""
    members = [np.nonzero(rack_of == r)[0] for r in range(n_racks)]
    cnt_r = np.zeros(n_brokers, dtype=np.int64)
    cnt_l = np.zeros(n_brokers, dtype=np.int64)
    out = np.full((n_partitions, rf), NONE, dtype=np.uint16)
    big = 1 << 60
    for p in range(n_partitions):
        used: List[int] = []
        for k in range(rf):
            mem = members[(t + p + k) % n_racks]
            key = (cnt_l[mem] << 40) + (cnt_r[mem] << 20) + mem if k == 0 else (cnt_r[mem] << 20) + mem
            if used:
                key = np.where(np.isin(mem, used), big, key)
            b = int(mem[int(np.argmin(key))])
            if k == 0:
                cnt_l[b] += 1
            cnt_r[b] += 1
            used.append(b)
            out[p, k] = b
    return out


def make_cluster(n_brokers0: int, n_racks: int, n_topics: int, n_partitions: int, rf: int, removed: Sequence[int],
                 added: Sequence[Tuple[int, int]], weights=DEFAULT_WEIGHTS, bounds_override=None,
                 new_rf: Optional[int] = None) -> List[Topic]:
    rack0 = np.arange(n_brokers0) % n_racks
    gone = set(int(b) for b in removed)
    target = [b for b in range(n_brokers0) if b not in gone] + [int(b) for b, _ in added]
    rack_t = [int(rack0[b]) for b in range(n_brokers0) if b not in gone] + [int(r) for _, r in added]
    lut = np.full(n_brokers0, NONE, dtype=np.uint16)
    for i, b in enumerate(target):
        if b < n_brokers0:
            lut[b] = i
    topics = []
    for t in range(n_topics):
        cur_old = balanced_fill(n_brokers0, n_racks, n_partitions, rf, t, rack0)
        topics.append(Topic(name=f"topic-{t:04d}", broker_ids=np.array(target, dtype=np.int32),
                            rack_of=np.array(rack_t, dtype=np.uint8), n_racks=n_racks, n_partitions=n_partitions,
                            rf=new_rf or rf, current=lut[cur_old], weights=weights,
                            bounds_override=dict(bounds_override or {})))
    return topics


def drift(topics: Sequence[Topic], frac: float = 0.2, seed: int = 1) -> List[Topic]:
    """The same topics after the cluster has drifted: `frac` of every topic's replica slots moved to a random target
    broker (never one already holding the partition).  The bands do not depend on the current assignment, so the
    instances stay feasible; what changes is that the optimum is no longer "keep everything that survived", i.e. the
    closed-form bound has a gap and the certificate has to come from K-bound."""
    out = []
    for ti, t in enumerate(topics):
        rng = SplitMix64(0xD21F7000 + seed * 1000003 + ti)
        cur = np.array(t.current).copy()
        P, rfc, B = cur.shape[0], cur.shape[1], len(t.broker_ids)
        for _ in range(int(P * rfc * frac)):
            p, k, nb = rng.below(P), rng.below(rfc), rng.below(B)
            if nb not in cur[p]:
                cur[p, k] = nb
        out.append(Topic(name=t.name + "-drift", broker_ids=t.broker_ids, rack_of=t.rack_of, n_racks=t.n_racks,
                         n_partitions=t.n_partitions, rf=t.rf, current=cur, weights=t.weights,
                         partition_ids=t.partition_ids, bounds_override=dict(t.bounds_override)))
    return out


def make_config(n: int, n_topics: Optional[int] = None) -> List[Topic]:
    """Topics of BASELINE.json config `n` (2..5); `n_topics` truncates."""
    rng = SplitMix64(CONFIG_SEED + n)
    if n == 2:
        return make_cluster(100, 4, n_topics or 1, 256, 3, [rng.below(100)], [])
    if n == 3:
        return make_cluster(200, 6, n_topics or 50, 64, 3, [], [(b, b % 6) for b in range(200, 220)])
    if n == 4:
        rm, add, nid = [], [], 500
        for r in range(10):
            for b in rng.sample([b for b in range(500) if b % 10 == r], 5):
                rm.append(b)
                add.append((nid, r))
                nid += 1
        return make_cluster(500, 10, n_topics or 200, 50, 3, rm, add)
    if n == 5:
        rm = rng.sample(list(range(1000)), 50)
        add = [(1000 + i, rng.below(20)) for i in range(50)]
        cap = -((-100 * 3) // 1000) + 1
        return make_cluster(1000, 20, n_topics or 1000, 100, 3, rm, add, bounds_override={"rep_hi": cap})
    raise ValueError("config must be 2..5 (config 1 is the README example: use topics_from_json)")


def north_star_topic(which: str) -> Topic:
    """The single large topics of the north-star regime (BASELINE config 5 and its drifted relatives), shared by bench.py's
    `roofline_big_topic` leg, tools/big_topic.py and the GPU tests:
      cfg5one  = config 5 taken literally as ONE topic: 1000 brokers / 20 racks, 100,000 partitions RF 3, 50 brokers replaced
                 (each new broker joins the rack of a removed one: with uneven racks the single-topic rack band would be
                 infeasible), per-broker cap ceil(avg)+1;
      drift30k = 1000 brokers / 20 racks x 30,000 partitions after a 20 % drift;   drift5k = 500 / 10 x 5,000 and
      drift100k = 1000 / 20 x 100,000 likewise."""
    if which == "drift30k":
        return drift(make_cluster(1000, 20, 1, 30000, 3, [], []), 0.2, 1)[0]
    if which == "drift5k":
        return drift(make_cluster(500, 10, 1, 5000, 3, [], []), 0.2, 1)[0]
    if which == "drift100k":      # the north-star size after a 20 % drift: no balanced start the init could simply keep
        return drift(make_cluster(1000, 20, 1, 100_000, 3, [], []), 0.2, 1)[0]
    if which == "cfg5one":
        rng = SplitMix64(CONFIG_SEED + 5)
        rm = rng.sample(list(range(1000)), 50)
        add = [(1000 + i, b % 20) for i, b in enumerate(rm)]
        return make_cluster(1000, 20, 1, 100_000, 3, rm, add, bounds_override={"rep_hi": 301})[0]
    raise ValueError("north_star_topic: drift30k | drift5k | drift100k | cfg5one")


def north_star_steps(kao, which: str, launches: int = 6, iters: int = 512, restarts: int = 0) -> dict:
    """`launches` K-search + K-eval steps of a session on north_star_topic(which) after the init launch: per-launch HIP-event
    times and SURVEY.md 8(d)'s algorithmic bytes (`kao` = the package, initialised)."""
    import time
    t = north_star_topic(which)
    opts = dict(seed=3, iters_per_launch=iters, profile=1)
    if restarts:
        opts["restarts"] = restarts
    with kao.Session([t], **opts) as s:
        s.step(1)                      # launch 0: best-insertion init + the first iterations (not timed)
        s.sync()
        a = s.stats()
        t0 = time.perf_counter()
        s.step(launches)
        s.sync()
        wall = time.perf_counter() - t0
        b = s.stats()
        best = s.best()[0]
    n = launches
    rf, P, B = t.rf, t.n_partitions, t.n_brokers
    nb = (b["delta_candidates"] - a["delta_candidates"]) / n
    return {"workload": which, "brokers": B, "partitions": P, "rf": rf, "restarts": b["n_restarts_total"], "iters_per_launch": iters,
            "launches_timed": n, "k_search_ms_per_launch": (b["ms_search"] - a["ms_search"]) / n, "k_eval_ms_per_launch": (b["ms_eval"] - a["ms_eval"]) / n,
            "wall_ms_per_launch": 1e3 * wall / n, "neighbours_per_launch": nb, "k_search_algorithmic_bytes_per_launch": nb * (8 * rf + 10),
            "k_eval_algorithmic_bytes_per_launch": b["n_restarts_total"] * (4 * rf * P + B), "k_search_lds_bytes": b["lds_bytes_search"],
            "k_search_workgroups": b["blocks_search"], "objective_after": int(best.objective), "violation_after": int(best.violations[0]), "drift": b["drift"]}
