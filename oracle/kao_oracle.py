"""CPU oracle for the Kafka partition-assignment 0-1 model -- TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU and in exact integer arithmetic, the model that
`/root/reference/README.md` (the whole reference snapshot) specifies, so that the HIP
path can be checked against it.  Nothing in the product (`kafka_assignment_optimizer_amd/`)
may import it; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg do.

PARITY STATUS: **parity unpinned beyond KAT-1**.  The reference's solver is lp_solve 5.5
(README.md:135-136, README.md:200), a third-party dependency that is neither vendored in
the snapshot, nor version-pinned by it (there is no build file), nor installed here.  The
exact solver used below is HiGHS through `scipy.optimize.milp` on the identical README
model; the only result the reference itself pins is the worked example (README.md:52-63 ->
README.md:85-91), which `readme_example()` / tests/golden/kat1.json reproduce.

Reference citations (README.md:N = /root/reference/README.md line N):
  variables, order ............ README.md:122-124, README.md:182-184
  objective ................... README.md:116-120, README.md:145-146
  C1 replication factor ....... README.md:148-151
  C2 one leader ............... README.md:153-156
  C3 replicas per broker ...... README.md:158-161
  C4 leaders per broker ....... README.md:163-166
  C5 leader+follower <= 1 ..... README.md:168-171 (README.md:111, README.md:126-129)
  C6 replicas per rack ........ README.md:173-176
  C7 replicas/partition/rack .. README.md:178-180
  C8 binaries ................. README.md:182-184
  JSON in/out ................. README.md:52-63, README.md:67-78, README.md:88
"""
from __future__ import annotations

import itertools
import json
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

NONE = 0xFFFF  # "no broker" marker in compact uint16 arrays (removed broker / empty slot)

# w[cur_role][new_role], role 0 = leader, 1 = follower.  README.md:146 shows the weight
# multiset {1, 2, 2, 4}; SURVEY.md H1 scheme (A): LL=4, LF=1, FL=2, FF=2.
DEFAULT_WEIGHTS = ((4, 1), (2, 2))


def _floor_ceil(num: int, den: int) -> Tuple[int, int]:
    lo = num // den
    hi = -((-num) // den)
    return lo, hi


@dataclass
class Topic:
    """One topic's sub-problem in compact form (dense broker / rack indices)."""

    name: str
    broker_ids: np.ndarray  # [B] external Kafka broker ids of the TARGET broker set
    rack_of: np.ndarray  # [B] dense rack index of each target broker
    n_racks: int
    n_partitions: int
    rf: int
    current: np.ndarray  # [P, rf_cur] uint16 dense broker index, NONE = not in target set
    weights: Tuple[Tuple[int, int], Tuple[int, int]] = DEFAULT_WEIGHTS
    partition_ids: Optional[np.ndarray] = None  # external partition numbers (default 0..P-1)
    # band overrides; -1 = derive floor/ceil of the average (README.md:159-160 etc.)
    bounds_override: Dict[str, int] = field(default_factory=dict)
    # optional broker weights: extra objective coefficients on every variable of a broker (kao_topic.broker_w / broker_wl):
    # broker_w[b] on t?b<b>p? and t?b<b>p?_l, broker_wl[b] on t?b<b>p?_l -- plain coefficients of the `max:` row (README.md:145-146)
    broker_w: Optional[np.ndarray] = None
    broker_wl: Optional[np.ndarray] = None

    @property
    def n_brokers(self) -> int:
        return int(len(self.broker_ids))

    @property
    def rf_cur(self) -> int:
        return int(self.current.shape[1])

    def bounds(self) -> Dict[str, int]:
        """Band right-hand sides.  The README shows only example values; floor/ceil of the
        average is the unique simple rule consistent with README.md:159-160 (20 replicas on
        19 brokers -> 1..2), README.md:164-165 (10 leaders -> 0..1), README.md:174-175
        (20 replicas / 2 racks -> exactly 10) and README.md:179 (RF 2 / 2 racks -> <= 1)."""
        B, R, P, RF = self.n_brokers, self.n_racks, self.n_partitions, self.rf
        d = {}
        d["rep_lo"], d["rep_hi"] = _floor_ceil(P * RF, B)
        d["lead_lo"], d["lead_hi"] = _floor_ceil(P, B)
        d["rack_lo"], d["rack_hi"] = _floor_ceil(P * RF, R)
        d["prack_lo"], d["prack_hi"] = _floor_ceil(RF, R)
        for k, v in self.bounds_override.items():
            if v is not None and v >= 0:
                d[k] = int(v)
        return d

    def weight_matrix(self) -> np.ndarray:
        """W[p, b, new_role] objective coefficient of putting broker b on partition p."""
        P, B = self.n_partitions, self.n_brokers
        W = np.zeros((P, B, 2), dtype=np.int64)
        w = self.weights
        for p in range(P):
            for k in range(self.rf_cur):
                b = int(self.current[p, k])
                if b == NONE or b >= B:
                    continue
                cr = 0 if k == 0 else 1
                W[p, b, 0] = w[cr][0]
                W[p, b, 1] = w[cr][1]
        if self.broker_w is not None:
            W += np.asarray(self.broker_w, dtype=np.int64)[None, :, None]
        if self.broker_wl is not None:
            W[:, :, 0] += np.asarray(self.broker_wl, dtype=np.int64)[None, :]
        return W


# --------------------------------------------------------------------------------------
# Verifier: full evaluation of one compact candidate (the K-eval kernel's oracle)
# --------------------------------------------------------------------------------------
def _band_excess(c: np.ndarray, lo: int, hi: int) -> int:
    c = np.asarray(c, dtype=np.int64)
    return int(np.maximum(c - hi, 0).sum() + np.maximum(lo - c, 0).sum())


def verify(topic: Topic, assign: np.ndarray) -> Tuple[int, np.ndarray]:
    """Objective and per-family violation magnitudes of a compact candidate.

    assign: [P, RF] dense broker index, slot 0 = leader, NONE/out-of-range = empty slot.
    Returns (objective, viol[8]) with viol[f] = sum over the rows of family Cf of the amount
    by which the row misses its right-hand side, viol[0] = total.  Feasible <=> viol[0]==0.
    """
    B, R, P, RF = topic.n_brokers, topic.n_racks, topic.n_partitions, topic.rf
    a = np.asarray(assign).reshape(P, RF).astype(np.int64)
    bd = topic.bounds()
    valid = (a >= 0) & (a < B)
    viol = np.zeros(8, dtype=np.int64)
    viol[1] = int((~valid).sum())  # C1: sum_b (f+l) = RF   (README.md:148-151)
    viol[2] = int((~valid[:, 0]).sum())  # C2: sum_b l = 1   (README.md:153-156)
    cnt_r = np.zeros(B, dtype=np.int64)
    cnt_l = np.zeros(B, dtype=np.int64)
    np.add.at(cnt_r, a[valid], 1)
    np.add.at(cnt_l, a[:, 0][valid[:, 0]], 1)
    viol[3] = _band_excess(cnt_r, bd["rep_lo"], bd["rep_hi"])  # README.md:158-161
    viol[4] = _band_excess(cnt_l, bd["lead_lo"], bd["lead_hi"])  # README.md:163-166
    rack = np.asarray(topic.rack_of, dtype=np.int64)
    rack_cnt = np.zeros(R, dtype=np.int64)
    np.add.at(rack_cnt, rack, cnt_r)
    viol[6] = _band_excess(rack_cnt, bd["rack_lo"], bd["rack_hi"])  # README.md:173-176
    W = topic.weight_matrix()
    obj = 0
    c5 = 0
    c7 = 0
    for p in range(P):
        occ: Dict[int, int] = {}
        pr = np.zeros(R, dtype=np.int64)
        for k in range(RF):
            if not valid[p, k]:
                continue
            b = int(a[p, k])
            occ[b] = occ.get(b, 0) + 1
            pr[rack[b]] += 1
            obj += int(W[p, b, 0 if k == 0 else 1])
        c5 += sum(v - 1 for v in occ.values())  # README.md:168-171
        c7 += _band_excess(pr, bd["prack_lo"], bd["prack_hi"])  # README.md:178-180
    viol[5] = c5
    viol[7] = c7
    viol[0] = int(viol[1:].sum())
    return int(obj), viol


def count_moves(topic: Topic, assign: np.ndarray) -> Tuple[int, int]:
    """(replica moves, leader changes) of `assign` relative to topic.current."""
    P, RF = topic.n_partitions, topic.rf
    a = np.asarray(assign).reshape(P, RF)
    moves = 0
    leader_changes = 0
    for p in range(P):
        cur = {int(x) for x in topic.current[p] if int(x) != NONE}
        moves += sum(1 for x in a[p] if int(x) not in cur)
        if int(a[p, 0]) != int(topic.current[p, 0]):
            leader_changes += 1
    return moves, leader_changes


# --------------------------------------------------------------------------------------
# The README's 0-1 model, materialised (variable order README.md:182-184)
# --------------------------------------------------------------------------------------
def var_index(topic: Topic, b: int, p: int, leader: bool) -> int:
    """Broker-major, partition-minor, follower-then-leader (README.md:184)."""
    return 2 * (b * topic.n_partitions + p) + (1 if leader else 0)


def build_rows(topic: Topic):
    """All rows of the README model as (family, {var: coef}, lo, hi) with lo/hi = None for
    a one-sided row.  Ranged rows (a band) carry both."""
    B, R, P, RF = topic.n_brokers, topic.n_racks, topic.n_partitions, topic.rf
    bd = topic.bounds()
    rows = []
    vi = lambda b, p, l: var_index(topic, b, p, l)
    for p in range(P):  # C1
        rows.append(("C1", {vi(b, p, l): 1 for b in range(B) for l in (False, True)}, RF, RF))
    for p in range(P):  # C2
        rows.append(("C2", {vi(b, p, True): 1 for b in range(B)}, 1, 1))
    for b in range(B):  # C3
        rows.append(("C3", {vi(b, p, l): 1 for p in range(P) for l in (False, True)},
                     bd["rep_lo"], bd["rep_hi"]))
    for b in range(B):  # C4
        rows.append(("C4", {vi(b, p, True): 1 for p in range(P)}, bd["lead_lo"], bd["lead_hi"]))
    for b in range(B):  # C5
        for p in range(P):
            rows.append(("C5", {vi(b, p, False): 1, vi(b, p, True): 1}, None, 1))
    for r in range(R):  # C6
        members = [b for b in range(B) if int(topic.rack_of[b]) == r]
        rows.append(("C6", {vi(b, p, l): 1 for b in members for p in range(P)
                            for l in (False, True)}, bd["rack_lo"], bd["rack_hi"]))
    for p in range(P):  # C7
        for r in range(R):
            members = [b for b in range(B) if int(topic.rack_of[b]) == r]
            if not members:
                continue
            rows.append(("C7", {vi(b, p, l): 1 for b in members for l in (False, True)},
                         bd["prack_lo"], bd["prack_hi"]))
    return rows


def objective_vector(topic: Topic) -> np.ndarray:
    B, P = topic.n_brokers, topic.n_partitions
    W = topic.weight_matrix()
    c = np.zeros(2 * B * P, dtype=np.int64)
    for b in range(B):
        for p in range(P):
            c[var_index(topic, b, p, False)] = W[p, b, 1]
            c[var_index(topic, b, p, True)] = W[p, b, 0]
    return c


def _sparse_model(topic: Topic):
    from scipy import sparse

    B, R, P, RF = topic.n_brokers, topic.n_racks, topic.n_partitions, topic.rf
    bd = topic.bounds()
    n = 2 * B * P
    bb, pp = np.meshgrid(np.arange(B), np.arange(P), indexing="ij")
    f_idx = (2 * (bb * P + pp)).ravel()
    l_idx = f_idx + 1
    b_of = bb.ravel()
    p_of = pp.ravel()
    rack = np.asarray(topic.rack_of, dtype=np.int64)
    r_of = rack[b_of]
    ri, ci = [], []
    lo, hi = [], []
    base = 0
    # C1 rows p
    ri += [base + p_of, base + p_of]; ci += [f_idx, l_idx]; lo += [RF] * P; hi += [RF] * P; base += P
    # C2
    ri += [base + p_of]; ci += [l_idx]; lo += [1] * P; hi += [1] * P; base += P
    # C3
    ri += [base + b_of, base + b_of]; ci += [f_idx, l_idx]
    lo += [bd["rep_lo"]] * B; hi += [bd["rep_hi"]] * B; base += B
    # C4
    ri += [base + b_of]; ci += [l_idx]; lo += [bd["lead_lo"]] * B; hi += [bd["lead_hi"]] * B; base += B
    # C5
    k = np.arange(B * P)
    ri += [base + k, base + k]; ci += [f_idx, l_idx]; lo += [-np.inf] * (B * P); hi += [1] * (B * P); base += B * P
    # C6
    ri += [base + r_of, base + r_of]; ci += [f_idx, l_idx]
    lo += [bd["rack_lo"]] * R; hi += [bd["rack_hi"]] * R; base += R
    # C7
    ri += [base + p_of * R + r_of, base + p_of * R + r_of]; ci += [f_idx, l_idx]
    lo += [bd["prack_lo"]] * (P * R); hi += [bd["prack_hi"]] * (P * R); base += P * R
    ri = np.concatenate(ri); ci = np.concatenate(ci)
    A = sparse.csr_matrix((np.ones(len(ri)), (ri, ci)), shape=(base, n))
    return A, np.array(lo, dtype=float), np.array(hi, dtype=float)


@dataclass
class ExactResult:
    status: str  # "optimal" | "infeasible" | "time_limit" | "error"
    objective: Optional[int]
    assign: Optional[np.ndarray]  # [P, RF] uint16, leader first, followers ascending id
    seconds: float
    x: Optional[np.ndarray] = None


def decode_solution(topic: Topic, x: np.ndarray) -> np.ndarray:
    """0/1 vector -> compact [P, RF]: leader first (README.md:153-156: `_l` = leader),
    retained followers in their current order, new followers in ascending broker index."""
    B, P, RF = topic.n_brokers, topic.n_partitions, topic.rf
    xr = np.rint(x).astype(np.int64).reshape(B, P, 2)
    out = np.full((P, RF), NONE, dtype=np.uint16)
    for p in range(P):
        leaders = np.nonzero(xr[:, p, 1])[0]
        fol = [int(b) for b in np.nonzero(xr[:, p, 0])[0]]
        cur = [int(v) for v in topic.current[p] if int(v) != NONE]
        kept = [b for b in cur if b in fol]
        new = sorted(b for b in fol if b not in kept)
        seq = [int(leaders[0])] if len(leaders) else []
        seq += kept + new
        out[p, : len(seq[:RF])] = seq[:RF]
    return out


def solve_exact(topic: Topic, time_limit: float = 120.0, extra_cuts=None) -> ExactResult:
    """Exact optimum of the README model via HiGHS branch-and-bound (lp_solve substitute)."""
    import time

    from scipy.optimize import Bounds, LinearConstraint, milp

    A, lo, hi = _sparse_model(topic)
    c = objective_vector(topic).astype(float)
    cons = [LinearConstraint(A, lo, hi)]
    if extra_cuts:
        for coef, clo, chi in extra_cuts:
            cons.append(LinearConstraint(coef.reshape(1, -1), clo, chi))
    t0 = time.perf_counter()
    res = milp(-c, constraints=cons, integrality=np.ones(len(c)), bounds=Bounds(0, 1),
               options={"time_limit": time_limit, "mip_rel_gap": 0.0})
    dt = time.perf_counter() - t0
    if res.status == 0 and res.x is not None:
        obj = int(round(-res.fun))
        return ExactResult("optimal", obj, decode_solution(topic, res.x), dt, np.rint(res.x))
    if res.status == 2:
        return ExactResult("infeasible", None, None, dt)
    if res.status == 1:
        return ExactResult("time_limit", None, None, dt)
    return ExactResult("error", None, None, dt)


def solve_exact_capped(topics: Sequence[Topic], replica_cap: Sequence[int], time_limit: float = 300.0):
    """Exact optimum of SEVERAL topics under cluster-wide per-broker load caps (BASELINE config 5): the README model of
    every topic (README.md:144-185; block diagonal, every variable and row carries its topic prefix) plus one coupling row
    per capped broker b over all topics:  sum_t sum_p (t<t>b<b>p<p> + t<t>b<b>p<p>_l) <= replica_cap[b]   (-1 = no cap).
    Returns (status, total objective, [assign per topic])."""
    from scipy import sparse
    from scipy.optimize import Bounds, LinearConstraint, milp

    B = topics[0].n_brokers
    blocks, los, his, cs, offs = [], [], [], [], [0]
    for t in topics:
        assert t.n_brokers == B
        A, lo, hi = _sparse_model(t)
        blocks.append(A); los.append(lo); his.append(hi); cs.append(objective_vector(t).astype(float))
        offs.append(offs[-1] + A.shape[1])
    A = sparse.block_diag(blocks, format="csr")
    rows, cols = [], []
    capped = [b for b in range(B) if replica_cap[b] >= 0]
    for ri, b in enumerate(capped):
        for ti, t in enumerate(topics):
            for p in range(t.n_partitions):
                for leader in (False, True):
                    rows.append(ri); cols.append(offs[ti] + var_index(t, b, p, leader))
    cons = [LinearConstraint(A, np.concatenate(los), np.concatenate(his))]
    if capped:
        C = sparse.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(len(capped), offs[-1]))
        cons.append(LinearConstraint(C, -np.inf, np.array([replica_cap[b] for b in capped], dtype=float)))
    c = np.concatenate(cs)
    res = milp(-c, constraints=cons, integrality=np.ones(len(c)), bounds=Bounds(0, 1), options={"time_limit": time_limit, "mip_rel_gap": 0.0})
    if res.status == 2:
        return "infeasible", None, None
    if res.status != 0 or res.x is None:
        return "time_limit", None, None
    assigns = [decode_solution(t, res.x[offs[i]:offs[i + 1]]) for i, t in enumerate(topics)]
    return "optimal", int(round(-res.fun)), assigns


def lp_bound(topic: Topic, method: str = "highs") -> Optional[float]:
    """Value of the LP relaxation (an upper bound on the 0-1 optimum).  method: "highs" (HiGHS chooses, dual simplex here) or
    "highs-ipm" (interior point + crossover: the only one that finishes on the 400 x 3000 topic within the hour)."""
    from scipy.optimize import linprog

    A, lo, hi = _sparse_model(topic)
    c = objective_vector(topic).astype(float)
    eq = lo == hi
    fin_lo = np.isfinite(lo) & ~eq
    fin_hi = np.isfinite(hi) & ~eq
    from scipy import sparse

    A_ub = sparse.vstack([A[fin_hi], -A[fin_lo]])
    b_ub = np.concatenate([hi[fin_hi], -lo[fin_lo]])
    res = linprog(-c, A_ub=A_ub, b_ub=b_ub, A_eq=A[eq], b_eq=lo[eq], bounds=(0, 1), method=method)
    if res.status != 0:
        return None
    return float(-res.fun)


def is_unique_optimum(topic: Topic, res: ExactResult, time_limit: float = 60.0) -> bool:
    """True iff no OTHER 0/1 point attains res.objective (no-good cut + re-solve)."""
    x = res.x
    ones = x > 0.5
    coef = np.where(ones, 1.0, -1.0)
    # sum_{x_i=1} x_i - sum_{x_i=0} x_i <= |ones| - 1   excludes exactly x
    cut = (coef, -np.inf, float(ones.sum() - 1))
    res2 = solve_exact(topic, time_limit, extra_cuts=[cut])
    if res2.status == "infeasible":
        return True
    return res2.status == "optimal" and res2.objective < res.objective


def brute_force(topic: Topic) -> Tuple[Optional[int], int]:
    """Exhaustive optimum over compact candidates (tiny instances only).  Returns
    (best objective or None if infeasible, number of optimal compact candidates counted
    with followers as an unordered set)."""
    B, P, RF = topic.n_brokers, topic.n_partitions, topic.rf
    per_part = []
    for leader in range(B):
        others = [b for b in range(B) if b != leader]
        for fol in itertools.combinations(others, RF - 1):
            per_part.append((leader,) + fol)
    if len(per_part) ** P > 5_000_000:
        raise ValueError("instance too large for brute force")
    best, n_best = None, 0
    for combo in itertools.product(per_part, repeat=P):
        a = np.array(combo, dtype=np.int64)
        obj, viol = verify(topic, a)
        if viol[0] != 0:
            continue
        if best is None or obj > best:
            best, n_best = obj, 1
        elif obj == best:
            n_best += 1
    return best, n_best


def provably_infeasible(topic: Topic) -> List[str]:
    """Necessary conditions of the README model that can be checked by counting: if any fails, no 0/1
    point satisfies the rows and lp_solve would print "This problem is infeasible".  Returns the failed conditions
    (empty list = not provably infeasible; the instance may still be infeasible for subtler reasons)."""
    B, R, P, RF = topic.n_brokers, topic.n_racks, topic.n_partitions, topic.rf
    bd = topic.bounds()
    n = P * RF
    rs = [0] * R
    for r in topic.rack_of:
        rs[int(r)] += 1
    why = []
    if RF > B:
        why.append("rf > brokers")
    if not B * bd["rep_lo"] <= n <= B * bd["rep_hi"]:
        why.append("replicas per broker band cannot hold P*RF replicas")
    if not B * bd["lead_lo"] <= P <= B * bd["lead_hi"]:
        why.append("leaders per broker band cannot hold P leaders")
    if not R * bd["rack_lo"] <= n <= R * bd["rack_hi"]:
        why.append("replicas per rack band cannot hold P*RF replicas")
    caps_hi = [min(bd["rack_hi"], rs[r] * bd["rep_hi"], P * min(bd["prack_hi"], rs[r])) for r in range(R)]
    caps_lo = [max(bd["rack_lo"], rs[r] * bd["rep_lo"], P * bd["prack_lo"]) for r in range(R)]
    for r in range(R):
        if caps_lo[r] > caps_hi[r]:
            why.append(f"rack {r}: needs at least {caps_lo[r]} replicas but can hold at most {caps_hi[r]}")
    if sum(caps_hi) < n:
        why.append("rack capacities sum below P*RF")
    if sum(caps_lo) > n:
        why.append("rack floors sum above P*RF")
    if sum(min(bd["prack_hi"], x) for x in rs) < RF:
        why.append("a partition cannot spread RF replicas over the racks")
    if R * bd["prack_lo"] > RF or any(bd["prack_lo"] > x for x in rs):
        why.append("per-partition rack floor cannot be met")
    return why


def upper_bound_simple(topic: Topic) -> int:
    """Combinatorial bound: each partition keeps its best RF surviving replicas in their best
    roles, ignoring every coupling constraint (C3, C4, C6, C7)."""
    w = topic.weights
    total = 0
    for p in range(topic.n_partitions):
        lead_alive = int(topic.current[p, 0]) != NONE
        n_fol = sum(1 for v in topic.current[p, 1:] if int(v) != NONE)
        best = 0
        # enumerate: who leads (current leader / a current follower / a new broker)
        options = []
        if lead_alive:
            options.append((w[0][0], n_fol, False))
        if n_fol:
            options.append((w[1][0], n_fol - 1, lead_alive))
        options.append((0, n_fol, lead_alive))
        for lead_gain, fol_avail, old_leader_as_follower in options:
            slots = topic.rf - 1
            gains = [w[1][1]] * fol_avail + ([w[0][1]] if old_leader_as_follower else [])
            gains.sort(reverse=True)
            best = max(best, lead_gain + sum(g for g in gains[:slots] if g > 0))
        total += best
    return total


# --------------------------------------------------------------------------------------
# Canonical tie-break: reproduces README.md:88 `[8,1]` on KAT-1
# --------------------------------------------------------------------------------------
def canonicalize(topic: Topic, assign: np.ndarray) -> np.ndarray:
    """Deterministic polish among equal-objective feasible solutions: scanning partitions
    and slots in order, every NEWLY placed replica (not a current replica of its partition)
    is moved to the lowest broker index that keeps the candidate feasible with the same
    objective; repeated to a fixpoint.  Then followers are ordered retained-first (current
    order), new ones ascending."""
    P, RF, B = topic.n_partitions, topic.rf, topic.n_brokers
    a = np.asarray(assign).reshape(P, RF).astype(np.int64).copy()
    obj0, v0 = verify(topic, a)
    if v0[0] != 0:
        return a.astype(np.uint16)
    changed = True
    while changed:
        changed = False
        for p in range(P):
            cur = {int(v) for v in topic.current[p] if int(v) != NONE}
            for k in range(RF):
                b = int(a[p, k])
                if b in cur:
                    continue
                for nb in range(b):
                    if nb in a[p] or nb in cur:
                        continue
                    a[p, k] = nb
                    obj, v = verify(topic, a)
                    if v[0] == 0 and obj == obj0:
                        changed = True
                        break
                    a[p, k] = b
    for p in range(P):
        cur = [int(v) for v in topic.current[p] if int(v) != NONE]
        fol = [int(v) for v in a[p, 1:]]
        kept = [b for b in cur if b in fol]
        new = sorted(b for b in fol if b not in kept)
        a[p, 1:] = kept + new
    return a.astype(np.uint16)


# --------------------------------------------------------------------------------------
# lp_solve LP-format text (README.md:144-185) and a reader for our own output
# --------------------------------------------------------------------------------------
def var_name(topic: Topic, t_index: int, b: int, p: int, leader: bool) -> str:
    """`t<topic>b<broker id>p<partition>[_l]` (README.md:146, README.md:184)."""
    pid = p if topic.partition_ids is None else int(topic.partition_ids[p])
    return f"t{t_index}b{int(topic.broker_ids[b])}p{pid}" + ("_l" if leader else "")


def write_lp(topic: Topic, t_index: int = 1) -> str:
    B, P = topic.n_brokers, topic.n_partitions
    names = [None] * (2 * B * P)
    for b in range(B):
        for p in range(P):
            names[var_index(topic, b, p, False)] = var_name(topic, t_index, b, p, False)
            names[var_index(topic, b, p, True)] = var_name(topic, t_index, b, p, True)
    c = objective_vector(topic)
    out = ["// Optimization function, based on current assignment"]
    terms = [f"{int(c[i])} {names[i]}" for i in range(len(c)) if c[i] != 0]
    out.append("max: " + (" + ".join(terms) if terms else "0") + ";")
    titles = {
        "C1": "Constrain on replication factor for every partition",
        "C2": "Constraint on having one and only one leader per partition",
        "C3": "Constraint on min/max replicas per broker",
        "C4": "Constraint on min/max leaders per broker",
        "C5": "Constraint on no leader and replicas on the same broker",
        "C6": "Constrain on min/max total replicas per racks",
        "C7": "Constrain on min/max replicas per partitions per racks",
    }
    last = None
    for fam, coefs, lo, hi in build_rows(topic):
        if fam != last:
            out.append("")
            out.append("// " + titles[fam])
            last = fam
        lhs = " + ".join(names[i] for i in sorted(coefs))
        if lo is not None and hi is not None and lo == hi:
            out.append(f"{lhs} = {lo};")
        else:
            if hi is not None:
                out.append(f"{lhs} <= {hi};")
            if lo is not None:
                out.append(f"{lhs} >= {lo};")
    out.append("")
    out.append("// All variables are binary")
    out.append("bin")
    out.append(", ".join(names) + ";")
    return "\n".join(out) + "\n"


def solve_lp_text(text: str, time_limit: float = 120.0) -> Tuple[str, Optional[int], Dict[str, int]]:
    """Parse the subset of lp_solve LP format that write_lp emits and solve it with HiGHS.
    Returns (status, objective, {var name: value})."""
    from scipy import sparse
    from scipy.optimize import Bounds, LinearConstraint, milp

    body = re.sub(r"//[^\n]*", "", text)
    stmts = [s.strip() for s in body.split(";") if s.strip()]
    names: Dict[str, int] = {}
    bin_stmt = [s for s in stmts if s.startswith("bin")]
    for nm in bin_stmt[0][3:].replace("\n", " ").split(","):
        names[nm.strip()] = len(names)
    n = len(names)
    c = np.zeros(n)
    ri, ci, vv, lo, hi = [], [], [], [], []

    def parse_lin(s):
        d = {}
        for term in s.split("+"):
            parts = term.split()
            if not parts:
                continue
            if len(parts) == 2:
                d[names[parts[1]]] = d.get(names[parts[1]], 0) + float(parts[0])
            elif parts[0] != "0":
                d[names[parts[0]]] = d.get(names[parts[0]], 0) + 1.0
        return d

    row = 0
    for s in stmts:
        if s.startswith("bin"):
            continue
        if s.startswith("max:"):
            for i, v in parse_lin(s[4:]).items():
                c[i] = v
            continue
        m = re.match(r"(.*?)(<=|>=|=)\s*(-?\d+)\s*$", s, re.S)
        d = parse_lin(m.group(1))
        rhs = float(m.group(3))
        for i, v in d.items():
            ri.append(row); ci.append(i); vv.append(v)
        lo.append(rhs if m.group(2) in (">=", "=") else -np.inf)
        hi.append(rhs if m.group(2) in ("<=", "=") else np.inf)
        row += 1
    A = sparse.csr_matrix((vv, (ri, ci)), shape=(row, n))
    res = milp(-c, constraints=[LinearConstraint(A, np.array(lo), np.array(hi))],
               integrality=np.ones(n), bounds=Bounds(0, 1),
               options={"time_limit": time_limit, "mip_rel_gap": 0.0})
    if res.status == 2:
        return "infeasible", None, {}
    if res.status != 0:
        return "error", None, {}
    vals = {nm: int(round(res.x[i])) for nm, i in names.items()}
    return "optimal", int(round(-res.fun)), vals


# --------------------------------------------------------------------------------------
# Reassignment JSON (README.md:52-63) <-> Topic
# --------------------------------------------------------------------------------------
def topics_from_json(doc: dict, broker_list: Sequence[int], racks: Dict[int, str],
                     rf: Optional[int] = None, weights=DEFAULT_WEIGHTS) -> List[Topic]:
    """`{"version":1,"partitions":[{"topic","partition","replicas"}]}` -> one Topic per
    topic name (sorted), brokers = `broker_list` order, racks densely numbered by sorted
    rack name."""
    brokers = [int(b) for b in broker_list]
    dense = {b: i for i, b in enumerate(brokers)}
    rack_names = sorted({str(racks[b]) for b in brokers})
    rack_idx = {r: i for i, r in enumerate(rack_names)}
    rack_of = np.array([rack_idx[str(racks[b])] for b in brokers], dtype=np.uint8)
    by_topic: Dict[str, List[dict]] = {}
    for e in doc["partitions"]:
        by_topic.setdefault(e["topic"], []).append(e)
    out = []
    for name in sorted(by_topic):
        parts = sorted(by_topic[name], key=lambda e: e["partition"])
        rf_cur = max(len(e["replicas"]) for e in parts)
        cur = np.full((len(parts), rf_cur), NONE, dtype=np.uint16)
        for i, e in enumerate(parts):
            for k, b in enumerate(e["replicas"]):
                cur[i, k] = dense.get(int(b), NONE)
        out.append(Topic(name=name, broker_ids=np.array(brokers, dtype=np.int32), rack_of=rack_of,
                         n_racks=len(rack_names), n_partitions=len(parts),
                         rf=rf if rf else rf_cur, current=cur, weights=weights,
                         partition_ids=np.array([e["partition"] for e in parts], dtype=np.int32)))
    return out


def assignment_to_json(topics: Sequence[Topic], assigns: Sequence[np.ndarray]) -> dict:
    parts = []
    for t, a in zip(topics, assigns):
        a = np.asarray(a).reshape(t.n_partitions, t.rf)
        for p in range(t.n_partitions):
            pid = p if t.partition_ids is None else int(t.partition_ids[p])
            parts.append({"topic": t.name, "partition": pid,
                          "replicas": [int(t.broker_ids[int(b)]) for b in a[p]]})
    return {"version": 1, "partitions": parts}


# --------------------------------------------------------------------------------------
# Instances: KAT-1 (README.md:25-91) and the BASELINE.json synthetic configs
# --------------------------------------------------------------------------------------
README_CURRENT = {"version": 1, "partitions": [
    {"topic": "x.y.z.t", "partition": 0, "replicas": [7, 18]},
    {"topic": "x.y.z.t", "partition": 1, "replicas": [8, 19]},
    {"topic": "x.y.z.t", "partition": 2, "replicas": [9, 10]},
    {"topic": "x.y.z.t", "partition": 3, "replicas": [0, 11]},
    {"topic": "x.y.z.t", "partition": 4, "replicas": [1, 12]},
    {"topic": "x.y.z.t", "partition": 5, "replicas": [2, 13]},
    {"topic": "x.y.z.t", "partition": 6, "replicas": [3, 14]},
    {"topic": "x.y.z.t", "partition": 7, "replicas": [4, 15]},
    {"topic": "x.y.z.t", "partition": 8, "replicas": [5, 16]},
    {"topic": "x.y.z.t", "partition": 9, "replicas": [6, 17]},
]}  # README.md:52-63


def readme_example() -> Topic:
    """KAT-1: 20 brokers, 2 AZ (odd -> b, even -> a: README.md:27-29), topic x.y.z.t
    10 partitions RF 2 (README.md:31), remove broker 19 (README.md:43-48)."""
    racks = {b: ("a" if b % 2 == 0 else "b") for b in range(20)}
    return topics_from_json(README_CURRENT, list(range(19)), racks)[0]


def splitmix64(state: int) -> Tuple[int, int]:
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


class _Rng:
    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.s, z = splitmix64(self.s)
        return z

    def below(self, n: int) -> int:
        return self.next() % n

    def sample(self, items: Sequence[int], k: int) -> List[int]:
        pool = list(items)
        out = []
        for _ in range(k):
            out.append(pool.pop(self.below(len(pool))))
        return out


def balanced_fill(n_brokers: int, n_racks: int, n_partitions: int, rf: int, t: int,
                  rack_of: Sequence[int]) -> np.ndarray:
    """Deterministic rack-aware balanced start: slot k of partition p goes to
    rack (t+p+k) mod R, on the least-loaded broker of that rack not already in p (leaders by
    (leaders, replicas, id); followers by (replicas, id))."""
    members = [[b for b in range(n_brokers) if rack_of[b] == r] for r in range(n_racks)]
    cnt_r = [0] * n_brokers
    cnt_l = [0] * n_brokers
    out = np.full((n_partitions, rf), NONE, dtype=np.uint16)
    for p in range(n_partitions):
        used = set()
        for k in range(rf):
            r = (t + p + k) % n_racks
            cand = [b for b in members[r] if b not in used]
            if k == 0:
                b = min(cand, key=lambda x: (cnt_l[x], cnt_r[x], x))
                cnt_l[b] += 1
            else:
                b = min(cand, key=lambda x: (cnt_r[x], x))
            cnt_r[b] += 1
            used.add(b)
            out[p, k] = b
    return out


@dataclass
class ClusterCase:
    """A multi-topic instance: shared target broker set + one Topic per topic."""

    name: str
    topics: List[Topic]
    removed: List[int]
    added: List[int]


def make_cluster(name: str, n_brokers0: int, n_racks: int, n_topics: int, n_partitions: int,
                 rf: int, removed: Sequence[int], added: Sequence[Tuple[int, int]],
                 weights=DEFAULT_WEIGHTS, bounds_override=None, new_rf: Optional[int] = None) -> ClusterCase:
    """Old cluster: brokers 0..n_brokers0-1, rack(b) = b mod R (generalises README.md:28-29);
    every topic balanced_fill'ed on it; then `removed` ids leave and `added` (id, rack)
    join.  Target broker list = survivors ascending, then added."""
    rack0 = [b % n_racks for b in range(n_brokers0)]
    removed = sorted(set(int(b) for b in removed))
    target = [b for b in range(n_brokers0) if b not in set(removed)] + [int(b) for b, _ in added]
    rack_t = [rack0[b] for b in range(n_brokers0) if b not in set(removed)] + [int(r) for _, r in added]
    dense = {b: i for i, b in enumerate(target)}
    topics = []
    for t in range(n_topics):
        cur_old = balanced_fill(n_brokers0, n_racks, n_partitions, rf, t, rack0)
        cur = np.full_like(cur_old, NONE)
        for p in range(n_partitions):
            for k in range(rf):
                cur[p, k] = dense.get(int(cur_old[p, k]), NONE)
        topics.append(Topic(name=f"topic-{t:04d}", broker_ids=np.array(target, dtype=np.int32),
                            rack_of=np.array(rack_t, dtype=np.uint8), n_racks=n_racks,
                            n_partitions=n_partitions, rf=new_rf or rf, current=cur, weights=weights,
                            bounds_override=dict(bounds_override or {})))
    return ClusterCase(name, topics, removed, [int(b) for b, _ in added])


CONFIG_SEED = 0x4B414F00


def gen_config(n: int, n_topics: Optional[int] = None) -> ClusterCase:
    """The five BASELINE.json configs.  `n_topics` truncates (for bounded test sizes)."""
    rng = _Rng(CONFIG_SEED + n)
    if n == 1:
        return ClusterCase("cfg1-readme", [readme_example()], [19], [])
    if n == 2:  # 100 brokers, 4 racks, 1 topic x 256 partitions RF 3, remove 1 broker
        rm = [rng.below(100)]
        return make_cluster("cfg2", 100, 4, n_topics or 1, 256, 3, rm, [])
    if n == 3:  # 200 brokers, 6 racks, 50 topics x 64 partitions RF 3, add 20 brokers
        add = [(b, b % 6) for b in range(200, 220)]
        return make_cluster("cfg3", 200, 6, n_topics or 50, 64, 3, [], add)
    if n == 4:  # 500 brokers, 10 racks, 200 topics x 50 partitions, rolling replace 50
        rm, add = [], []
        nid = 500
        for r in range(10):
            members = [b for b in range(500) if b % 10 == r]
            for b in rng.sample(members, 5):
                rm.append(b)
                add.append((nid, r))
                nid += 1
        return make_cluster("cfg4", 500, 10, n_topics or 200, 50, 3, rm, add)
    if n == 5:  # 1000 brokers, 20 racks, 100k partitions as 1000 x 100, mixed add+remove, caps
        rm = rng.sample(list(range(1000)), 50)
        add = [(1000 + i, rng.below(20)) for i in range(50)]
        P, RF, B = 100, 3, 1000
        cap = -((-P * RF) // B) + 1  # per-broker load cap = ceil(avg)+1
        return make_cluster("cfg5", 1000, 20, n_topics or 1000, P, RF, rm, add,
                            bounds_override={"rep_hi": cap})
    raise ValueError(n)


def random_case(seed: int, max_b: int = 12, max_p: int = 8) -> Topic:
    """Small random instance with forced rebalancing (brokers removed AND added, possibly an
    RF change), for cross-checks against the exact solver.  May be infeasible."""
    rng = _Rng(0xC0FFEE00 + seed)
    R = 1 + rng.below(3)
    B0 = max(R * 2, 3 + rng.below(max_b - 2))
    P = 1 + rng.below(max_p)
    rf = 1 + rng.below(min(3, B0 - 1))
    n_rm = rng.below(max(1, B0 // 3))
    n_add = rng.below(3)
    rm = rng.sample(list(range(B0)), n_rm)
    add = [(B0 + i, rng.below(R)) for i in range(n_add)]
    new_rf = rf
    if rng.below(4) == 0:
        new_rf = max(1, min(rf + (1 if rng.below(2) else -1), B0 - n_rm + n_add - 1))
    c = make_cluster(f"rand{seed}", B0, R, 1, P, rf, rm, add, new_rf=new_rf)
    t = c.topics[0]
    # scramble part of the start so that the balanced fill is not already optimal
    cur = t.current.copy()
    for _ in range(rng.below(P + 1)):
        p = rng.below(P)
        k = rng.below(cur.shape[1])
        nb = rng.below(t.n_brokers)
        if nb not in cur[p]:
            cur[p, k] = nb
    t.current = cur
    return t


def random_case_wide(seed: int, max_b: int = 60, max_p: int = 96) -> Topic:
    """Wider random family than random_case: up to 8 racks of uneven size (added brokers land in random
    racks), RF up to 4 with RF changes, random objective weights (README.md:116-120 allows any scheme with
    the leader weight largest), heavier scrambling of the start.  May be infeasible."""
    rng = _Rng(0x51DE0000 + seed)
    R = 1 + rng.below(8)
    B0 = max(R + 2, 6 + rng.below(max_b - 5))
    P = 4 + rng.below(max_p - 3)
    rf = 1 + rng.below(min(4, B0 - 1))
    n_rm = rng.below(max(1, B0 // 3))
    n_add = rng.below(1 + B0 // 6)
    rm = rng.sample(list(range(B0)), n_rm)
    add = [(B0 + i, rng.below(R)) for i in range(n_add)]
    new_rf = rf
    if rng.below(3) == 0:
        new_rf = max(1, min(4, rf + (1 if rng.below(2) else -1), B0 - n_rm + n_add - 1))
    weights = DEFAULT_WEIGHTS
    k = rng.below(4)
    if k == 1:
        weights = ((4, 2), (2, 1))      # the README's prose scheme (README.md:116-120)
    elif k == 2:
        ll = 3 + rng.below(6)
        weights = ((ll, rng.below(ll)), (rng.below(ll), 1 + rng.below(ll - 1)))
    c = make_cluster(f"wide{seed}", B0, R, 1, P, rf, rm, add, weights=weights, new_rf=new_rf)
    t = c.topics[0]
    cur = t.current.copy()
    for _ in range(rng.below(2 * P + 1)):
        p = rng.below(P)
        kk = rng.below(cur.shape[1])
        nb = rng.below(t.n_brokers)
        if nb not in cur[p]:
            cur[p, kk] = nb
    t.current = cur
    return t


def random_case_rf(seed: int, max_b: int = 26, max_p: int = 24) -> Topic:
    """Random instances with HIGH replication factors (5..8 replicas, RF changes across the 4 / 5 boundary included):
    README.md:148-151 puts no cap on the replication factor and README.md:9 lists an RF change as a use case.  May be
    infeasible."""
    rng = _Rng(0x8F000000 + seed)
    R = 1 + rng.below(6)
    rf = 4 + rng.below(5)                       # current RF 4..8
    B0 = max(R * 2, rf + 2 + rng.below(max_b - rf))
    P = 2 + rng.below(max_p - 1)
    n_rm = rng.below(max(1, B0 // 5))
    n_add = rng.below(3)
    rm = rng.sample(list(range(B0)), n_rm)
    add = [(B0 + i, rng.below(R)) for i in range(n_add)]
    new_rf = rf
    k = rng.below(3)
    if k == 1:
        new_rf = rf + 1
    elif k == 2:
        new_rf = rf - 1
    new_rf = max(5 if rf <= 5 and k != 2 else 3, min(8, new_rf, B0 - n_rm + n_add - 1))
    if rf <= 4 and new_rf <= 4:
        new_rf = 5
    weights = DEFAULT_WEIGHTS if rng.below(3) else ((4, 2), (2, 1))
    c = make_cluster(f"rf{seed}", B0, R, 1, P, rf, rm, add, weights=weights, new_rf=new_rf)
    t = c.topics[0]
    cur = t.current.copy()
    for _ in range(rng.below(P + 1)):
        p = rng.below(P)
        kk = rng.below(cur.shape[1])
        nb = rng.below(t.n_brokers)
        if nb not in cur[p]:
            cur[p, kk] = nb
    t.current = cur
    return t


def topic_to_dict(t: Topic) -> dict:
    return {"name": t.name, "broker_ids": [int(x) for x in t.broker_ids],
            "rack_of": [int(x) for x in t.rack_of], "n_racks": t.n_racks,
            "n_partitions": t.n_partitions, "rf": t.rf,
            "current": [[int(x) for x in row] for row in t.current],
            "weights": [list(w) for w in t.weights], "bounds_override": dict(t.bounds_override),
            **({"broker_w": [int(x) for x in t.broker_w]} if t.broker_w is not None else {}),
            **({"broker_wl": [int(x) for x in t.broker_wl]} if t.broker_wl is not None else {})}


def topic_from_dict(d: dict) -> Topic:
    return Topic(name=d["name"], broker_ids=np.array(d["broker_ids"], dtype=np.int32),
                 rack_of=np.array(d["rack_of"], dtype=np.uint8), n_racks=int(d["n_racks"]),
                 n_partitions=int(d["n_partitions"]), rf=int(d["rf"]),
                 current=np.array(d["current"], dtype=np.uint16).reshape(int(d["n_partitions"]), -1),
                 weights=tuple(tuple(w) for w in d["weights"]),
                 bounds_override=dict(d.get("bounds_override", {})),
                 broker_w=np.array(d["broker_w"], dtype=np.int32) if "broker_w" in d else None,
                 broker_wl=np.array(d["broker_wl"], dtype=np.int32) if "broker_wl" in d else None)


if __name__ == "__main__":
    t = readme_example()
    r = solve_exact(t)
    print(r.status, r.objective, count_moves(t, r.assign), r.seconds)
    print(json.dumps(assignment_to_json([t], [canonicalize(t, r.assign)])))


def _partition_value(w, rf: int, lead_kept: bool, n_fol: int, leader_may_lead: bool = True) -> int:
    """Best objective one partition can collect from a kept set: its current leader (if kept) and n_fol
    kept current followers, at most rf replicas, exactly one leader (coupling rows ignored).  With
    leader_may_lead=False the current leader may only stay as a follower."""
    slots = rf - 1

    def fol_sum(n_ff: int, old_leader: bool) -> int:
        gains = [w[1][1]] * n_ff + ([w[0][1]] if old_leader else [])
        gains.sort(reverse=True)
        return sum(g for g in gains[:slots] if g > 0)

    best = fol_sum(n_fol, lead_kept)  # a new broker leads
    if lead_kept and leader_may_lead:
        best = max(best, w[0][0] + fol_sum(n_fol, False))
    if n_fol:
        best = max(best, w[1][0] + fol_sum(n_fol - 1, lead_kept))
    return best


def upper_bound_forced(topic: Topic) -> int:
    """Tighter combinatorial bound = upper_bound_simple minus the cheapest way to perform the evictions
    (and leader changes) that EVERY feasible assignment must perform.

    f_p(K) = best value partition p can collect from a kept subset K of its surviving current replicas
    (_partition_value).  Any assignment keeps some K_p per partition and scores at most sum_p f_p(K_p).
      * evictions: with s_b / s_r / s_(p,r) the surviving replicas per broker / rack / (partition, rack)
        cell, at least k = max(sum_b (s_b - rep_hi)+, sum_r (s_r - rack_hi)+, sum_cells (s - prack_hi)+,
        n_surv + sum_b (rep_lo - s_b)+ - P*RF, n_surv + sum_r (rack_lo - s_r)+ - P*RF) replicas cannot be
        kept (one eviction lowers one broker, one rack and one cell count by one; lower bands need arrivals,
        and only P*RF - kept slots can take them).  g_p(j) = f_p(all) - max_{|K| = n_p - j} f_p(K) is the
        cheapest loss of evicting j replicas of p; relaxing "which replicas" to "any", the total loss is
        at least the minimum of sum_p g_p(j_p) over sum_p j_p >= k, bounded below by the k smallest
        marginals of the lower convex envelopes of the g_p.
      * leader changes: brokers holding more current leaders than lead_hi force that many partitions to
        change leader; each costs at least f_p(all) - f_p(all, current leader not leading).
    The two losses may fall on the same partitions, so the larger one is subtracted."""
    B, R, P, RF = topic.n_brokers, topic.n_racks, topic.n_partitions, topic.rf
    w = topic.weights
    bd = topic.bounds()
    rack = [int(r) for r in topic.rack_of]
    s_b = [0] * B
    s_r = [0] * R
    lead_b = [0] * B
    cell_excess = 0
    n_surv = 0
    total = 0
    marginals: List[int] = []
    lead_losses: List[Tuple[int, int]] = []  # (broker, loss if this partition's leader stops leading)
    for p in range(P):
        cur = [int(v) for v in topic.current[p]]
        lead_alive = cur[0] != NONE
        fol = [b for b in cur[1:] if b != NONE]
        n_p = len(fol) + (1 if lead_alive else 0)
        n_surv += n_p
        cell: Dict[int, int] = {}
        for b in fol + ([cur[0]] if lead_alive else []):
            s_b[b] += 1
            s_r[rack[b]] += 1
            cell[rack[b]] = cell.get(rack[b], 0) + 1
        cell_excess += sum(max(0, c - bd["prack_hi"]) for c in cell.values())
        f_all = _partition_value(w, RF, lead_alive, len(fol))
        total += f_all
        # g_p(j): cheapest loss of evicting j replicas (followers are interchangeable)
        g = []
        for j in range(n_p + 1):
            best = -1
            for drop_lead in ((False, True) if lead_alive else (False,)):
                df = j - (1 if drop_lead else 0)
                if 0 <= df <= len(fol):
                    best = max(best, _partition_value(w, RF, lead_alive and not drop_lead, len(fol) - df))
            g.append(f_all - best)
        # lower convex envelope of (j, g[j]) -> non-decreasing marginals
        hull = [(0, g[0])]
        for j in range(1, n_p + 1):
            hull.append((j, g[j]))
            while len(hull) >= 3:
                (x0, y0), (x1, y1), (x2, y2) = hull[-3], hull[-2], hull[-1]
                if (y1 - y0) * (x2 - x0) >= (y2 - y0) * (x1 - x0):  # middle point on or above the chord
                    hull.pop(-2)
                else:
                    break
        for (x0, y0), (x1, y1) in zip(hull, hull[1:]):
            # integer-valued relaxation of the envelope: floor of the running total keeps it a lower bound
            for t in range(x0 + 1, x1 + 1):
                lo_prev = y0 + ((y1 - y0) * (t - 1 - x0)) // (x1 - x0)
                lo_now = y0 + ((y1 - y0) * (t - x0)) // (x1 - x0)
                marginals.append(lo_now - lo_prev)
        if lead_alive:
            lead_b[cur[0]] += 1
            # value when the current leader may stay only as a follower, or leaves
            alt = max(_partition_value(w, RF, False, len(fol)), _partition_value(w, RF, True, len(fol), leader_may_lead=False))
            lead_losses.append((cur[0], max(0, f_all - alt)))
    need_b = sum(max(0, bd["rep_lo"] - c) for c in s_b)
    need_r = sum(max(0, bd["rack_lo"] - c) for c in s_r)
    k = max(sum(max(0, c - bd["rep_hi"]) for c in s_b), sum(max(0, c - bd["rack_hi"]) for c in s_r), cell_excess,
            n_surv + need_b - P * RF, n_surv + need_r - P * RF, 0)
    marginals.sort()
    evict_loss = sum(marginals[:k])
    lead_loss = 0
    for b in range(B):
        ex = lead_b[b] - bd["lead_hi"]
        if ex > 0:
            lead_loss += sum(sorted(l for bb, l in lead_losses if bb == b)[:ex])
    return min(total - max(evict_loss, lead_loss), upper_bound_broker(topic))


def upper_bound_broker(topic: Topic) -> int:
    """Per-broker capacity bound with a global cap on leading survivors.

    A broker keeps at most rep_hi of its surviving replicas and at most lead_hi of them can be leading; a replica
    that leads is worth w[cur_role][0], one that follows w[cur_role][1] (one-leader-per-partition and rack rows
    relaxed).  v_b(L) = best value on broker b with at most L survivors leading.  Brokers with fewer than lead_lo
    survivors must receive lead_lo - s_b NEW leaders, so at most Lcap = min(#partitions with a survivor,
    P - sum_b (lead_lo - s_b)+) partitions keep a surviving replica as leader: the bound is
    sum_b v_b(0) + the Lcap largest marginals of the (upper concave envelopes of the) v_b.  Unlike the eviction
    bound it charges forced evictions AND forced leader changes together."""
    B, P = topic.n_brokers, topic.n_partitions
    w = topic.weights
    bd = topic.bounds()
    n_l = [0] * B
    n_f = [0] * B
    with_survivor = 0
    for p in range(P):
        alive = False
        for k in range(topic.rf_cur):
            b = int(topic.current[p, k])
            if b != NONE:
                alive = True
                if k == 0:
                    n_l[b] += 1
                else:
                    n_f[b] += 1
        with_survivor += alive
    lcap = min(with_survivor, P - sum(max(0, bd["lead_lo"] - (n_l[b] + n_f[b])) for b in range(B)))
    base = 0
    marginals: List[int] = []
    for b in range(B):
        lmax = min(bd["lead_hi"], bd["rep_hi"], n_l[b] + n_f[b])
        v = [-1] * (lmax + 1)
        for x in range(0, min(n_l[b], lmax) + 1):
            for y in range(0, min(n_f[b], lmax - x) + 1):
                val = x * w[0][0] + y * w[1][0]
                slots = bd["rep_hi"] - x - y
                gains = sorted([w[0][1]] * (n_l[b] - x) + [w[1][1]] * (n_f[b] - y), reverse=True)
                val += sum(g for g in gains[:slots] if g > 0)
                v[x + y] = max(v[x + y], val)
        for i in range(1, lmax + 1):  # "at most L leading"
            v[i] = max(v[i], v[i - 1])
        base += v[0]
        # upper concave envelope -> non-increasing marginals (a relaxation: envelope >= v)
        hull = [(0, v[0])]
        for i in range(1, lmax + 1):
            hull.append((i, v[i]))
            while len(hull) >= 3:
                (x0, y0), (x1, y1), (x2, y2) = hull[-3], hull[-2], hull[-1]
                if (y1 - y0) * (x2 - x0) <= (y2 - y0) * (x1 - x0):  # middle point on or below the chord
                    hull.pop(-2)
                else:
                    break
        for (x0, y0), (x1, y1) in zip(hull, hull[1:]):
            for t in range(x0 + 1, x1 + 1):  # ceil of the running total keeps it an upper bound
                up_prev = y0 + -((-(y1 - y0) * (t - 1 - x0)) // (x1 - x0))
                up_now = y0 + -((-(y1 - y0) * (t - x0)) // (x1 - x0))
                marginals.append(up_now - up_prev)
    marginals.sort(reverse=True)
    return base + sum(m for m in marginals[: max(0, lcap)] if m > 0)
