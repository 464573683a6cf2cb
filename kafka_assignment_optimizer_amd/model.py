"""Host-side instance model: reassignment JSON <-> compact per-topic arrays.

Input/output format is the one `kafka-reassign-partitions` prints and consumes
(README.md:52-63, README.md:67-78): ``{"version":1,"partitions":[{"topic":..,"partition":..,
"replicas":[..]}]}`` with replicas[0] the preferred leader.  The target broker set is the
``--broker-list`` CSV (README.md:48); the broker->rack map is given only in prose by the README
(README.md:27-29), so this package defines it as JSON ``{"<brokerId>": "<rack>"}``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

NONE = 0xFFFF
DEFAULT_WEIGHTS = ((4, 1), (2, 2))  # w[cur_role][new_role]; README.md:146 shows the multiset {1,2,2,4}
BOUND_KEYS = ("rep_lo", "rep_hi", "lead_lo", "lead_hi", "rack_lo", "rack_hi", "prack_lo", "prack_hi")


@dataclass
class Topic:
    name: str
    broker_ids: np.ndarray          # [B] external ids of the target brokers (dense index -> id)
    rack_of: np.ndarray             # [B] dense rack index
    n_racks: int
    n_partitions: int
    rf: int
    current: np.ndarray             # [P, rf_cur] uint16 dense index / NONE
    weights: tuple = DEFAULT_WEIGHTS
    partition_ids: Optional[np.ndarray] = None
    bounds_override: Dict[str, int] = field(default_factory=dict)
    broker_w: Optional[np.ndarray] = None    # [B] extra objective weight per replica on the broker (kao_topic.broker_w)
    broker_wl: Optional[np.ndarray] = None   # [B] ... per leader on the broker

    def __post_init__(self):
        self.broker_ids = np.ascontiguousarray(self.broker_ids, dtype=np.int32)
        self.rack_of = np.ascontiguousarray(self.rack_of, dtype=np.uint8)
        self.current = np.ascontiguousarray(self.current, dtype=np.uint16).reshape(self.n_partitions, -1)

    @property
    def n_brokers(self) -> int:
        return int(self.broker_ids.shape[0])

    @property
    def rf_cur(self) -> int:
        return int(self.current.shape[1])

    @classmethod
    def from_dict(cls, d: dict) -> "Topic":
        """A topic from its plain-JSON form (the layout of the fixtures under tests/golden/: broker_ids, rack_of, n_racks,
        n_partitions, rf, current rows, weights, bounds_override, optional broker_w / broker_wl)."""
        P = int(d["n_partitions"])
        return cls(name=str(d.get("name", "t")), broker_ids=np.array(d["broker_ids"], dtype=np.int32),
                   rack_of=np.array(d["rack_of"], dtype=np.uint8), n_racks=int(d["n_racks"]), n_partitions=P, rf=int(d["rf"]),
                   current=np.array(d["current"], dtype=np.uint16).reshape(P, -1),
                   weights=tuple(tuple(int(x) for x in w) for w in d.get("weights", DEFAULT_WEIGHTS)),
                   bounds_override=dict(d.get("bounds_override", {})),
                   broker_w=None if d.get("broker_w") is None else np.array(d["broker_w"], dtype=np.int32),
                   broker_wl=None if d.get("broker_wl") is None else np.array(d["broker_wl"], dtype=np.int32))


def topics_from_json(doc: dict, broker_list: Sequence[int], racks: Dict, rf: Optional[int] = None,
                     weights=DEFAULT_WEIGHTS) -> List[Topic]:
    brokers = [int(b) for b in broker_list]
    if len(set(brokers)) != len(brokers):
        raise ValueError("duplicate ids in broker list")
    racks = {int(k): str(v) for k, v in racks.items()}
    missing = [b for b in brokers if b not in racks]
    if missing:
        raise ValueError(f"no rack given for brokers {missing}")
    dense = {b: i for i, b in enumerate(brokers)}
    rack_names = sorted({racks[b] for b in brokers})
    rack_idx = {r: i for i, r in enumerate(rack_names)}
    rack_of = np.array([rack_idx[racks[b]] for b in brokers], dtype=np.uint8)
    by_topic: Dict[str, list] = {}
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
                         n_racks=len(rack_names), n_partitions=len(parts), rf=int(rf) if rf else rf_cur,
                         current=cur, weights=weights,
                         partition_ids=np.array([e["partition"] for e in parts], dtype=np.int32)))
    return out


def assignment_to_json(topics: Sequence[Topic], assigns: Sequence[np.ndarray]) -> dict:
    parts = []
    for t, a in zip(topics, assigns):
        a = np.asarray(a).reshape(t.n_partitions, t.rf)
        if a.size and int(a.max()) >= t.n_brokers:
            raise ValueError(f"topic {t.name}: assignment has empty / out-of-range slots (a NO_FEASIBLE or "
                             "INFEASIBLE_PROVEN result carries no plan)")
        for p in range(t.n_partitions):
            pid = p if t.partition_ids is None else int(t.partition_ids[p])
            parts.append({"topic": t.name, "partition": pid,
                          "replicas": [int(t.broker_ids[int(b)]) for b in a[p]]})
    return {"version": 1, "partitions": parts}
