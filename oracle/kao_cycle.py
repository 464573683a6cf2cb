"""KAO-CX (cyclic-exchange improvement) restated with numpy -- TEST INFRASTRUCTURE ONLY.

The reference has no counterpart (lp_solve returns the exact optimum, README.md:135-136); this is the oracle of the product's
own `kao_improve_cycles` / `kao_cycle_matrices` (include/kao.h): every definition below -- edge keys, tie-breaks, configuration
numbering, candidate order, realisation, merge rule -- is what the HIP kernels and the host driver must reproduce bit for bit.
Never imported by the product package.

One round, from a FEASIBLE assignment A of one topic (model: README.md:144-185):
  * transfer graphs on the brokers plus SLACK nodes -- one per rack, Z_r = B + r, and a global one, Z = B + R (round 4; until
    then a single node Z = B: on topics whose band has slack -- P*RF not a multiple of B, README.md:158-166 with hi > lo -- a
    path through it moved a replica unit from one rack to another, rows C6 (README.md:173-175) rejected the realisation, and the
    rack-conserving path of the same cost was never found: fixpoints 8-21 units below the optimum):
      F: edge u -> v = "some follower slot holding u takes v instead" (one replica unit moves u -> v), cheapest slot per pair;
      S: edge u -> v = "a partition led by u with follower v swaps the two roles" (one leader unit moves u -> v);
      L: edge u -> v = "a partition led by u gets leader v" -- nominally a replica unit AND a leader unit move u -> v; five
         variants (plain replacement; v enters and u stays as follower while another follower leaves; a follower is promoted
         and u leaves; role swap; replacement plus one follower replaced), each carrying the cost of the F path that
         compensates the difference between its replica effect and u -> v -- built AFTER the F closure;
      F: u -> Z_rack(u) when u may take one more replica inside its band (C3), Z_rack(v) -> v when v may give one up: a path through
         Z_r alone leaves every rack total as it is; Z_r -> Z when rack r may take one more inside ITS band (C6), Z -> Z_r when it
         may give one up;   S: u -> Z when u may lead one more partition (C4), Z -> v when v may give one up (leader units know no
         racks);   L: no slack edges;
  * bounded-hop closures by three min-plus squarings (paths of <= 8 edges), with the midpoint of every pair;
  * a negative diagonal entry is an improving cyclic exchange by itself; otherwise SEEDS are enumerated -- for every partition
    every new row that replaces at most two replicas (one of them by a current replica of the partition) and picks any
    leader -- and priced as  gain(seed) - cheapest closure of its replica imbalance (F) - of its leader imbalance (S), or of both
    at once through L;
  * candidates are realised (paths unrolled into slot changes, a partition may be used once), evaluated exactly, and the
    best one -- or a merge of partition-disjoint ones -- becomes the next assignment.
"""
from __future__ import annotations

import itertools
from typing import List, Optional, Tuple

import numpy as np

CINF = 1 << 17      # "no edge / no path"
CB = 1 << 16        # bias of the cost field inside an edge key
NO_SLOT = 0xFFFFFFFF
NO_EDGE = 0xFFFFFFFFFFFFFFFF
LEVELS = 3          # squarings: paths of <= 2**LEVELS edges
MAX_EVAL = 512      # realisations evaluated per round
BULK_SLOTS = 131072  # topics beyond this many replica slots: cycle candidates are merged BEFORE they are scored (round_step, bulk mode)
MAX_RF = 8


def supported(t) -> bool:
    return 2 <= t.rf <= MAX_RF and t.n_brokers + t.n_racks + 1 <= 2048


def n_cfg(rf: int, rf_cur: int) -> int:
    return (rf - 1) + rf * rf + (rf * (rf - 1) // 2) * rf_cur * rf


def _wt_tables(t):
    """WL[p, b], WF[p, b]: objective weight of broker b as leader / follower of partition p (README.md:145-146), plus the
    topic's broker weights (kao_topic.broker_w on both variables of a broker, broker_wl on the `_l` one) when it carries them."""
    P, B = t.n_partitions, t.n_brokers
    WL = np.zeros((P, B), dtype=np.int64)
    WF = np.zeros((P, B), dtype=np.int64)
    w = t.weights
    cur = np.asarray(t.current).astype(np.int64)
    for k in range(cur.shape[1]):
        cr = 0 if k == 0 else 1
        ok = cur[:, k] < B
        ps = np.nonzero(ok)[0]
        WL[ps, cur[ps, k]] = w[cr][0]
        WF[ps, cur[ps, k]] = w[cr][1]
    if getattr(t, "broker_w", None) is not None:
        WL += np.asarray(t.broker_w, dtype=np.int64)[None, :]
        WF += np.asarray(t.broker_w, dtype=np.int64)[None, :]
    if getattr(t, "broker_wl", None) is not None:
        WL += np.asarray(t.broker_wl, dtype=np.int64)[None, :]
    return WL, WF


class Round:
    """All intermediate objects of one round (the parity tests compare them with the device's)."""

    def __init__(self, t, A):
        self.t = t
        self.A = np.asarray(A).astype(np.int64).reshape(t.n_partitions, t.rf)
        self.B, self.P, self.RF, self.R = t.n_brokers, t.n_partitions, t.rf, t.n_racks
        self.n = self.B + self.R + 1      # brokers, one slack node per rack (B + r), the global slack node
        self.Z = self.B + self.R
        self.bd = t.bounds()
        self.rack = np.asarray(t.rack_of).astype(np.int64)
        self.cur = np.asarray(t.current).astype(np.int64)
        self.WL, self.WF = _wt_tables(t)
        self.c = np.bincount(self.A.reshape(-1), minlength=self.B)
        self.l = np.bincount(self.A[:, 0], minlength=self.B)
        self.K = np.bincount(self.rack[self.A.reshape(-1)], minlength=self.R)     # replicas per rack (C6)
        self._edges()
        self._closures()

    # ---- C7 of a row completed by one more broker: which brokers y may complete `base` (README.md:178-180) ----
    def _completions(self, base: List[int]) -> Optional[np.ndarray]:
        lo, hi = self.bd["prack_lo"], self.bd["prack_hi"]
        cnt = np.bincount(self.rack[base], minlength=self.R) if base else np.zeros(self.R, dtype=np.int64)
        if (cnt > hi).any():
            return None
        deficient = np.nonzero(cnt < lo)[0]
        if len(deficient) > 1:
            return None
        cy = cnt[self.rack]
        ok = cy + 1 <= hi
        if len(deficient) == 1:
            ok &= (self.rack == deficient[0]) & (cy + 1 >= lo)
        return ok

    def _edges(self):
        n, B, P, RF = self.n, self.B, self.P, self.RF
        EF = np.full((n, n), NO_EDGE, dtype=np.uint64)
        ES = np.full((n, n), NO_EDGE, dtype=np.uint64)
        EL = np.full((n, n), NO_EDGE, dtype=np.uint64)
        allb = np.arange(B)
        for p in range(P):
            row = [int(x) for x in self.A[p]]
            inrow = np.zeros(B, dtype=bool)
            inrow[row] = True
            for k in range(1, RF):
                u = row[k]
                ok = self._completions([row[j] for j in range(RF) if j != k])
                if ok is not None:
                    ok = ok & ~inrow
                    cost = int(self.WF[p, u]) - self.WF[p]
                    key = ((cost + CB).astype(np.uint64) << np.uint64(32)) | np.uint64(p * RF + k)
                    vs = allb[ok]
                    EF[u, vs] = np.minimum(EF[u, vs], key[ok])
                # role swap: leader row[0] <-> follower row[k]
                a, v = row[0], u
                cs = int(self.WL[p, a] + self.WF[p, v] - self.WL[p, v] - self.WF[p, a])
                ks = np.uint64(((cs + CB) << 32) | (p * RF + k))
                if ks < ES[a, v]:
                    ES[a, v] = ks
        zkey = np.uint64((CB << 32) | NO_SLOT)
        bs = np.arange(B)
        up, down = self.c < self.bd["rep_hi"], self.c > self.bd["rep_lo"]
        EF[bs[up], B + self.rack[up]] = zkey
        EF[B + self.rack[down], bs[down]] = zkey
        rs = np.arange(self.R)
        EF[B + rs[self.K < self.bd["rack_hi"]], self.Z] = zkey
        EF[self.Z, B + rs[self.K > self.bd["rack_lo"]]] = zkey
        ES[:B, self.Z][self.l < self.bd["lead_hi"]] = zkey
        ES[self.Z, :B][self.l > self.bd["lead_lo"]] = zkey
        self.EF, self.ES, self.EL = EF, ES, EL    # L has no slack edges: it moves two kinds of units at once

    @staticmethod
    def _dist0(E):
        D = np.where(E == np.uint64(NO_EDGE), CINF, (E >> np.uint64(32)).astype(np.int64) - CB).astype(np.int64)
        np.fill_diagonal(D, 0)
        return D

    @staticmethod
    def _square(D):
        """D'[i, j] = min(CINF, min_k D[i, k] + D[k, j]); mid = the minimising k, k = i preferred, then the lowest k."""
        n = D.shape[0]
        out = np.empty_like(D)
        mid = np.empty((n, n), dtype=np.int64)
        prio = np.arange(n, dtype=np.int64) + 1
        for i0 in range(0, n, 16):
            i1 = min(n, i0 + 16)
            s = D[i0:i1, :, None] + D[None, :, :]                      # [i, k, j]
            pr = np.broadcast_to(prio[None, :, None], s.shape).copy()
            for i in range(i0, i1):
                pr[i - i0, i, :] = 0
            comp = s * 4096 + pr
            km = comp.argmin(axis=1)
            mid[i0:i1] = km
            out[i0:i1] = np.minimum(CINF, np.take_along_axis(s, km[:, None, :], axis=1)[:, 0, :])
        return out, mid

    def _closures(self):
        self.DF = [self._dist0(self.EF)]
        self.DS = [self._dist0(self.ES)]
        self.MF: List[Optional[np.ndarray]] = [None]
        self.MS: List[Optional[np.ndarray]] = [None]
        for _ in range(LEVELS):
            d, m = self._square(self.DF[-1]); self.DF.append(d); self.MF.append(m)
            d, m = self._square(self.DS[-1]); self.DS.append(d); self.MS.append(m)
        self._edges_L()
        self.DL = [self._dist0L(self.EL)]
        self.ML: List[Optional[np.ndarray]] = [None]
        for _ in range(LEVELS):
            d, m = self._square(self.DL[-1]); self.DL.append(d); self.ML.append(m)

    @staticmethod
    def _dist0L(E):
        D = np.where(E == np.uint64(NO_EDGE), CINF, (E >> np.uint64(44)).astype(np.int64) - CB).astype(np.int64)
        np.fill_diagonal(D, 0)
        return D

    def _edges_L(self):
        """Generalised leader-transfer edges u -> v (the leader unit AND, nominally, a replica unit move u -> v); variants whose
        replica effect differs from u -> v carry the cost of the compensating F path (level-3 closure):
          0 plain   : v replaces u as leader
          1 demote  : v enters as leader, u stays as follower, follower slot k leaves        (+ F path u -> row[k])
          2 promote : follower slot k becomes leader, u leaves, y enters as follower           (+ F path y -> v)
          3 swap    : follower slot k becomes leader, u becomes follower                       (+ F path u -> v)
          4 double  : v replaces u as leader and follower slot k takes y (v or y a current replica of the partition) (+ F path y -> row[k])
        key = (cost + CB) << 44 | p << 20 | variant << 16 | k << 12 | y"""
        n, B, P, RF = self.n, self.B, self.P, self.RF
        EL = np.full((n, n), NO_EDGE, dtype=np.uint64)
        DF3 = self.DF[LEVELS]
        allb = np.arange(B)
        neg = any((np.diag(self.DF[lev])[:B] < 0).any() for lev in range(1, LEVELS + 1))
        def put(u, vs, cost, payload):
            cost = np.asarray(cost, dtype=np.int64)
            good = cost < CINF // 2
            if not good.any(): return
            key = ((cost + CB).astype(np.uint64) << np.uint64(44)) | payload.astype(np.uint64)
            vs = np.asarray(vs)[good]; key = key[good]
            EL[u, vs] = np.minimum(EL[u, vs], key)
        for p in range(P):
            row = [int(x) for x in self.A[p]]
            u = row[0]
            inrow = np.zeros(B, dtype=bool); inrow[row] = True
            w0 = self.row_weight(p, row)
            wl, wf = self.WL[p], self.WF[p]
            pb = np.uint64(p << 20)
            I = [int(b) for b in self.cur[p] if b < B and b not in row]
            fsum = sum(int(wf[b]) for b in row[1:])
            # 0 plain
            ok = self._completions(row[1:])
            if ok is not None:
                ok = ok & ~inrow
                vs = allb[ok]
                put(u, vs, int(wl[u]) - wl[vs], np.full(len(vs), int(pb), dtype=np.uint64))
            if neg:
                continue          # compensations are only priced on a closure without negative cycles
            for k in range(1, RF):
                b = row[k]; others = [row[j] for j in range(1, RF) if j != k]
                osum = sum(int(wf[x]) for x in others)
                # 1 demote: row' = (v; u, others)
                ok = self._completions([u] + others)
                if ok is not None and DF3[u, b] < CINF:
                    ok = ok & ~inrow
                    vs = allb[ok]
                    cost = w0 - (wl[vs] + int(wf[u]) + osum) + int(DF3[u, b])
                    put(u, vs, cost, np.full(len(vs), int(pb) | (1 << 16) | (k << 12), dtype=np.uint64))
                # 2 promote: v = row[k]; row' = (v; y, others)
                v = b
                ok = self._completions([v] + others)
                if ok is not None:
                    ok = ok & ~inrow
                    if ok.any():
                        cy = np.where(ok, w0 - (int(wl[v]) + wf + osum) + DF3[:B, v], 1 << 40)
                        y = int(cy.argmin())
                        put(u, [v], [int(cy[y])], np.array([int(pb) | (2 << 16) | (k << 12) | y], dtype=np.uint64))
                # 3 swap
                cs = int(wl[u] + wf[v] - wl[v] - wf[u])
                put(u, [v], [cs + int(DF3[u, v])], np.array([int(pb) | (3 << 16) | (k << 12)], dtype=np.uint64))
                # 4 double: new leader v (not in row) and follower slot k -> y; at least one of v, y a current replica
                for i in I:
                    # (a) y = i fixed, v generic
                    okv = self._completions(others + [i])
                    if okv is not None and DF3[i, b] < CINF:
                        okv = okv & ~inrow
                        okv[i] = False
                        vs = allb[okv]
                        cost = w0 - (wl[vs] + int(wf[i]) + osum) + int(DF3[i, b])
                        put(u, vs, cost, np.full(len(vs), int(pb) | (4 << 16) | (k << 12) | i, dtype=np.uint64))
                    # (b) v = i fixed, y generic
                    oky = self._completions(others + [i])
                    if oky is not None:
                        oky = oky & ~inrow
                        oky[i] = False
                        if oky.any():
                            cy = np.where(oky, w0 - (int(wl[i]) + wf + osum) + DF3[:B, b], 1 << 40)
                            y = int(cy.argmin())
                            put(u, [i], [int(cy[y])], np.array([int(pb) | (4 << 16) | (k << 12) | y], dtype=np.uint64))
        self.EL = EL
    # ---- candidates ----
    def cycle_candidates(self, all_levels: bool = False) -> List[Tuple[int, int, int, int]]:
        """(gain, layer, level, b) of the lowest level of each layer that has a negative diagonal entry -- of EVERY level that has
        one when `all_levels` (bulk mode, round 4: the longer walks of the higher levels often use other partitions than the
        short ones, and a bulk round keeps whatever is partition-disjoint)."""
        out = []
        for layer, Ds in ((0, self.DF), (1, self.DS), (2, self.DL)):
            for lev in range(1, LEVELS + 1):
                dg = np.diag(Ds[lev])[: self.B]
                if (dg < 0).any():
                    out += [(int(-dg[b]), layer, lev, int(b)) for b in np.nonzero(dg < 0)[0]]
                    if not all_levels:
                        break
        out.sort(key=lambda c: (-c[0], c[1], c[2], c[3]))
        return out

    def row_weight(self, p: int, row) -> int:
        return int(self.WL[p, row[0]] + sum(self.WF[p, b] for b in row[1:]))

    def seed_table(self) -> np.ndarray:
        """[P, n_cfg, 2] int64: (total, y + 4096 * option) of the best completion y of every configuration (total <= 0: none).
        option 1 = the replica imbalance of a one-replica seed is closed through L (leader replacements) instead of F: free when
        y replaces the leader AS leader (both units travel back together), plus a swap path r -> y when the leader stays."""
        P, RF, B = self.P, self.RF, self.B
        rfc = self.cur.shape[1]
        DF3, DS3, DL3 = self.DF[LEVELS], self.DS[LEVELS], self.DL[LEVELS]
        tab = np.zeros((P, n_cfg(RF, rfc), 2), dtype=np.int64)
        pairs = list(itertools.combinations(range(RF), 2))
        ys = np.arange(B)
        for p in range(P):
            row = [int(x) for x in self.A[p]]
            w0 = self.row_weight(p, row)
            inrow = np.zeros(B, dtype=bool)
            inrow[row] = True
            for k in range(1, RF):
                nr = list(row); nr[0], nr[k] = nr[k], nr[0]
                tot = self.row_weight(p, nr) - w0 - int(DS3[row[k], row[0]])
                if tot > 0:
                    tab[p, k - 1] = (tot, 0)

            def complete(cfg0, base, removed):
                ok = self._completions(base)
                if ok is None:
                    return
                ok = ok & ~inrow
                for b in base:
                    ok[b] = False
                if not ok.any():
                    return
                if len(removed) == 1:
                    cR = DF3[:B, removed[0]]
                else:
                    i = base[-1]
                    cR = np.minimum(DF3[i, removed[0]] + DF3[:B, removed[1]], DF3[i, removed[1]] + DF3[:B, removed[0]])
                wf_base = sum(int(self.WF[p, b]) for b in base)
                for li in range(RF):
                    if li < RF - 1:
                        ld = base[li]
                        gain = int(self.WL[p, ld]) + wf_base - int(self.WF[p, ld]) + self.WF[p] - w0
                        cL = int(DS3[ld, row[0]])
                    else:
                        gain = self.WL[p] + wf_base - w0
                        cL = DS3[:B, row[0]]
                    tot = np.where(ok, gain - cR - cL, -(1 << 40))
                    opt = np.zeros(B, dtype=np.int64)
                    if len(removed) == 1:
                        alt = None
                        if li == RF - 1 and removed[0] == row[0]:
                            alt = gain - DL3[:B, removed[0]]
                        elif li < RF - 1 and base[li] == row[0]:
                            alt = gain - DL3[:B, removed[0]] - DS3[removed[0], :B]
                        if alt is not None:
                            alt = np.where(ok, alt, -(1 << 40))
                            opt = (alt > tot).astype(np.int64)      # F closure preferred on ties
                            tot = np.maximum(tot, alt)
                    y = int(tot.argmax())          # lowest y among equal totals
                    if tot[y] > 0:
                        tab[p, cfg0 + li] = (int(tot[y]), y + 4096 * int(opt[y]))

            for rmi in range(RF):
                complete((RF - 1) + rmi * RF, [row[j] for j in range(RF) if j != rmi], [row[rmi]])
            for pi, (a, b) in enumerate(pairs):
                kept = [row[j] for j in range(RF) if j not in (a, b)]
                for ii in range(rfc):
                    i = int(self.cur[p, ii])
                    if i >= B or inrow[i]:
                        continue
                    complete((RF - 1) + RF * RF + (pi * rfc + ii) * RF, kept + [i], [row[a], row[b]])
        return tab

    def seed_row(self, p: int, cfg: int, y: int) -> List[int]:
        """The new row of configuration `cfg` of partition p completed by y (leader first)."""
        RF = self.RF
        rfc = self.cur.shape[1]
        row = [int(x) for x in self.A[p]]
        if cfg < RF - 1:
            nr = list(row); nr[0], nr[cfg + 1] = nr[cfg + 1], nr[0]
            return nr
        cfg -= RF - 1
        if cfg < RF * RF:
            rmi, li = divmod(cfg, RF)
            base = [row[j] for j in range(RF) if j != rmi]
        else:
            cfg -= RF * RF
            q, li = divmod(cfg, RF)
            pi, ii = divmod(q, rfc)
            a, b = list(itertools.combinations(range(RF), 2))[pi]
            base = [row[j] for j in range(RF) if j not in (a, b)] + [int(self.cur[p, ii])]
        full = base + [int(y)]
        ld = full[li]
        return [ld] + [b for b in full if b != ld]

    def seed_candidates(self, tab=None):
        tab = self.seed_table() if tab is None else tab
        ps, cs = np.nonzero(tab[:, :, 0] > 0)
        out = [(int(tab[p, c, 0]), int(p), int(c), int(tab[p, c, 1])) for p, c in zip(ps, cs)]
        out.sort(key=lambda c: (-c[0], c[1], c[2]))
        return out

    # ---- realisation ----
    def _path(self, mids, u: int, v: int, lev: int) -> List[int]:
        if u == v:
            return [u]
        if lev == 0:
            return [u, v]
        m = int(mids[lev][u, v])
        a = self._path(mids, u, m, lev - 1)
        b = self._path(mids, m, v, lev - 1)
        return a + b[1:]

    def _walk(self, X, used: set, layer: int, pth: List[int]) -> bool:
        E = (self.EF, self.ES, self.EL)[layer]
        RF = self.RF
        for s, d in zip(pth[:-1], pth[1:]):
            if s == d or s >= self.B or d >= self.B:      # slack nodes carry no slot
                continue
            key = int(E[s, d])
            if key == NO_EDGE:
                return False
            if layer == 2:
                pay = key & ((1 << 44) - 1)
                q, var, k, y = pay >> 20, (pay >> 16) & 15, (pay >> 12) & 15, pay & 4095
                if q in used:
                    return False
                used.add(q)
                row = [int(x) for x in self.A[q]]
                u = row[0]
                comp = None
                if var == 0:
                    X[q, 0] = d
                elif var == 1:
                    b = row[k]; X[q, 0] = d; X[q, k] = u; comp = (u, b)
                elif var == 2:
                    X[q, 0] = d; X[q, k] = y; comp = (y, d)
                elif var == 3:
                    X[q, 0], X[q, k] = row[k], u; comp = (u, d)
                else:
                    b = row[k]; X[q, 0] = d; X[q, k] = y; comp = (y, b)
                if comp is not None and comp[0] != comp[1]:
                    if not self._walk(X, used, 0, self._path(self.MF, comp[0], comp[1], LEVELS)):
                        return False
                continue
            q, j = divmod(key & 0xFFFFFFFF, RF)
            if q in used:
                return False
            used.add(q)
            if layer == 1:
                X[q, 0], X[q, j] = X[q, j], X[q, 0]
            else:
                X[q, j] = d
        return True

    def realise_cycle(self, cand):
        _, layer, lev, b = cand
        mids = (self.MF, self.MS, self.ML)[layer]
        m = int(mids[lev][b, b])
        pth = self._path(mids, b, m, lev - 1) + self._path(mids, m, b, lev - 1)[1:]
        X = self.A.copy()
        used: set = set()
        return [(X, used)] if self._walk(X, used, layer, pth) else []

    def realise_seed(self, cand):
        _, p, cfg, y = cand
        opt, y = divmod(y, 4096)
        row = [int(x) for x in self.A[p]]
        new = self.seed_row(p, cfg, y)
        Rm = [b for b in row if b not in new]
        Ad = [b for b in new if b not in row]
        if opt:
            X = self.A.copy()
            X[p] = new
            used = {p}
            good = self._walk(X, used, 2, self._path(self.ML, Ad[0], Rm[0], LEVELS))
            if good and new[0] == row[0]:
                good = self._walk(X, used, 1, self._path(self.MS, Rm[0], Ad[0], LEVELS))
            return [(X, used)] if good else []
        DF3 = self.DF[LEVELS]
        if len(Rm) == 2:
            m0 = DF3[Ad[0], Rm[0]] + DF3[Ad[1], Rm[1]]
            m1 = DF3[Ad[0], Rm[1]] + DF3[Ad[1], Rm[0]]
            orders = [(Rm[0], Rm[1]), (Rm[1], Rm[0])] if m0 <= m1 else [(Rm[1], Rm[0]), (Rm[0], Rm[1])]
        else:
            orders = [tuple(Rm)]
        out = []
        for order in orders:
            X = self.A.copy()
            X[p] = new
            used = {p}
            good = True
            for a, r in zip(Ad, order):
                good = good and self._walk(X, used, 0, self._path(self.MF, a, r, LEVELS))
            if good and new[0] != row[0]:
                good = self._walk(X, used, 1, self._path(self.MS, new[0], row[0], LEVELS))
            if good:
                out.append((X, used))
        return out

    def realisations(self, limit: int = MAX_EVAL, all_levels: bool = False):
        """Up to `limit` (assignment, used partitions) in candidate order; duplicates (same used set and rows) dropped."""
        cyc = self.cycle_candidates(all_levels)
        out, seen = [], set()
        cands = [("c", c) for c in cyc] if cyc else [("s", c) for c in self.seed_candidates()]
        for kind, c in cands:
            for X, used in (self.realise_cycle(c) if kind == "c" else self.realise_seed(c)):
                sig = tuple(sorted((q, tuple(int(v) for v in X[q])) for q in used))
                if sig in seen:
                    continue
                seen.add(sig)
                out.append((X, used))
                if len(out) >= limit:
                    return out
        return out


def evaluate(t, X) -> Tuple[int, int]:
    import kao_oracle as ko
    obj, viol = ko.verify(t, np.asarray(X).astype(np.uint16))
    return int(obj), int(np.asarray(viol).sum() if np.ndim(viol) else viol)


def round_step(t, A, evaluator=evaluate, bulk_slots: int = BULK_SLOTS):
    """One round.  Returns (new assignment or None, info).
    Bulk mode (round 4; topics of more than `bulk_slots` replica slots whose round has cycle candidates): up to 8 * MAX_EVAL
    candidates are unrolled, a partition-disjoint set is taken in candidate order BEFORE any scoring, and only its merges -- the
    first m, m/2, m/4, ..., 1 of them -- are scored; the best feasible improving merge wins (ties: the larger).  A round none of
    whose merges improves falls through to the one-by-one path on the first MAX_EVAL realisations."""
    rd = Round(t, A)
    base, v0 = evaluator(t, rd.A)
    assert v0 == 0, "KAO-CX starts from a feasible assignment"
    bulk = t.n_partitions * t.rf > bulk_slots and bool(rd.cycle_candidates())
    reals = rd.realisations(8 * MAX_EVAL if bulk else MAX_EVAL, all_levels=bulk)
    if bulk:
        chosen, taken = [], set()
        for X, used in reals:
            if used & taken:
                continue
            chosen.append((X, used))
            taken |= used
        win = None
        k = len(chosen)
        while k >= 1:
            merged = rd.A.copy()
            for X, used in chosen[:k]:
                for q in used:
                    merged[q] = X[q]
            o, v = evaluator(t, merged)
            if v == 0 and o > base and (win is None or o > win[0]):
                win = (o, k, merged)
            k //= 2
        if win is not None:
            return win[2].astype(np.uint16), dict(base=base, realisations=len(reals), improving=win[1], objective=win[0], merged=win[1], bulk=True)
        reals = reals[:MAX_EVAL]
    good = []
    for idx, (X, used) in enumerate(reals):
        o, v = evaluator(t, X)
        if v == 0 and o > base:
            good.append((o, idx, X, used))
    info = dict(base=base, realisations=len(reals), improving=len(good))
    if not good:
        return None, info
    best = max(good, key=lambda g: (g[0], -g[1]))
    chosen = []                                        # partition-disjoint improving realisations, candidate order
    taken: set = set()
    for g in good:
        if g[3] & taken:
            continue
        chosen.append(g)
        taken |= g[3]
    out, out_obj, n_taken = best[2], best[0], 1
    # merges of the first m, m/2, m/4, ... chosen ones (two compounds may still clash on a band's slack): the best feasible wins
    sizes = []
    k = len(chosen)
    while k >= 2:
        sizes.append(k)
        k //= 2
    for k in sizes:
        merged = rd.A.copy()
        for o, idx, X, used in chosen[:k]:
            for q in used:
                merged[q] = X[q]
        o, v = evaluator(t, merged)
        if v == 0 and o > out_obj:
            out, out_obj, n_taken = merged, o, k
    info.update(objective=out_obj, merged=n_taken)
    return out.astype(np.uint16), info


def improve(t, A, max_rounds: int = 64, evaluator=evaluate, bulk_slots: int = BULK_SLOTS):
    A = np.asarray(A).astype(np.uint16).reshape(t.n_partitions, t.rf)
    hist = []
    for _ in range(max_rounds):
        X, info = round_step(t, A, evaluator, bulk_slots)
        hist.append(info)
        if X is None:
            break
        A = X
    return A, hist
