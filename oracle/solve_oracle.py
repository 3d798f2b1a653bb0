"""solve_oracle.py -- checker for the solve stage.  TEST INFRASTRUCTURE ONLY.

The reference hands a MIQP/MILP to Gurobi 11.0.2 (src/ILP_index.cpp:757-771, 1412-1418), a
licensed third-party solver that is absent from /root/reference and from this image, and the
reference ships no expected output: PARITY UNPINNED against Gurobi itself.  What this module
provides instead, each independently of the product's DP:

  build_milp()      the reference's `-q0` program restated constraint by constraint
                    (k-mer rows :786-833, expanded graph + objective :1160-1315, flow rows
                    :1325-1401), solved by HiGHS through scipy.optimize.milp;
  brute_force()     exhaustive enumeration of the s->e paths of the expanded graph
                     with the exact objective of section 9.6;
  evaluate_path()   feasibility + objective of one decoded path on the same model.

README.md:87 of the reference states that the IQP (-q1) and ILP (-q0) programs solve the same
problem; SURVEY.md section 9.6 shows the two constraint sets have identical feasible sets.
"""
from collections import defaultdict

import numpy as np


class Model:
    """Kept anchors + graph in the shape the model builder of ILP_index.cpp:776-1409 sees them."""

    def __init__(self, graph, st, recombination=100):
        self.g = graph
        self.R = recombination
        self.c_half = recombination // 2                # c_1/2 with integer division (:1276, :1299)
        self.paths = graph.paths
        self.haps = [[] for _ in range(graph.n_vtx)]    # haps[v] (:109)
        self.idx = []                                   # elementIndexMaps (:1230-1239)
        for h, p in enumerate(self.paths):
            m = {}
            for i, v in enumerate(p):
                self.haps[v].append(h)
                m[v] = i
            self.idx.append(m)
        # anchors with >= 2 vertices per minimiser id (:795/:846 skip single-vertex ones)
        self.anchors = defaultdict(list)                # r -> [(h, t0, t1)]
        for r, h, t0, t1 in zip(st.a_r.tolist(), st.a_h.tolist(), st.a_t0.tolist(), st.a_t1.tolist()):
            if t1 > t0:
                self.anchors[r].append((h, t0, t1))
        self.minimizers = sorted(self.anchors)          # those that get a z_i (:822/:868)

    # ------------------------------------------------------------------ exact objective of a path
    def successors(self, v, h):
        """Feasible transitions out of state (v,h): [(v', h', switched)]."""
        p = self.paths[h]
        i = self.idx[h][v]
        if i == len(p) - 1:
            return []                                   # (last(h),h) must take its e variable
        out = [(p[i + 1], h, False)]
        for v2 in self.g.adj[v]:
            if v2 != p[i + 1]:
                for h2 in self.haps[v2]:
                    out.append((v2, h2, True))
        return out

    def objective(self, states):
        """states: [(v,h)] from a start state to an end state.  Returns (objective, n_covered, n_switch)
        or raises ValueError if the sequence is not a feasible s->e flow."""
        v0, h0 = states[0]
        if self.paths[h0][0] != v0:
            raise ValueError("path does not start at the first vertex of its walk")
        n_sw = 0
        for (v, h), (v2, h2) in zip(states, states[1:]):
            succ = self.successors(v, h)
            if (v2, h2, False) in succ:
                continue
            if (v2, h2, True) in succ:
                n_sw += 1
                continue
            raise ValueError(f"no transition ({v},{h}) -> ({v2},{h2})")
        vl, hl = states[-1]
        if self.paths[hl][-1] != vl:
            raise ValueError("path does not end at the last vertex of its walk")
        # hap-h edges used: consecutive states on the same walk joined by that walk's own edge
        used = set()
        for (v, h), (v2, h2) in zip(states, states[1:]):
            if h == h2:
                i = self.idx[h][v]
                if i + 1 < len(self.paths[h]) and self.paths[h][i + 1] == v2:
                    used.add((h, i))
        covered = 0
        for r in self.minimizers:
            if any(all((h, t) in used for t in range(t0, t1)) for h, t0, t1 in self.anchors[r]):
                covered += 1
        return covered - 2 * self.c_half * n_sw, covered, n_sw

    # ------------------------------------------------------------------ brute force
    def brute_force(self, limit=2_000_000):
        """max objective and every optimal state sequence."""
        best, arg, n = None, [], 0
        stack = [[(p[0], h)] for h, p in enumerate(self.paths)]
        while stack:
            st = stack.pop()
            v, h = st[-1]
            succ = self.successors(v, h)
            if not succ:
                n += 1
                if n > limit:
                    raise RuntimeError("brute force limit exceeded")
                val = self.objective(st)[0]
                if best is None or val > best:
                    best, arg = val, [st]
                elif val == best:
                    arg.append(st)
                continue
            for v2, h2, _ in succ:
                stack.append(st + [(v2, h2)])
        return best, arg

    # ------------------------------------------------------------------ the reference's -q0 MILP
    def build_milp(self):
        """Returns (c, A, lb, ub, integrality, var_names) for scipy.optimize.milp (minimise)."""
        from scipy.sparse import coo_matrix
        names, integ, cost = [], [], []
        var = {}

        def add(name, is_int, c=0.0):
            var[name] = len(names)
            names.append(name)
            integ.append(1 if is_int else 0)
            cost.append(c)
            return var[name]

        rows, cols, vals, lo, hi = [], [], [], [], []

        def constraint(terms, lb, ub):
            r = len(lo)
            for j, a in terms:
                rows.append(r); cols.append(j); vals.append(a)
            lo.append(lb); hi.append(ub)

        # k-mer rows (:786-833)
        for i in self.minimizers:
            zs = []
            for n, (j, t0, t1) in enumerate(self.anchors[i]):
                z = add(f"z_{i}_{j}_{n}", True)
                terms = []
                for t in range(t0, t1):
                    u, v = self.paths[j][t], self.paths[j][t + 1]
                    nm = f"{u}_{j}_{v}_{j}"
                    if nm not in var:
                        add(nm, False)                                      # -m1: continuous (:809)
                    terms.append((var[nm], 1.0))
                terms.append((z, -float(t1 - t0)))
                constraint(terms, 0.0, np.inf)                              # sum x >= weight * z (:817)
                zs.append(z)
            zi = add(f"z_{i}", True, -1.0)                                  # objective sum (1 - z_i) (:1310)
            constraint([(z, 1.0) for z in zs] + [(zi, -1.0)], 0.0, 0.0)     # :830
        const = float(len(self.minimizers))
        # start / end (:1167-1195)
        s_var, e_var = [], []
        for h, p in enumerate(self.paths):
            s_var.append(add(f"s_{p[0]}_{h}", False))
            e_var.append(add(f"{p[-1]}_{h}_e", False))
        constraint([(j, 1.0) for j in s_var], 1.0, 1.0)
        constraint([(j, 1.0) for j in e_var], 1.0, 1.0)
        new_adj = defaultdict(list)
        for h, p in enumerate(self.paths):                                  # :1204-1227
            for t in range(len(p) - 1):
                u, v = p[t], p[t + 1]
                nm = f"{u}_{h}_{v}_{h}"
                new_adj[f"{u}_{h}"].append(f"{v}_{h}")
                if nm not in var:
                    add(nm, False)
        for u in range(self.g.n_vtx):                                       # :1242-1304
            for v in self.g.adj[u]:
                w = f"w_{u}_{v}"
                used = False
                for h in self.haps[u]:
                    i = self.idx[h][u]
                    if i == len(self.paths[h]) - 1 or self.paths[h][i + 1] != v:
                        used = True
                        nm = f"{u}_{h}_{w}"
                        new_adj[f"{u}_{h}"].append(w)
                        if nm not in var:
                            add(nm, False)
                        cost[var[nm]] += self.c_half
                if used:
                    for h in self.haps[v]:
                        nm = f"{w}_{v}_{h}"
                        new_adj[w].append(f"{v}_{h}")
                        if nm not in var:
                            add(nm, False)
                        cost[var[nm]] += self.c_half
        in_new = defaultdict(list)
        for a, outs in new_adj.items():
            for b in outs:
                in_new[b].append(a)
        for h, p in enumerate(self.paths):                                  # :1326-1348
            for t in range(1, len(p) - 1):
                vtx = f"{p[t]}_{h}"
                terms = [(var[f"{a}_{vtx}"], 1.0) for a in in_new[vtx]] + [(var[f"{vtx}_{b}"], -1.0) for b in new_adj[vtx]]
                constraint(terms, 0.0, 0.0)
        for u in range(self.g.n_vtx):                                       # :1350-1372
            for v in self.g.adj[u]:
                w = f"w_{u}_{v}"
                if w in new_adj:
                    terms = [(var[f"{a}_{w}"], 1.0) for a in in_new[w]] + [(var[f"{w}_{b}"], -1.0) for b in new_adj[w]]
                    constraint(terms, 0.0, 0.0)
        for h, p in enumerate(self.paths):                                  # :1375-1401
            vtx = f"{p[0]}_{h}"
            constraint([(s_var[h], 1.0)] + [(var[f"{vtx}_{b}"], -1.0) for b in new_adj[vtx]], 0.0, 0.0)
        for h, p in enumerate(self.paths):
            vtx = f"{p[-1]}_{h}"
            constraint([(var[f"{a}_{vtx}"], 1.0) for a in in_new[vtx]] + [(e_var[h], -1.0)], 0.0, 0.0)
        A = coo_matrix((vals, (rows, cols)), shape=(len(lo), len(names))).tocsr()
        return (np.asarray(cost), A, np.asarray(lo), np.asarray(hi), np.asarray(integ), names, const)

    def milp_solve(self, time_limit=600.0):
        """Objective of the reference's program in this module's sign convention
        (max covered - 2*(R/2)*switches), via HiGHS."""
        from scipy.optimize import Bounds, LinearConstraint, milp
        c, A, lo, hi, integ, names, const = self.build_milp()
        res = milp(c, constraints=LinearConstraint(A, lo, hi), integrality=integ, bounds=Bounds(0.0, 1.0),
                   options={"time_limit": time_limit, "mip_rel_gap": 0.0})
        if res.status != 0:
            raise RuntimeError(f"HiGHS: {res.message}")
        minimised = res.fun + const                 # sum cost*x + sum (1 - z_i)
        return int(round(len(self.minimizers) - minimised)), res, names

    def model_size(self):
        c, A, lo, hi, integ, names, const = self.build_milp()
        return len(names), A.shape[0]


def states_from_path(path_vtx, path_hap):
    return list(zip([int(v) for v in path_vtx], [int(h) for h in path_hap]))
