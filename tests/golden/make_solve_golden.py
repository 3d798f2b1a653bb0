"""Golden objectives of the solve stage from an INDEPENDENT exact solver.

Gurobi (the reference's solver, ILP_index.cpp:757-771) is absent, so the reference's `-q0` program
is restated (oracle/solve_oracle.py) and solved by HiGHS (scipy.optimize.milp) on inputs small
enough for it.  Writes tests/golden/solve_golden.json.  Run in the build container:
    python tests/golden/make_solve_golden.py
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402
from oracle import solve_oracle as S  # noqa: E402
from phi_amd import synth  # noqa: E402


def syn_case(name, R, k=31, w=25, T=1.0):
    gk, rk = synth.CONFIGS[name]
    g = synth.make_graph(**gk)
    bases, off, truth = synth.make_reads(g, **rk)
    G = O.Graph(seg_names=[str(i) for i in range(g.n_vtx)],
                node_seq=[bytes(g.seq_concat[g.seq_off[v]:g.seq_off[v + 1]]) for v in range(g.n_vtx)],
                adj=[g.adj[g.adj_off[v]:g.adj_off[v + 1]].tolist() for v in range(g.n_vtx)],
                paths=[g.walk_vtx[g.walk_off[h]:g.walk_off[h + 1]].tolist() for h in range(g.n_walks)],
                hap_names=g.hap_names)
    O.kahn(G)
    reads = [bytes(bases[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
    st = O.run_stage12(G, reads, k, w, T)
    m = S.Model(G, st, R)
    t = time.time()
    val, res, names = m.milp_solve(time_limit=3000)
    return {"config": name, "k": k, "w": w, "T": T, "R": R, "objective": val, "n_in_model": int(st.n_in_model),
            "spectrum_size": int(len(st.spectrum)), "highs_seconds": round(time.time() - t, 1),
            "model_vars": len(names)}


def main():
    out = []
    for R in (100, 10, 2):
        out.append(syn_case("tiny", R))
        print(out[-1], flush=True)
    for R in (100, 6):
        out.append(syn_case("small", R))
        print(out[-1], flush=True)
    json.dump(out, open(os.path.join(HERE, "solve_golden.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
