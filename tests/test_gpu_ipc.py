"""The exchange between PROCESSES through mapped hit vectors (include/phi_amd.h phi_ipc_*, phi_amd/csrc/phi_ipc.hip), with
the processes on ONE GPU (IPC handles work on the same device): every line of the protocol but the xGMI hop itself --
shared-memory block and host barriers, handles published and mapped, four rotating hit vectors, the flag handshake inside
the gather kernel, the gather on its own stream beside the next read set's scoring, observers that wait for it, the lists
of novel read hashes through mapped buffers, teardown; and a rank that does not come (the others give up after the
timeout instead of hanging the GPU)."""
import multiprocessing as mp
import os

import numpy as np
import pytest

import ipc_worker

pytestmark = pytest.mark.gpu

K, W, STEPS = 15, 10, 7


def _spawn(n_ranks, seed, skip_step=-1, env=None):
    import phi_amd
    uid = phi_amd.Context.ipc_unique_id()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    old = {}
    for k_, v_ in (env or {}).items():
        old[k_] = os.environ.get(k_)
        os.environ[k_] = v_
    try:
        procs = [ctxm.Process(target=ipc_worker.run_rank, args=(r, n_ranks, uid, seed, STEPS, K, W, q, skip_step)) for r in range(n_ranks)]
        for p in procs:
            p.start()
        outs = [q.get(timeout=240) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0, p.exitcode
    finally:
        for k_, v_ in old.items():
            if v_ is None:
                del os.environ[k_]
            else:
                os.environ[k_] = v_
    for o in outs:
        assert o.get("error") is None, o["error"]
    return sorted(outs, key=lambda o: o["rank"])


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_processes_on_one_gpu_exchange_through_mapped_hit_vectors(oracle, ctx_factory, n_ranks):
    seed = 4100 + n_ranks
    outs = _spawn(n_ranks, seed)
    g, sets = ipc_worker.make_case(seed, STEPS)
    A = g.arrays()
    ref = ctx_factory(k=K, w=W, threshold=1.0, recombination=6)
    ref.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
    walk_h = np.concatenate([oracle.sketch(b"".join(g.node_seq[v] for v in p), K, W)[0] for p in g.paths])
    _, first = np.unique(walk_h, return_index=True)
    uniq = walk_h[np.sort(first)]
    # the hit vectors looked at along the way = those of ALL the step's reads, on every rank
    for s in (0, 3, STEPS - 2):
        rh = np.unique(np.concatenate([oracle.sketch(r, K, W)[0] for r in sets[s]]))
        want = np.isin(uniq, rh).astype(np.uint8)
        for o in outs:
            assert np.array_equal(o["hits"][s], want), (s, o["rank"])
    # the job: every rank solves as one context with all the reads does
    ref.add_reads(sets[STEPS - 1])
    want_res = ref.solve()
    want_distinct = ref.reads_stats()["n_distinct"]
    for o in outs:
        assert o["stats"] == want_distinct, o["rank"]
        for key, v in want_res.items():
            got = o["res"][key]
            assert got == (v.tolist() if hasattr(v, "tolist") else v), (o["rank"], key)
    ref.reset_reads()
    ref.add_reads(sets[0])
    alone = ref.solve()["spectrum_size"]
    assert all(o["alone"] == alone for o in outs)


def test_a_rank_that_does_not_come_is_an_error_not_a_hang():
    outs = _spawn(2, 4200, skip_step=2, env={"PHI_IPC_TIMEOUT_S": "0.5"})
    # rank 1 left out the exchange of read set 2: rank 0's gather of that set gave up after the timeout (rank 1's own
    # gathers of the later sets may or may not have found rank 0's flags in time: what matters is that both came back)
    assert outs[0]["check"].startswith("error"), outs[0]["check"]
