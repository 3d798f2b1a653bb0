"""bench.py's own N > 1 branch under the driver's launcher, on ONE GPU: two ranks started by
torch.distributed.run, both on cuda:0 -- the same sharding, step, barrier / max-over-ranks timing and solve agreement
code as the multi-GPU run.  The exchange: --rehearse-ipc = the library's mapped hit vectors (phi_ipc_*: what a multi-GPU
run of an MHC-sized graph takes by default, here with both processes on one device), --rehearse-gloo = staged through
gloo on the host (the library's RCCL communicator is covered for one rank by test_rccl_communicator_inside_the_library)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("scaling,exchange,n_ranks", [("weak", "gloo", 2), ("strong", "gloo", 2), ("weak", "ipc", 2), ("strong", "ipc", 2), ("weak", "ipc", 4)])
def test_bench_two_ranks_rehearsed_on_one_gpu(scaling, exchange, n_ranks):
    import __graft_entry__
    __graft_entry__.ensure_built()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n_ranks), "--steps", "3", "--warmup", "1",
           "--rehearse-" + exchange, "--config", "small", "--strong-config", "small", "--scaling", scaling, "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == n_ranks and d["steps"] == 3 and d["scaling"] == scaling and d["value"] > 0
    assert d["result"]["optimal"] == 1
    if exchange == "ipc":
        assert "mapped hit vectors" in d["config"]["exchange"] and d["step_split"]["ranks"] == n_ranks
    other = "strong_scaling" if scaling == "weak" else "weak_scaling"
    assert d[other]["value"] > 0
    if scaling == "strong":
        assert f"{n_ranks} shard(s)" in d["config"]["workload"]
    # with the reads of both ranks (weak: two read sets of one sample; strong: one set in two shards) the truth walks come back
    assert d["result"]["path_walks"] == d["result"]["truth_walks"]
