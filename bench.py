#!/usr/bin/env python3
"""bench.py -- headline benchmark of the PHI hot path on MI355X.

Metric (BASELINE.json): read Gbases/s scored on the 49-haplotype MHC graph, plus end-to-end
seconds.  The real graph cannot be built offline, so the workload is SURVEY.md 8(d)'s
deterministic stand-in "synMHC-49" (config C2: 49 walks x ~5.15 Mbp, nodes <= 30 bp, 1x 150-bp
reads).  One STEP = one scoring pass of a rank's read set, reads already resident in HBM:
    reset spectrum/hits -> 2-bit pack -> (w,k)-minimiser sketch + murmur3 -> spectrum insert +
    probe of the walk-minimiser table   [+ RCCL all-reduce(MAX) of the hit vector when N > 1]
The graph index (walk sketch + table) is built once before the timed region and the exact solve
(filter + DP + certificate) runs once after it; both are timed and reported separately, and
`gpu_path_s` = index build + one step + solve (inputs in memory); `end_to_end_s` = the command line from files,
process start to the closed FASTA.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU; every rank holds the full graph index and a shard of reads.  The path
has ONE exchange per read set: the hit vector (one byte per distinct walk
minimiser) is all-reduced (MAX) by the library's own RCCL communicator (phi_comm_*, include/phi_amd.h)
on the stream the kernels run on -- once per step, inside the timed region, since a step is a whole
read set.  The once-per-job merge of the read hashes that are not walk minimisers (it only feeds log
counters) runs with the solve.  Two scalings:
    --scaling weak    (default; `value`)  every rank scores its own C2-sized read set per step;
    --scaling strong  ONE read set (--strong-config, default C3: 10x reads) is cut into N contiguous
                      shards balanced by bases (phi_amd.dist.shard_bounds); a step is shard score +
                      all-reduce, value = the set's bases / step time.
Whichever is chosen for `value`, the other is measured in the same run with fewer steps and reported
under "strong_scaling" / "weak_scaling".
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def _host_threads():
    """Threads for the CPU baseline: the cores this process may use, at most the box's CPU share of one
    GPU (16) unless PHI_BASELINE_THREADS says otherwise."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("PHI_BASELINE_THREADS", "16"))))


def cpu_baseline(g_arrays, reads_concat, read_off, k, w, budget_s=6.0):
    """The oracle (scalar string-based port of compute_hashes / index_kmers / compute_anchors / the filter,
    with the reference's OpenMP loops) on the host cores: checker code timed as the CPU baseline on a bounded
    sample, never part of the measured GPU path.  Reads sketch at -t1 and -t<threads>, then stages 1-2 of
    the whole configuration at -t<threads> (the reference parallelises the walk sketch over walks and the
    read sketch over reads, ILP_index.cpp:559,617; match and filter are timed as one-thread restatements)."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])    # the checker: built on demand, as tests/conftest.py does
    from oracle import oracle as O
    L = O.lib()
    raw = reads_concat.tobytes()
    off = np.ascontiguousarray(read_off, np.int64)
    n_reads, n_bases = len(off) - 1, int(off[-1])
    nt = _host_threads()

    def leg(threads, budget):
        L.orc_set_threads(threads)
        t0 = time.perf_counter()
        done = 0
        while True:
            L.orc_sketch_reads(raw, off.ctypes.data, n_reads, k, w)
            done += n_bases
            if time.perf_counter() - t0 > budget:
                break
        dt = time.perf_counter() - t0
        return done / dt / 1e9, done, dt
    v1, d1, t1 = leg(1, budget_s)
    vn, dn, tn = leg(nt, budget_s)
    out = {"value": vn, "unit": "Gbases/s", "cores": nt, "kind": "port",
           "sample": f"read sketch (oracle/phi_oracle.c orc_sketch_reads, OpenMP over reads as ILP_index.cpp:617): "
                     f"{dn / 1e6:.0f} Mbases of the same 150-bp synthetic reads in {tn:.1f} s on {nt} threads; "
                     f"{d1 / 1e6:.0f} Mbases in {t1:.1f} s on 1 thread.  The port hashes only when the window's minimum changes and allocates nothing: "
                     f"it is ~20x faster per core than the reference's own loop, measured at 0.0025 Gbases/s/core (BASELINE.md section 3)",
           "value_1core": v1}
    # stages 1-2 of the whole configuration (walk sketch, read sketch + spectrum, anchors, filter)
    L.orc_set_threads(nt)
    A = g_arrays
    if len(A["walk_vtx"]) > 50_000_000:
        out["stages"] = {"note": "skipped: this configuration's walks take the scalar port minutes (see the C2 line for the per-stage times)"}
        return out
    t0 = time.perf_counter()
    stg = O.run_stage12_arrays(A, reads_concat, off, k, w, 1.0, threads=nt, want_minimizers=False)
    t_all = time.perf_counter() - t0
    st = stg.stage_s
    out["_stage12"] = stg                                   # main() compares its counters with the GPU's (parity_checked)
    walk_bases = int((A["seq_off"][A["walk_vtx"] + 1] - A["seq_off"][A["walk_vtx"]]).sum())
    out["stages"] = {"threads": nt, "walk_sketch_s": st[0], "read_sketch_spectrum_s": st[1], "anchors_s": st[2],
                     "filter_s": st[3], "total_s": t_all, "walk_gbases_per_s": walk_bases / max(st[0], 1e-9) / 1e9,
                     "note": "whole configuration, oracle orc_run; the Gurobi solve of the reference cannot run here (BASELINE.md quotes its published times)"}
    return out


def _phi_stage_table(stderr):
    """The stage table PHI prints under PHI_TIMING=1: {stage: [begin_s, end_s]} on the process's own clock, plus the
    epochs of main() and of the closed FASTA."""
    import re
    stages, info = {}, {}
    for l in stderr.splitlines():
        m = re.match(r"\[phi timing\] main: stage (.+?)\s+(-?[\d.]+) ->\s+(-?[\d.]+) s", l)
        if m:
            stages[m.group(1).strip()] = [float(m.group(2)), float(m.group(3))]
        m = re.search(r"main: entered at epoch ([\d.]+)", l)
        if m:
            info["main_entered_epoch"] = float(m.group(1))
        m = re.search(r"FASTA closed at epoch ([\d.]+)", l)
        if m:
            info["fasta_closed_epoch"] = float(m.group(1))
        m = re.match(r"\[phi timing\] (ctx_create|gfa_read): (.+?)\s+([\d.]+) ms", l)
        if m:
            stages.setdefault("detail", {})[f"{m.group(1)}: {m.group(2).strip()}"] = float(m.group(3))
    return stages, info


def file_to_fasta(g, bases, off, k, w, runs=5, fastq=False):
    """The drop-in command line on this configuration's files (GFA 1.1 + FASTA / FASTQ in a temp dir, page cache warm):
    `PHI -g -r -o` from process start to the closed FASTA, parse and H2D included.  Three ways,
    none of whose definitions relies on a pause:
      back to back   `runs` consecutive commands, nothing between them, each ONE process (the default: when the command
                     returns its memory and its GPU state are given back, as the reference's does): the median command =
                     end_to_end_s, the wall clock of all of them / runs = back_to_back_s (what a harness that loops PHI over
                     samples sees, data/run_batch_4_miqp.py:31-46);
      detached       PHI_DETACH=1: the command returns when the FASTA is closed, its teardown runs on behind it (round 3's
                     end_to_end_s; here with 0.4 s between two runs, because the NEXT start would otherwise wait for that teardown);
      several jobs   ONE command with `runs` read sets against the graph (-r .. -o .. -r .. -o ..): graph parsed and indexed once."""
    from phi_amd import synth
    phi = os.path.join(ROOT, "phi_amd", "PHI")
    if not os.path.exists(phi):
        return None
    with tempfile.TemporaryDirectory(prefix="phi_bench_") as d:
        gfa, rd, fa = os.path.join(d, "g.gfa"), os.path.join(d, "r.fq" if fastq else "r.fa"), os.path.join(d, "out.fa")
        synth.write_gfa(g, gfa)
        synth.write_reads(bases, off, rd, fastq=fastq)
        env = dict(os.environ, PHI_TIMING="1")
        env.pop("PHI_DETACH", None)

        def one(e, extra=()):
            t_spawn = time.time()
            t0 = time.perf_counter()
            r = subprocess.run([phi, "-g", gfa, "-r", rd, "-o", fa, "-k", str(k), "-w", str(w)] + list(extra), capture_output=True, text=True, env=e)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": r.stderr[-300:]}
            stages, info = _phi_stage_table(r.stderr)
            return {"wall_s": dt, "spawn_to_main_s": info.get("main_entered_epoch", t_spawn) - t_spawn,
                    "spawn_to_fasta_closed_s": info.get("fasta_closed_epoch", t_spawn + dt) - t_spawn, "stages": stages}
        first = one(env)                                        # (the page cache, and a first run right behind the bench's own allocations)
        if "error" in first:
            return first
        t0 = time.perf_counter()
        recs = [one(env) for _ in range(runs)]
        back_to_back = (time.perf_counter() - t0) / runs
        if any("error" in x for x in recs):
            return [x for x in recs if "error" in x][0]
        det = []
        for _ in range(runs):
            time.sleep(0.4)
            det.append(one(dict(env, PHI_DETACH="1")))
        extra = []
        for i in range(1, runs):
            extra += ["-r", rd, "-o", os.path.join(d, f"out{i}.fa")]
        t0 = time.perf_counter()
        multi = one(env, extra)
        multi_wall = time.perf_counter() - t0
        med = sorted(recs, key=lambda x: x["wall_s"])[len(recs) // 2]
        det_ok = [x["wall_s"] for x in det if "error" not in x]
        return {"seconds": med["wall_s"], "seconds_all_runs": [x["wall_s"] for x in recs], "seconds_min": min(x["wall_s"] for x in recs), "seconds_first_run": first["wall_s"],
                "back_to_back_s": back_to_back,
                "detached_seconds": float(np.median(det_ok)) if det_ok else None, "detached_all_runs": det_ok,
                "several_jobs_per_read_set_s": None if "error" in multi else multi_wall / runs, "several_jobs_command_s": None if "error" in multi else multi_wall,
                "spawn_to_fasta_closed_s": med["spawn_to_fasta_closed_s"], "spawn_to_main_s": med["spawn_to_main_s"],
                "stages_s": {k_: v for k_, v in med["stages"].items() if k_ != "detail"}, "detail_ms": med["stages"].get("detail", {}),
                "gfa_mb": os.path.getsize(gfa) / 1e6, "reads_mb": os.path.getsize(rd) / 1e6,
                "note": f"phi_amd/PHI on uncompressed files (page cache warm).  seconds = the median of {runs} consecutive commands with nothing between them, each one process "
                        "(teardown inside the wall clock); back_to_back_s = the wall clock of all of them / their number; detached_seconds = the same command with PHI_DETACH=1 "
                        "(it returns when the FASTA is closed, the teardown runs on behind it; 0.4 s between two runs); several_jobs_per_read_set_s = ONE command with "
                        f"{runs} read sets against the graph, its wall clock / {runs}; stages_s = [begin, end] on the process's clock (median run)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2", help="workload of phi_amd.synth.CONFIGS, or C1syn (reference MHC_4 graph + generator reads)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--strong-config", default="C3", help="read set that --scaling strong shards (same graph as --config)")
    ap.add_argument("--prof-launches", type=int, default=48,
                    help="sketch launches bracketed by HIP events for roofline.kernel_avg_ms: a leg of its own right after the timed "
                         "steps, same step, every launch bracketed (a bracketed launch costs the stream ~5 us more than a plain one, "
                         "so the timed steps themselves carry none)")
    ap.add_argument("--job-repeats", type=int, default=5,
                    help="fresh context -> index build -> reads -> solve, this many times after the timed steps: *_cold = this process's "
                         "first, the plain names = medians over the repeats")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solve", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the second scaling mode, the H2D-inclusive rate and the command-line run")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="N > 1 on ONE GPU for testing: every rank uses cuda:0 and the exchange goes through gloo on the host")
    ap.add_argument("--rehearse-ipc", action="store_true",
                    help="N > 1 on ONE GPU for testing: every rank uses cuda:0, the exchange is the library's mapped hit vectors (phi_ipc_*), "
                         "torch.distributed (gloo) only carries the group's id and the timing reductions")
    ap.add_argument("--exchange", choices=("auto", "ipc", "rccl"), default="auto",
                    help="N > 1: the per-read-set exchange of the hit vectors.  ipc: every rank maps the others' vectors and ORs them with one kernel on a "
                         "stream of its own, beside the next read set's scoring (phi_ipc_*); rccl: ncclAllReduce inside the library (phi_comm_*); "
                         "auto: ipc for hit vectors of up to 4 M flags (every MHC-sized graph), rccl beyond, and rccl when the ipc group cannot be set up or fails its check")
    args = ap.parse_args()

    import torch
    import __graft_entry__
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        __graft_entry__.ensure_built()                         # a checkout without built libraries (no fallback exists)
    else:
        t_wait = time.time()
        libs = [os.path.join(ROOT, "phi_amd", n) for n in ("libphi_amd.so", "libphi_host.so")]
        while not all(os.path.exists(p) for p in libs) and time.time() - t_wait < 600:
            time.sleep(1.0)                                    # rank 0 is building
    import phi_amd
    from phi_amd import dist as pdist
    from phi_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch through torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        rehearsal = args.rehearse_gloo or args.rehearse_ipc
        if rehearsal:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctl_dev = "cpu" if (world > 1 and (args.rehearse_gloo or args.rehearse_ipc)) else dev      # where control-plane tensors of torch.distributed live

    K, W = 31, 25
    t0 = time.perf_counter()
    native = args.config in synth.NATIVE_CONFIGS
    if native:
        # chromosome-scale configurations come from the native generator (libphi_synth.so): counter-based, so every
        # rank makes its own reads -- its own set (weak) or its shard of the common set (strong) -- without the others'
        gk, s_seed, n_mosaic, r_seed, cov = synth.NATIVE_CONFIGS[args.config]
        g = synth.NativeGraph(**gk)
        truth = g.sample(s_seed, n_mosaic)
        rk = dict(seed=r_seed)
        bases, off = g.reads(r_seed + 1000 * rank, 0, g.n_reads(cov))
    elif args.config == "C1syn":
        # the reference's own graph (test/MHC_4.gfa.gz, 5 walks) with the generator's reads
        gk = dict(seed="tests/golden/data/MHC_4.gfa.gz")
        g = synth.graph_from_gfa(os.path.join(ROOT, "tests", "golden", "data", "MHC_4.gfa.gz"))
        rk = dict(coverage=1.0, seed=4102, n_mosaic=2)
        strong_rk = dict(coverage=10.0, seed=4103, n_mosaic=2)
    else:
        gk, rk = synth.CONFIGS[args.config]
        g = synth.make_graph(**gk)
        strong_rk = dict(synth.CONFIGS[args.strong_config][1]) if args.strong_config in synth.CONFIGS else dict(rk)
    if not native:
        rk = dict(rk)
        if rank:
            rk["sample_seed"] = rk["seed"] + 1000 * rank       # weak scaling: each rank scores its own reads of the same sample
        bases, off, truth = synth.make_reads(g, **rk)
    t_gen = time.perf_counter() - t0
    n_reads, n_bases = len(off) - 1, int(off[-1])

    # one explicit stream for everything: the kernels behind the C ABI, torch's copies and the RCCL
    # collective are ordered on it (torch's default stream is the null stream, which the library
    # would replace by a private non-blocking one -- unordered against torch)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = phi_amd.Context(local_rank)
    ctx.set_params(k=K, w=W, threshold=1.0, recombination=100)
    ctx.set_stream(stream.cuda_stream)
    A = g.arrays()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
    torch.cuda.synchronize()
    t_index = time.perf_counter() - t0
    index_info = ctx.index_stats()
    walk_bases = index_info["walk_bases"]
    # graph-side de-duplication: walk bases indexed per second of GPU time of the class sketch
    index_info["sketch_gbases_per_s"] = walk_bases / max(index_info["sketch_gpu_ms"], 1e-6) / 1e6
    index_info["dedup_factor_bases"] = walk_bases / max(1, index_info["class_bases"])

    # The exchange of a read set's hit vectors, inside the library; torch.distributed only carries the 128 bytes of the group's id.
    #   ipc : every rank maps the others' hit vectors (hipIpc*) and ORs them with ONE kernel on a stream of its own -- ordered by
    #         flags in device memory, no host call per exchange, beside the next read set's scoring (phi_amd/csrc/phi_ipc.hip);
    #   rccl: ncclAllReduce(MAX, uint8) on the context's stream (phi_comm.hip): 25-40 us of latency for 0.5 MB, more than C2's step.
    # No multi-GPU node has been available to any round: the ipc group is CHECKED here before it is used (one exchange of known
    # vectors, the ranks compare what came out) and the run falls back to RCCL if it cannot be set up or the check fails.
    use_ipc = False
    exchange_note = None
    if world > 1 and not args.rehearse_gloo:
        want_ipc = args.rehearse_ipc or args.exchange == "ipc" or (args.exchange == "auto" and index_info["n_distinct_minimizers"] <= (4 << 20))
        if want_ipc:
            ok = 1
            try:
                box = [phi_amd.Context.ipc_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                ctx.ipc_init(box[0], rank, world)
                # the check: rank r scores a read set of its own (the generator's seed + r), the exchanged vector must hold at least the
                # rank's own flags and the SAME number of flags on every rank
                cb, co, _ = (g.reads(7000 + rank, 0, 2000) + (None,)) if native else synth.make_reads(g, **dict(rk, coverage=0.02, seed=7000 + rank, sample_seed=7000 + rank))
                d_cb, d_co = torch.from_numpy(np.ascontiguousarray(cb)).to(dev), torch.from_numpy(np.ascontiguousarray(co)).to(dev)
                ctx.reset_reads()
                ctx.add_reads_device(d_cb.data_ptr(), d_co.data_ptr(), len(co) - 1, int(co[-1]))
                p_, n_ = ctx.hits_buffer()
                stream.synchronize()
                own = int(torch.as_tensor(pdist.DevArray(p_, n_), device=dev).sum().item())
                ctx.ipc_allreduce_hits()
                ctx.ipc_check()
                p_, n_ = ctx.hits_buffer()
                stream.synchronize()
                got = int(torch.as_tensor(pdist.DevArray(p_, n_), device=dev).sum().item())
                t = torch.tensor([got, -got, own], dtype=torch.int64, device=ctl_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                if int(t[0]) != -int(t[1]) or got < own or (world > 1 and got < int(t[2])):
                    ok = 0
                    exchange_note = f"ipc check failed: {got} flags here, max {int(t[0])} / min {-int(t[1])} over ranks, own {own}"
                ctx.reset_reads()
            except Exception as e:                                      # noqa: BLE001 -- whatever went wrong, the run goes on over RCCL
                ok = 0
                exchange_note = f"ipc group not usable: {e}"
            t = torch.tensor([ok], dtype=torch.int64, device=ctl_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            use_ipc = bool(int(t[0]))
            if not use_ipc:
                try:
                    ctx.ipc_destroy()
                except Exception:                                       # noqa: BLE001
                    pass
                if args.rehearse_ipc or args.exchange == "ipc":
                    raise SystemExit(f"bench.py: the ipc exchange was asked for and is not usable ({exchange_note})")
    use_lib_comm = world > 1 and not args.rehearse_gloo and not use_ipc
    if use_lib_comm:
        box = [phi_amd.Context.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctx.comm_init(box[0], rank, world)

    def dev_sync():
        # (a group of processes: the gather of the LAST exchange starts with the next read launch -- or with this flush)
        if use_ipc:
            ctx.ipc_flush()
        torch.cuda.synchronize()

    def allreduce_hits():
        # the one data-path collective of a read set: OR (= MAX) of the hit vectors, in place
        if use_ipc:
            ctx.ipc_allreduce_hits()
        elif use_lib_comm:
            ctx.comm_allreduce_hits()
        elif world > 1:
            hit_ptr, n_unique = ctx.hits_buffer()
            stream.synchronize()
            pdist.allreduce_hits(torch.as_tensor(pdist.DevArray(hit_ptr, n_unique), device=dev))

    def make_step(d_b, d_o, nr, nb, with_offsets=False):
        # reads of one length (the 150-bp sets of C2 / C3 / C5) are handed over WITHOUT offsets, as the command line's device-side
        # record splitter hands them over: the kernel computes the read starts (include/phi_amd.h, phi_add_reads_device)
        off_ptr = d_o.data_ptr() if (with_offsets or not _uniform(d_o, nr, nb)) else None

        def step():
            ctx.reset_reads()
            ctx.add_reads_device(d_b.data_ptr(), off_ptr, nr, nb)
            allreduce_hits()
        step.uses_offsets = off_ptr is not None
        return step

    def _uniform(d_o, nr, nb):
        if nr == 0 or nb % nr or nb // nr < 32:
            return False
        L = nb // nr
        return bool((d_o[1:] - d_o[:-1] == L).all().item())

    def timed(step, steps, warmup):
        for _ in range(warmup):
            step()
        if world > 1:
            dist.barrier()
        dev_sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        dev_sync()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=ctl_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # ---- read sets on the device: this rank's own set (weak) and its shard of the common set (strong)
    d_bases = torch.from_numpy(bases).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    weak_step = make_step(d_bases, d_off, n_reads, n_bases)
    need_strong = args.scaling == "strong" or not args.no_extra_legs
    strong = None
    if need_strong and native:
        # the common set is the configuration's own read set; this rank makes only its shard of it
        tot = g.n_reads(cov)
        lo, hi = tot * rank // world, tot * (rank + 1) // world
        sh_bases, sh_off = g.reads(r_seed, lo, hi)
        strong = dict(total_bases=tot * 150, total_reads=tot, n_reads=hi - lo, n_bases=int(sh_off[-1]),
                      d_b=torch.from_numpy(sh_bases).to(dev), d_o=torch.from_numpy(sh_off).to(dev))
        args.strong_config = args.config
    elif need_strong:
        sb, so, _ = synth.make_reads(g, **strong_rk)            # the same set on every rank
        lo, hi = pdist.shard_bounds(so, world, rank)
        sh_off = (so[lo:hi + 1] - so[lo]).astype(np.int64)
        sh_bases = sb[int(so[lo]):int(so[hi])]
        strong = dict(total_bases=int(so[-1]), total_reads=len(so) - 1, n_reads=hi - lo, n_bases=int(sh_off[-1]),
                      d_b=torch.from_numpy(np.ascontiguousarray(sh_bases)).to(dev), d_o=torch.from_numpy(sh_off).to(dev))
    if need_strong:
        strong["step"] = make_step(strong["d_b"], strong["d_o"], strong["n_reads"], strong["n_bases"])

    # clock ramp: a fresh box runs its first launches at idle clocks (a third slower for the first tens
    # of milliseconds); untimed, before the W warmup steps (no collective inside: ranks need not agree on its length)
    primary_step = weak_step if args.scaling == "weak" else strong["step"]
    ramp_off = d_off.data_ptr() if weak_step.uses_offsets else None
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.25:
        for _ in range(50 if n_bases < 5e7 else 1):
            ctx.reset_reads()
            ctx.add_reads_device(d_bases.data_ptr(), ramp_off, n_reads, n_bases)
        dev_sync()

    for _ in range(args.warmup):
        primary_step()
    elapsed = timed(primary_step, args.steps, 0)
    # the kernel's own duration: a leg of its own, the same step with EVERY sketch launch bracketed by HIP events on the
    # stream the kernel runs on (the timed steps above carry no events: a bracketed launch costs the stream ~5 us more)
    ctx.prof_read()                                            # drop earlier timings
    ctx.prof_enable(1)
    n_prof = max(24, args.prof_launches) if n_bases < 5e8 else max(3, min(args.prof_launches, args.steps))
    for _ in range(n_prof):
        primary_step()
    dev_sync()
    ctx.prof_enable(False)
    n_launch, kern_ms, kern_bases = ctx.prof_read()
    # several GPUs: where a step's time goes -- the rank's own scoring, and the exchange (which also absorbs the wait for the slowest rank)
    step_split = None
    if world > 1:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        sc, ar = [], []
        d_b, d_o, nr, nb = (d_bases, d_off, n_reads, n_bases) if args.scaling == "weak" else (strong["d_b"], strong["d_o"], strong["n_reads"], strong["n_bases"])
        for _ in range(20):
            if use_ipc:
                ctx.ipc_flush()
            ctx.reset_reads()
            ev[0].record(stream)
            ctx.add_reads_device(d_b.data_ptr(), d_o.data_ptr(), nr, nb)
            ev[1].record(stream)
            allreduce_hits()
            ev[2].record(stream)
            stream.synchronize()
            sc.append(ev[0].elapsed_time(ev[1])); ar.append(ev[1].elapsed_time(ev[2]))
        t = torch.tensor([float(np.median(sc)), float(np.median(ar))], dtype=torch.float64, device=ctl_dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        step_split = {"score_ms_median_this_rank": float(t[0]), "allreduce_ms_median_this_rank": float(t[1]),
                      "score_ms_max_over_ranks": float(tmax[0]), "allreduce_ms_max_over_ranks": float(tmax[1]),
                      "hit_vector_bytes": int(index_info["n_distinct_minimizers"]), "ranks": ctx.ipc_info()[1] if use_ipc else (ctx.comm_info()[1] if use_lib_comm else world),
                      "step_ms": elapsed / args.steps * 1e3,
                      "exchange_exposed_ms": max(0.0, elapsed / args.steps * 1e3 - float(tmax[0])),
                      "note": ("GPU time between events on the context's stream, 20 steps: reset + sketch launch = score; the exchange (ipc) runs on a stream of its own beside "
                               "the NEXT read set's scoring, so allreduce_ms here is only what the context's stream sees of it (the issue of the gather: ~0); "
                               "exchange_exposed_ms = timed step - score = what the exchange adds to a step (waiting for the slowest rank included)") if use_ipc else
                              ("GPU time between events on the stream, 20 steps: reset + sketch launch, then ncclAllReduce(MAX, uint8) of the hit vector; "
                               "the all-reduce figure includes waiting for the slowest rank's sketch; exchange_exposed_ms = timed step - score")}
    # the same read set handed over WITH its offsets array (reads of uneven lengths always are): the general path, beside `value`
    value_with_offsets = None
    if args.scaling == "weak" and not primary_step.uses_offsets and not args.no_extra_legs:
        gstep = make_step(d_bases, d_off, n_reads, n_bases, with_offsets=True)
        g_steps = max(10, args.steps)
        el_g = timed(gstep, g_steps, max(2, args.warmup))
        value_with_offsets = n_bases * world * g_steps / el_g / 1e9
        primary_step()                                         # leave the context holding the primary read set
        dev_sync()
    # |Sp_R| is made from the log of novel read hashes when it is first asked for (once per read set, not once per step):
    # phi_reads_stats here, phi_solve in a job -- timed as its own leg, and inside solve_s / gpu_path_s of the job below
    dev_sync()
    t0 = time.perf_counter()
    stats = ctx.reads_stats()
    spectrum_dedupe_ms = (time.perf_counter() - t0) * 1e3
    density = stats["n_emitted"] / max(1, stats["n_bases"])
    ms_per_step = elapsed / args.steps * 1e3
    step_bases = n_bases * world if args.scaling == "weak" else strong["total_bases"]
    value = step_bases * args.steps / elapsed / 1e9

    # ---- the other scaling mode, in the same run, with fewer steps
    other = None
    if not args.no_extra_legs:
        o_steps, o_warm = max(5, args.steps // 4), max(2, args.warmup // 4)
        if args.scaling == "weak":
            el = timed(strong["step"], o_steps, o_warm)
            other = ("strong_scaling", {"value": strong["total_bases"] * o_steps / el / 1e9, "unit": "Gbases/s", "ms_per_step": el / o_steps * 1e3, "steps": o_steps,
                                        "workload": f"{args.strong_config} reads: ONE set of {strong['total_reads']} reads, {strong['total_bases'] / 1e6:.2f} Mbases, cut into {world} shard(s) balanced by bases; "
                                                    f"step = reset + shard score + hit all-reduce; value = set bases / step time",
                                        "shard_bases_this_rank": strong["n_bases"]})
        else:
            el = timed(weak_step, o_steps, o_warm)
            other = ("weak_scaling", {"value": n_bases * world * o_steps / el / 1e9, "unit": "Gbases/s", "ms_per_step": el / o_steps * 1e3, "steps": o_steps,
                                      "workload": f"{args.config} reads: every rank its own set of {n_bases / 1e6:.2f} Mbases per step + hit all-reduce"})
        # leave the context holding the primary read set for the solve below
        primary_step()
        dev_sync()

    # ---- the rest of the job, once: spectrum merge (N > 1), filter + exact solve
    t_solve = None
    res = None
    solve_info = None
    if not args.no_solve:
        dev_sync()
        t0 = time.perf_counter()
        if use_ipc:
            ctx.ipc_exchange()                                 # hit gather (idempotent) + the lists of novel read hashes through mapped buffers
        elif use_lib_comm:
            ctx.comm_exchange()                                # hit all-reduce (idempotent) + spectrum all-gather / import
        else:
            pdist.merge_spectrum_into(ctx, dev)
        res = ctx.solve()
        dev_sync()
        t_solve = time.perf_counter() - t0
        solve_info = ctx.solve_stats()
        solve_info["dp_mode_name"] = ("every vertex a step", "event chain", "blocks in parallel, walk lanes", "blocks in parallel, rows on class lanes")[solve_info["dp_mode"]]
        if world > 1:
            o = torch.tensor([res["objective"], -res["objective"], res["spectrum_size"], -res["spectrum_size"]], dtype=torch.int64, device=ctl_dev)
            dist.all_reduce(o, op=dist.ReduceOp.MAX)
            assert int(o[0].item()) == -int(o[1].item()), "ranks disagree on the objective"
            assert int(o[2].item()) == -int(o[3].item()), "ranks disagree on the spectrum size"
        assert res["optimal"] == 1, "the quoted configuration must be solved to proven optimality"

    # ---- the whole GPU path again on FRESH contexts (index build -> one read set -> solve): what the first numbers of
    #      a process owe to its start (code objects, first allocations of every size, idle clocks) shows as *_cold against
    #      the medians of the repeats
    repeats = None
    if not args.no_solve and world == 1 and args.job_repeats > 0 and walk_bases < 2e9:
        reps = []
        for _ in range(args.job_repeats):
            c2 = phi_amd.Context(local_rank)
            c2.set_params(k=K, w=W, threshold=1.0, recombination=100)
            c2.set_stream(stream.cuda_stream)
            dev_sync()
            t0 = time.perf_counter()
            c2.set_graph(A["seq_concat"], A["seq_off"], A["adj_off"], A["adj"], A["walk_off"], A["walk_vtx"], A["top_rank"])
            dev_sync()
            t1 = time.perf_counter()
            c2.add_reads_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, n_bases)
            dev_sync()
            t2 = time.perf_counter()
            r2 = c2.solve()
            dev_sync()
            t3 = time.perf_counter()
            assert r2["objective"] == res["objective"] and np.array_equal(r2["path_vtx"], res["path_vtx"]) and np.array_equal(r2["path_hap"], res["path_hap"]), "a fresh context gave another result"
            reps.append((t1 - t0, t2 - t1, t3 - t2))
            c2.close()
        med = lambda i: float(np.median([r[i] for r in reps]))
        repeats = {"n": len(reps), "index_build_s": med(0), "first_batch_s": med(1), "solve_s": med(2), "gpu_path_s": float(np.median([sum(r) for r in reps])),
                   "all": [[round(x, 6) for x in r] for r in reps]}

    # ---- roofline of the dominant kernel.  Algorithmic bytes per base by SURVEY.md 8d: 1 (ASCII) + 0.5 (packed 2-bit
    #      write + read) + 24 d (an 8-byte record and a 16-byte probe per emitted minimiser, density d).  Since the
    #      fusion of the preparation launch into the sketch kernel the packed form lives in LDS only: the kernel itself
    #      moves 1 + 24 d; `frac` keeps the SURVEY formula, `kernel_own_frac` prices the kernel by what it moves.
    b_packed, b_sketch = 0.5, 1.0 + 24.0 * density
    b_alg = b_packed + b_sketch
    kern_avg_ms = kern_ms / max(1, n_launch)
    launch_bases = kern_bases / max(1, n_launch)
    achieved = launch_bases * b_alg / (kern_avg_ms * 1e-3) / 1e9 if n_launch else 0.0
    per_rank_bases = n_bases if args.scaling == "weak" else strong["n_bases"]
    step_frac = per_rank_bases * b_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS

    # HBM bytes per launch from PMC counters: measured in a separate rocprofv3 run (bench.py cannot
    # profile itself) and kept, with the commands, under profiles/
    traffic, traffic_source = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if tj.get("config") == args.config and args.scaling == "weak":
            traffic = tj["bytes_per_launch"]
            traffic_source = tj.get("source", "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE run of this command in an earlier session, not this run")
    except (OSError, ValueError, KeyError):
        pass

    out = {
        "metric": "read Gbases/s scored (49-hap MHC graph stand-in synMHC-49); reads resident in HBM, parsing and H2D excluded",
        "value": value, "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"{args.config}: {'reference graph' if args.config == 'C1syn' else (f'native syn-{g.n_walks} graph' if native else f'synMHC-{g.n_walks} graph')} (seed {gk['seed']}{', backbone = ' + os.path.basename(gk['backbone_fasta']) if isinstance(gk, dict) and gk.get('backbone_fasta') else ''}, {g.n_walks} walks, {g.n_vtx} vertices{'' if args.config == 'C1syn' else ' <=30 bp'}, "
                               f"{walk_bases / 1e6:.1f} Mbases of walks) + " +
                               (f"{n_reads} reads of mean {n_bases / max(1, n_reads):.0f} bp per GPU (seed {rk['seed']}, {n_bases / 1e6:.2f} Mbases)" if args.scaling == "weak" else
                                f"ONE set of {strong['total_reads']} reads ({args.strong_config}, {strong['total_bases'] / 1e6:.2f} Mbases) in {world} shard(s)"),
                   "k": K, "w": W, "R": 100, "reads_per_gpu_bases": per_rank_bases, "parallelism": f"read-shard x{world}",
                   "exchange": "none (1 GPU)" if world == 1 else ("gloo through the host (rehearsal)" if args.rehearse_gloo else
                                ("mapped hit vectors (hipIpc*), one OR-gather kernel per read set on a stream of its own beside the next read set's scoring, inside libphi_amd.so (phi_ipc_*)"
                                 + (" -- rehearsal: all ranks on ONE GPU" if args.rehearse_ipc else "") if use_ipc else
                                 "RCCL all-reduce(MAX, uint8) of the hit vector per step, inside libphi_amd.so" + (f" ({exchange_note})" if exchange_note else "")))},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "phi_sketch_pool_kernel (batches of 12 Mbases and more)" if per_rank_bases >= 4 * 6144 * 512 else "phi_sketch_kernel<PHI_MODE_PROBE>", "kernel_avg_ms": kern_avg_ms,
                     "kernel_launches": n_launch, "kernel_timed": "a leg of its own after the timed steps, every launch bracketed", "bytes_per_base_algorithmic": b_alg, "minimiser_density": density,
                     "bytes_per_base_split": {"moved by phi_sketch_kernel (ASCII read, record, probe)": b_sketch,
                                              "packed 2-bit write + read of the SURVEY formula (stays in LDS since the fusion: not moved)": b_packed},
                     "kernel_own_frac": (launch_bases * b_sketch / (kern_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if n_launch else 0.0,
                     "step_frac": step_frac,
                     "note": "achieved/frac price the sketch kernel's time against the SURVEY 8d bytes of a base (1.5 + 24 d); kernel_own_frac prices it against "
                             "the bytes the fused kernel itself moves (1 + 24 d); step_frac = SURVEY bytes / whole step time (the step is this one launch)",
                     "kernel_gbases_per_s": launch_bases / (kern_avg_ms * 1e-3) / 1e9 if n_launch else 0.0},
        "spectrum_dedupe_ms": spectrum_dedupe_ms,
        "step_accounting": "a step = phi_reset_reads + phi_add_reads_device: sketch, hash, table probe (hit flags) and the coalesced LOG of the read hashes that are not walk "
                           "minimisers, one launch.  The set of those hashes (it only feeds |Sp_R| and the log counters derived from it, ILP_index.cpp:641, 738-743, 883) is "
                           "made from the log ONCE per read set, when |Sp_R| is first asked for: spectrum_dedupe_ms = that call (phi_reads_stats: set from the log + flag count, "
                           "host wall clock with its waits), outside the step, INSIDE solve_s / gpu_path_s / end_to_end_s of a job (phi_solve asks first there)",
        "reads_handed_over": "without offsets (one length: the kernel computes the read starts)" if not primary_step.uses_offsets else "with an offsets array",
        "value_with_offsets": value_with_offsets,
        "index_build_s": repeats["index_build_s"] if repeats else t_index, "index_build_s_cold": t_index,
        "graph_gbases_per_s": walk_bases / (repeats["index_build_s"] if repeats else t_index) / 1e9, "index": index_info,
        "solve_s": repeats["solve_s"] if repeats else t_solve, "solve_s_cold": t_solve, "solve": solve_info,
        # the GPU path alone, inputs in memory: index build + one step + solve (what rounds 1-2 called end_to_end_s)
        "gpu_path_s": (repeats["gpu_path_s"] if repeats else (t_index + ms_per_step * 1e-3 + t_solve)) if t_solve is not None else None,
        "gpu_path_s_cold": (t_index + ms_per_step * 1e-3 + t_solve) if t_solve is not None else None,
        "job_repeats": repeats,
        # process start -> FASTA closed, from files, with the command line (filled in below)
        "end_to_end_s": None, "end_to_end_one_process_s": None, "back_to_back_s": None, "end_to_end_detached_s": None, "several_jobs_per_read_set_s": None,
        "synthetic_gen_s": t_gen,
    }
    if step_split is not None:
        out["step_split"] = step_split
    if other is not None:
        out[other[0]] = other[1]
    if res is not None:
        out["result"] = {k_: int(res[k_]) for k_ in ("objective", "upper_bound", "optimal", "n_dp_runs", "n_covered",
                                                      "recombination_count", "n_switches", "spectrum_size", "filtered",
                                                      "n_in_model", "hap_len")}
        out["result"]["truth_walks"] = truth["walks"]
        ph = res["path_hap"]
        out["result"]["path_walks"] = [int(x) for x in ph[np.r_[True, ph[1:] != ph[:-1]]]] if len(ph) else []

    if rank == 0 and world == 1 and not args.no_extra_legs and n_bases < 2e9:
        # PCIe-inclusive rate: the same read set handed over as HOST buffers (pinned) through phi_add_reads
        hb = np.ascontiguousarray(bases)
        ctx._chk(ctx._L.phi_host_register(ctx._h, hb.ctypes.data, hb.nbytes))
        reps = max(3, min(20, int(2e8 // max(n_bases, 1))))
        for _ in range(2):
            ctx.reset_reads(); ctx.add_reads((hb, off))
        dev_sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.reset_reads(); ctx.add_reads((hb, off))
        dev_sync()
        out["h2d_inclusive_gbases_per_s"] = n_bases * reps / (time.perf_counter() - t0) / 1e9
        ctx._chk(ctx._L.phi_host_unregister(ctx._h, hb.ctypes.data))
        if walk_bases < 2e9:
            out["file_to_fasta"] = file_to_fasta(g, bases, off, K, W)
            f2f = out["file_to_fasta"] or {}
            out["file_to_fasta_s"] = f2f.get("seconds")
            # process start -> FASTA closed, from files.  end_to_end_s: ONE process, teardown inside, the median of
            # consecutive commands (= end_to_end_one_process_s: the key the review asked for); back_to_back_s: their total / their number;
            # end_to_end_detached_s: round 3's definition (the command returns at the closed FASTA, teardown behind it)
            out["end_to_end_s"] = f2f.get("seconds")
            out["end_to_end_one_process_s"] = f2f.get("seconds")
            out["back_to_back_s"] = f2f.get("back_to_back_s")
            out["end_to_end_detached_s"] = f2f.get("detached_seconds")
            out["several_jobs_per_read_set_s"] = f2f.get("several_jobs_per_read_set_s")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(A, bases, off, K, W)
        # the oracle's stages 1-2 over the WHOLE configuration (timed above as the CPU baseline's stage table) against what
        # the GPU path just reported for the same graph and read set: every counter of the reference's log
        # (ILP_index.cpp:563, 641, 725-743, 883).  A mismatch fails the run.
        stg = out["cpu_baseline"].pop("_stage12", None)
        if stg is not None and res is not None and args.scaling == "weak":      # (strong: the context holds the strong read set)
            checks = {"n_minimizers_per_walk": np.array_equal(res["n_minimizers"], stg.n_minimizers),
                      "spectrum_size": int(res["spectrum_size"]) == len(stg.spectrum),
                      "filtered": int(res["filtered"]) == int(stg.filtered), "retained": int(res["retained"]) == int(stg.retained),
                      "n_in_model": int(res["n_in_model"]) == int(stg.n_in_model),
                      "n_anchors_per_walk": np.array_equal(res["n_anchors"], stg.n_anchors),
                      "n_kept_anchors": int(res["n_anchors"].sum()) == len(stg.a_r)}
            out["parity_checked"] = {"against": "oracle/phi_oracle.c orc_run, stages 1-2 of this configuration and read set on the host cores",
                                     "ok": bool(all(checks.values())), "checks": {k_: bool(v) for k_, v in checks.items()},
                                     "values": {"spectrum_size": len(stg.spectrum), "filtered": int(stg.filtered), "retained": int(stg.retained),
                                                "n_in_model": int(stg.n_in_model), "n_walk_minimizers": int(stg.n_minimizers.sum()), "n_kept_anchors": len(stg.a_r)},
                                     "note": "per-walk (hash, position) arrays, the spectrum as a set and the kept anchors as a multiset are compared in "
                                             "tests/test_gpu_parity.py::test_full_size_vs_oracle on the same graph and reads"}
            if not out["parity_checked"]["ok"]:
                print(json.dumps(out["parity_checked"]), file=sys.stderr, flush=True)
                raise SystemExit("bench.py: the GPU path's counters differ from the oracle's on this configuration")
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
