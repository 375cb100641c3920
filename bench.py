#!/usr/bin/env python3
"""bench.py — walk-steps/sec of the `--cmd randomwalk` hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json metric: "walk-steps/sec ... on 1B-edge RMAT"): RMAT scale-26, edge factor 16
(1.07 B edge lines -> 2.15 B adjacency entries, 32.8 M present vertices), undirected, p = q = 1,
walkLength = 80, generated and built into CSR on the device (synthetic, seed 42).  One bench "step" = one walk
iteration (numWalks = 1): one walker per present vertex, 81 walk-steps each = 2.66e9 walk-steps.
Multi-GPU: the graph (≈72 GB with sampling tables) fits one 288 GB GPU, so it is replicated and the walk
iterations are sharded across ranks with NO data-path collective (walkers are independent; the keyed Philox
stream makes every path independent of which GPU computes it) -> "scaling": "weak" (each rank runs K
iterations).  `--shard vertex` instead runs the vertex-sharded path with the per-super-step walker
all-to-all (RCCL) that graphs beyond one GPU need.

Timed region: K walk iterations, inputs resident in HBM, outputs (paths) left in HBM; barrier +
torch.cuda.synchronize() on both sides; max over ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(args):
    """The CPU restatement of the reference algorithm (oracle/, kind "port": the Scala/Spark reference cannot be
    built here), timed on this box's host cores on a bounded sample.  The 1 B-edge graph cannot be assembled on
    the CPU inside the time budget, so the sample runs on the same RMAT family at scale 20 (BASELINE config 2)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_py
    cores = os.cpu_count() or 1
    scale = args.cpu_scale
    t0 = time.time()
    s, d = oracle_py.rmat_edges(scale, 16 << scale, seed=42)
    g = oracle_py.Graph.from_coo(s, d, None, directed=False)
    t_build = time.time() - t0
    verts = g.vertices()
    n_src = args.cpu_sources or max(64, 3 * cores)
    src = verts[np.linspace(0, len(verts) - 1, n_src).astype(np.int64)]
    # faithful = the reference's own O(deg(curr) * deg(prev)) computeSecondOrderWeights (linear `exists`)
    t0 = time.time()
    _, _, steps = g.walk(sources=src, p=args.p, q=args.q, walk_length=args.cpu_walk_length, num_walks=1, seed=42,
                         faithful=True, threads=cores)
    dt = time.time() - t0
    return {"value": steps / dt, "unit": "walk-steps/s", "cores": cores, "kind": "port",
            "sample": "CPU restatement of RandomSample/RandomWalk (faithful linear-exists variant), RMAT scale-%d ef16 "
                      "undirected p=%g q=%g, %d evenly spaced sources x %d steps, %d threads; %.1f s walk, %.1f s graph build"
                      % (scale, args.p, args.q, n_src, args.cpu_walk_length + 1, cores, dt, t_build)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=int, default=26, help="RMAT scale (26 = the 1B-edge headline graph)")
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--walk-length", type=int, default=80)
    ap.add_argument("--p", type=float, default=1.0)
    ap.add_argument("--q", type=float, default=1.0)
    ap.add_argument("--weighted", type=int, default=0)
    ap.add_argument("--directed", type=int, default=0)
    ap.add_argument("--sampler", choices=["reference", "alias"], default="reference")
    ap.add_argument("--shard", choices=["replicate", "vertex"], default="replicate")
    ap.add_argument("--nt-loads", type=int, default=-1, help="-1 auto, 0 cached, 1 nontemporal record loads")
    ap.add_argument("--compact", type=int, default=1, help="0: do not use the 16-byte lattice records")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-scale", type=int, default=20)
    ap.add_argument("--cpu-sources", type=int, default=0, help="0 = max(64, 3 per host core)")
    ap.add_argument("--cpu-walk-length", type=int, default=80)
    args = ap.parse_args()

    import torch
    import _pkg
    pkg = _pkg.load()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                         % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if "RANK" in os.environ:  # launched by torch.distributed.run (any world size, including 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        try:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        except TypeError:  # older signature without device_id
            dist.init_process_group(backend="nccl")

    def barrier_sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    n_edges = args.edge_factor << args.scale
    K, W = args.steps, args.warmup
    walk_kw = dict(p=args.p, q=args.q, walk_length=args.walk_length, num_walks=1, seed=42)
    if args.nt_loads >= 0:
        walk_kw["nt_loads"] = bool(args.nt_loads)
    if args.sampler == "alias":
        walk_kw["sampler"] = "alias"
    if not args.compact:
        walk_kw["compact"] = False

    if args.shard == "vertex" and dist is not None:
        from importlib import import_module
        sharded = import_module("stellar_random_walk_amd.distributed")
        drv = sharded.ShardedWalker(device=local_rank, rank=rank, world=world)
        drv.generate_rmat(args.scale, n_edges, seed=42, weighted=bool(args.weighted), directed=bool(args.directed))
        nv, ne = drv.engine.stats()
        for it in range(W):
            drv.walk_iteration(iteration=it, **walk_kw)
        barrier_sync()
        t0 = time.perf_counter()
        steps = 0
        kernel_ms = []
        for it in range(W, W + K):
            st = drv.walk_iteration(iteration=it, **walk_kw)
            steps += st["n_steps"]
            kernel_ms.append(st["kernel_ms"])
        barrier_sync()
        dt = time.perf_counter() - t0
        stats = {"ent_reads": 0, "n_walkers": 0, "kernel_kind": 2, "sum_deg_curr": st.get("sum_deg_curr", 0)}
        parallelism = "vertex-sharded x%d, RCCL all-to-all per super-step" % world
        scaling = "strong"
    else:
        eng = pkg.Engine(device=local_rank)
        eng.generate_rmat(args.scale, n_edges, seed=42, weighted=bool(args.weighted), directed=bool(args.directed))
        nv, ne = eng.stats()
        base = rank * (W + K)  # disjoint walk-iteration indices per rank: numWalks = world * K in total
        for it in range(W):
            eng.walk(fetch=False, first_walk=base + it, **walk_kw)
        barrier_sync()
        t0 = time.perf_counter()
        steps = 0
        kernel_ms = []
        stats = None
        for it in range(W, W + K):
            st = eng.walk(fetch=False, first_walk=base + it, **walk_kw)  # srw_walk: launch + hipEvents on its stream
            steps += st["n_steps"]
            kernel_ms.append(st["kernel_ms"])
            stats = st
        barrier_sync()
        dt = time.perf_counter() - t0
        parallelism = "graph replicated, walk iterations sharded x%d, no collective" % world if world > 1 else "1 GPU"
        scaling = "weak"

    total_steps, max_dt = steps, dt
    if dist is not None:
        t = torch.tensor([float(steps)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_steps = int(t.item())
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        max_dt = float(t.item())

    if rank == 0:
        avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        if stats["kernel_kind"] == 1:
            # first-order guide-table kernel, per launch: the linked CDF/guide records actually read
            # (counted by the kernel; 16 B compact lattice records or 32 B exact records) + 4 B path store per step; per walker 4 B seed + 16 B row + 4 B len
            per_launch_steps = steps / max(K, 1)
            alg_bytes = per_launch_steps * 4 + stats["ent_reads"] * stats["record_bytes"] + stats["n_walkers"] * 24
            kernel_name = "k_walk_first_order"
        elif stats["kernel_kind"] == 3:
            # Mode A: 32-B alias records read (counted) + 4 B path store per step; membership probes of
            # rejected/accepted candidates are not counted (lower bound)
            per_launch_steps = steps / max(K, 1)
            alg_bytes = per_launch_steps * 4 + stats["ent_reads"] * 32 + stats["n_walkers"] * 24
            kernel_name = "k_walk_alias"
        else:
            # general kernel (SURVEY §8d Mode R): 16 + 8*deg(curr) + 4 per step (+ 16 + 4*deg(prev) when q != 1) for the
            # steps that stream N(curr); steps served by the binned prefix-sum search count what their membership strategy
            # reads instead (P1: ids of N(prev) + the entries they land on; P2/P3: entries + hash slot / bitmap word; W: both sorted rows)
            per_launch_steps = steps / max(K, 1)
            alg_bytes = (per_launch_steps * 20 + stats["sum_deg_curr"] * 8 + stats.get("sum_deg_prev", 0) * 4
                         + stats.get("trials", 0))
            kernel_name = "k_walk_general"
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("kernel") == kernel_name and j.get("scale") == args.scale:
                    traffic = j.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "walk-steps/sec", "value": total_steps / max_dt, "unit": "walk-steps/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": max_dt / max(K, 1) * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f64 CDF tables (u32 lattice compares in the walk), int32 ids", "data": "synthetic",
            "config": {"workload": "RMAT scale-%d ef%d (%d edge lines, %d adjacency entries, %d vertices) %s "
                                   "%s p=%g q=%g walkLength=%d, 1 walk iteration per step, %s"
                                   % (args.scale, args.edge_factor, n_edges, ne, nv,
                                      "directed" if args.directed else "undirected",
                                      "weighted" if args.weighted else "unweighted", args.p, args.q, args.walk_length,
                                      "Mode A (alias + rejection)" if args.sampler == "alias" else "Mode R (reference-exact)"),
                       "walk_steps_per_bench_step": int(steps / max(K, 1)), "parallelism": parallelism,
                       "rng": "Philox4x32-10 keyed (iteration, source, step)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel_name,
                         "kernel_ms_avg": avg_ms, "algorithmic_bytes_per_launch": int(alg_bytes),
                         "record_bytes": stats.get("record_bytes", 0),
                         # context (profiles/r01_microbench_random_gather.txt): measured MI355X ceiling of dependent random
                         # reads, the access pattern of this kernel: 51.2e9 16-B records/s, 40.3e9 32-B records/s
                         "random_gather_ceiling_records_per_s": {16: 51.2e9, 32: 40.3e9}.get(stats.get("record_bytes", 0))},
        }
        if world == 1 and args.cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
