#!/usr/bin/env python3
"""bench.py — walk-steps/sec of the `--cmd randomwalk` hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

Headline workload (BASELINE.json metric: "walk-steps/sec ... on 1B-edge RMAT"): RMAT scale-26, edge factor 16
(1.07 B edge lines -> 2.15 B adjacency entries, 32.8 M present vertices), undirected, p = q = 1,
walkLength = 80, generated and built into CSR on the device (synthetic, seed 42).  One bench "step" = one walk
iteration (numWalks = 1): one walker per present vertex, 81 walk-steps each = 2.66e9 walk-steps.
Timed region: K walk iterations, inputs (graph + sampling tables) resident in HBM, outputs (paths) left in HBM;
barrier + torch.cuda.synchronize() on both sides; max over ranks.  Everything the timed region excludes is reported next
to it: `setup_s` (graph generation + CSR build + sampling tables) and `end_to_end` (one iteration through
srw_walk_and_save: walk + device formatter + PCIe + part files, text bytes per second).

After the headline, with one GPU, `configs` carries the other BASELINE.json configurations that fit one GPU — C2
(RMAT-20, p = q = 1), C3 (weighted RMAT-24, p = .25 q = 4; Mode R = bit-exact reference sampler, and Mode A = alias tables
+ rejection; the same graph with q = 1), C5's stand-in (directed RMAT-26 ef 27, p = 4 q = .5; Mode R and Mode A) — each with its own value, kernel
time, setup and roofline.

Multi-GPU (`--gpus N` under torch.distributed.run): `value` is the replicated mode (the graph fits one 288 GB GPU, so
it is replicated and the walk iterations are sharded across ranks with NO data-path collective; "scaling": "weak"),
and `vertex_sharded` reports north_star's split next to it (graph sharded by source vertex, walkers exchanged by an
RCCL all-to-all per super-step, paths kept on the home GPU; stellar-random-walk_amd/distributed.py).
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _rmat_weights_np(np, s, d, seed=42):
    """Vectorised oracle/srw_oracle.c:orc_rmat_weight: w = 1 + (mix32(min, max, seed) & 15)."""
    a = np.minimum(s, d).astype(np.uint32)
    b = np.maximum(s, d).astype(np.uint32)
    with np.errstate(over="ignore"):
        x = b * np.uint32(0x85EBCA77)
        h = np.uint32(seed) ^ (a * np.uint32(0x9E3779B1)) ^ ((x << np.uint32(13)) | (x >> np.uint32(19)))
        h ^= h >> np.uint32(16); h *= np.uint32(0x85EBCA6B); h ^= h >> np.uint32(13)
        h *= np.uint32(0xC2B2AE35); h ^= h >> np.uint32(16)
    return (np.uint32(1) + (h & np.uint32(15))).astype(np.float32)


def cpu_baseline(args):
    """The CPU restatement of the reference algorithm (oracle/, kind "port": the Scala/Spark reference cannot be built here —
    no JVM in the image, probed again below), timed on this box's host cores on BOUNDED samples of BASELINE.md §3's plan:
    karate, RMAT-14, RMAT-16 and an RMAT-20 sample, each at (p, q) = (1, 1) unweighted and (0.25, 4) weighted, the faithful
    variant (the reference's own O(deg(curr) * deg(prev)) linear `exists`, RandomSample.scala:27-44) and the fast one (sorted
    membership: same outputs).  The object's own value is the RMAT-20 (1, 1) faithful sample, as in rounds 1-2; `plan` lists
    every measurement, so that the biased GPU lines have their CPU number beside them.  The default run times the RMAT-20 samples only
    (`--cpu-plan sample`: ~30 s of CPU work, as the bench contract asks); `--cpu-plan full` the whole plan."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_py
    cores = os.cpu_count() or 1
    jvm = {t: shutil.which(t) for t in ("java", "spark-submit")}
    plan = []

    def timed(g, label, src, p, q, L, faithful):
        t0 = time.time()
        _, _, steps = g.walk(sources=src, p=p, q=q, walk_length=L, num_walks=1, seed=42, faithful=faithful, threads=cores)
        dt = max(time.time() - t0, 1e-9)
        e = {"workload": label, "p": p, "q": q, "variant": "faithful (linear exists, as the reference)" if faithful else "fast (sorted membership)",
             "value": steps / dt, "unit": "walk-steps/s", "cores": cores, "kind": "port", "walk_steps": int(steps), "seconds": dt,
             "sources": int(len(src))}
        plan.append(e)
        return e

    karate = os.path.join(ROOT, "tests", "golden", "karate.txt")
    if os.path.exists(karate):
        gk = oracle_py.Graph.load(karate, directed=False)
        for (p, q) in ((1.0, 1.0), (0.25, 4.0)):
            for faithful in (True, False):
                timed(gk, "karate.txt (34 vertices), walkLength 10", gk.vertices(), p, q, 10, faithful)
    head = None
    t_build = 0.0
    # --cpu-plan sample (default): the RMAT-20 samples only, ~30 s of CPU work in all (the object's own value, its fast variant, one biased
    # sample beside the biased GPU rows); --cpu-plan full: BASELINE.md §3's whole plan, ~100 s (profiles/r05f_bench_n1.json has one)
    full = args.cpu_plan == "full"
    sizes = ((14, 2048, 0), (16, 1024, 4096), (args.cpu_scale, args.cpu_sources or max(256, 4 * cores), 16384)) if full else \
            ((args.cpu_scale, args.cpu_sources or max(256, 4 * cores), 4096),)
    for scale, n_faithful, n_fast in sizes:
        t0 = time.time()
        s, d = oracle_py.rmat_edges(scale, 16 << scale, seed=42)
        graphs = {(1.0, 1.0): oracle_py.Graph.from_coo(s, d, None, directed=False),
                  (0.25, 4.0): oracle_py.Graph.from_coo(s, d, _rmat_weights_np(np, s, d), directed=False)}
        t_build += time.time() - t0
        for (p, q), g in graphs.items():
            verts = g.vertices()
            for faithful, n_src in ((True, n_faithful if (q == 1.0 or scale < args.cpu_scale) else max(64, n_faithful // (3 if full else 8))), (False, n_fast)):   # (the biased faithful walk is ~3x slower per step)
                if not full and not faithful and q != 1.0:
                    continue
                src = verts if (n_src == 0 or n_src >= len(verts)) else verts[np.linspace(0, len(verts) - 1, n_src).astype(np.int64)]
                e = timed(g, "RMAT scale-%d ef16 undirected %s, walkLength %d, %s" % (
                    scale, "unweighted" if q == 1.0 else "weighted", args.cpu_walk_length,
                    "every vertex" if len(src) == len(verts) else "%d evenly spaced sources" % len(src)), src, p, q, args.cpu_walk_length, faithful)
                if scale == args.cpu_scale and faithful and q == 1.0:
                    head = e
        del graphs
    if head is None:
        head = plan[-1]
    fast = next((e for e in plan if e["workload"].startswith("RMAT scale-%d" % args.cpu_scale) and e["q"] == 1.0 and e["variant"].startswith("fast")), None)
    return {"value": head["value"], "unit": "walk-steps/s", "cores": cores, "kind": "port",
            "walk_steps": head["walk_steps"], "seconds": head["seconds"], "value_per_core": head["value"] / max(cores, 1),
            # the same outputs with a sorted membership test instead of the reference's linear `exists` (what a CPU port that is not
            # bound to the reference's O(deg * deg) would run): the sample is 16 384 sources, >= 1e6 walk-steps
            "value_fast_variant": fast["value"] if fast else None, "walk_steps_fast_variant": fast["walk_steps"] if fast else None,
            "sample": "CPU restatement of RandomSample/RandomWalk (faithful linear-exists variant), %s, p=%g q=%g, %d threads; %.1f s walk; "
                      "graph builds of the whole plan %.1f s" % (head["workload"], head["p"], head["q"], cores, head["seconds"], t_build),
            "reference_toolchain_probe": {k: (v or "absent") for k, v in jvm.items()},
            "plan": plan}


def _library_commit():
    """The commit libstellar_rw.so was built from (srw_version(): 'stellar_rw gfx950 rN <commit>')."""
    try:
        import _pkg
        return _pkg.load().version().split()[-1]
    except Exception:
        return "?"


def pmc_child_pass(kernel_substr, one_walk_args, timeout_s=240):
    """HBM-side counters of the dominant kernel, taken IN this run: child `rocprofv3 --kernel-trace --pmc` passes over
    tools/one_walk.py with the same workload, in their own processes (the guide's rule: counters in a run of their own, --kernel-trace only;
    FETCH_SIZE needs three of the four TCC counters, so WRITE_SIZE and the request count share a second pass).  Returns
    {"hbm_bytes_per_launch", "requests_per_launch", ...} per launch of the kernel, or None when rocprofv3 is not on the box / a pass fails."""
    import csv, glob, subprocess
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    tool = os.path.join(ROOT, "tools", "one_walk.py")
    per_launch = {}
    t0 = time.perf_counter()
    for counters in (["FETCH_SIZE"], ["WRITE_SIZE", "TCC_EA0_RDREQ_sum"]):
        d = tempfile.mkdtemp(prefix="srw_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp", GRAFT_REPO_ROOT=ROOT)
            r = subprocess.run([exe, "--kernel-trace", "--pmc"] + counters + ["-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable, tool]
                               + [str(a) for a in one_walk_args], cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               text=True, timeout=timeout_s)
            acc, seen = {}, {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel_substr not in row["Kernel_Name"]:
                        continue
                    c = row["Counter_Name"]
                    acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
                    seen.setdefault(c, set()).add(row["Dispatch_Id"])
            if r.returncode != 0 or not acc:
                return None
            for c, v in acc.items():
                per_launch[c] = v / max(len(seen[c]), 1)
                per_launch["launches_" + c] = len(seen[c])
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if "FETCH_SIZE" not in per_launch or "WRITE_SIZE" not in per_launch:
        return None
    # FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE = TCC_EA0_RDREQ x 64 B; a wide coalesced
    # 16 B/lane stream is tallied at half its bytes — these kernels gather 16-byte records at random (one 64-byte request each), where
    # request x 64 B IS what crossed the fabric: no doubling.  WRITE_SIZE is uncalibrated for partial-line stores (taken as reported).
    return {"hbm_bytes_per_launch": (per_launch["FETCH_SIZE"] + per_launch["WRITE_SIZE"]) * 1024.0,
            "fetch_bytes_per_launch": per_launch["FETCH_SIZE"] * 1024.0, "write_bytes_per_launch": per_launch["WRITE_SIZE"] * 1024.0,
            "requests_per_launch": per_launch.get("TCC_EA0_RDREQ_sum"), "launches": per_launch.get("launches_FETCH_SIZE"),
            "seconds": time.perf_counter() - t0,
            "source": "measured in this run: child rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE | WRITE_SIZE + TCC_EA0_RDREQ_sum) over tools/one_walk.py "
                      + " ".join(str(a) for a in one_walk_args) + ", per launch of " + kernel_substr}


def apply_pmc(roof, pm, avg_ms, steps_per_launch, ceiling=None):
    """The counters of pmc_child_pass into a roofline object (traffic per launch, like `achieved`)."""
    if not pm:
        return roof
    roof["traffic"] = pm["hbm_bytes_per_launch"]
    roof["traffic_commit"] = _library_commit()
    roof["traffic_source"] = pm["source"]
    roof["physical_traffic_frac"] = pm["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    roof["traffic_over_algorithmic"] = pm["hbm_bytes_per_launch"] / max(roof.get("algorithmic_bytes_per_launch", 0), 1)
    if pm.get("requests_per_launch"):
        roof["requests_per_launch_pmc"] = pm["requests_per_launch"]
        roof["requests_per_step_pmc"] = pm["requests_per_launch"] / max(steps_per_launch, 1)
        if "requests_per_step" not in roof:
            roof["requests_per_step"] = roof["requests_per_step_pmc"]
            roof["requests_per_s"] = pm["requests_per_launch"] / (avg_ms * 1e-3)
            if ceiling:
                roof["request_rate_frac"] = roof["requests_per_s"] / ceiling["reads_per_s"]
    roof["pmc_pass_seconds"] = pm["seconds"]
    return roof


def promote_vertex_sharded(out, vs, world):
    """At N > 1 the line's `value` is north_star's split — the graph sharded by source vertex, one all_to_all per super-step (the RCCL driver's
    leg; the one-process cluster driver if that leg failed) — on the headline graph; the replicated weak-scaling figure (no collective: it
    says nothing about the shard / exchange design) moves to `replicated_weak_scaling`.  VERDICT r05 item 4; replaces the shuffle of
    RandomWalk.scala:92-93,186-192."""
    key = next((k for k in ("rccl", "cluster") if isinstance(vs.get(k), dict) and vs[k].get("value")), None)
    if key is None:
        out["vertex_sharded_note"] = "no vertex-sharded leg finished: `value` is the replicated weak-scaling figure"
        return out
    leg = vs[key]
    top = {k: out[k] for k in ("metric", "unit", "higher_is_better", "vs_baseline", "dtype", "data") if k in out}
    top.update({"value": leg["value"], "n_gpus": world, "steps": leg.get("steps", out.get("steps")), "warmup": leg.get("warmup", out.get("warmup")),
                "ms_per_step": leg["ms_per_step"], "scaling": "strong",
                "config": {"workload": leg.get("workload", out["config"].get("workload")),
                           "parallelism": "source-vertex shards x%d, one all_to_all per super-step (%s); paths on the home GPU"
                                          % (world, "RCCL all_to_all_single over xGMI, one process per GPU" if key == "rccl"
                                             else "one process driving all devices, chunks stored into the peers' buffers over xGMI"),
                           "driver": key, "rng": out["config"].get("rng")},
                "per_superstep_unoverlapped": leg.get("per_superstep_unoverlapped"), "exchange_model": leg.get("exchange_model"),
                "roofline": out.get("roofline"),
                "roofline_note": "the replicated single-GPU kernel's (k_walk_first_order on this rank's copy of the graph): the sharded step's kernels are timed apart in per_superstep_unoverlapped",
                "replicated_weak_scaling": {"value": out["value"], "ms_per_step": out["ms_per_step"], "scaling": "weak", "steps": out.get("steps"),
                                            "parallelism": out["config"].get("parallelism"), "walk_steps_per_bench_step": out["config"].get("walk_steps_per_bench_step")}})
    for k in ("setup_s", "end_to_end"):
        if k in out:
            top[k] = out[k]
    return top


def roofline_of(stats, steps_per_launch, avg_ms, scale=None, n_entries=None, ceiling=None, scan=None, q=1.0, wl=None):
    """Algorithmic bytes per launch / average kernel time of the dominant kernel."""
    kind = stats["kernel_kind"]
    if kind == 1:
        # first-order guide-table kernel (§4.3): linked CDF/guide records actually read (counted by the kernel; 16 B compact
        # or 32 B exact) + 4 B path store per step; per walker 4 B seed + 16 B row + 4 B len
        alg = steps_per_launch * 4 + stats["ent_reads"] * stats["record_bytes"] + stats["n_walkers"] * 24
        name = "k_walk_first_order"
        formula = "records_read*record_bytes + 4 B path/step + 24 B/walker (the O(deg) scan of SURVEY §8d is replaced by an exact precomputed CDF + guide table: its 16+8*deg+4 B/step are never issued)"
    elif kind == 3:
        # Mode A (§4.6): 32-B alias records read (counted) + 4 B path store per step (membership probes not counted)
        alg = steps_per_launch * 4 + stats["ent_reads"] * 32 + stats["n_walkers"] * 24
        name = "k_walk_alias"
        formula = "alias records read*32 + 4 B path/step + 24 B/walker (probes of the rejection test not counted: lower bound)"
    elif kind == 2 and stats.get("strategy_steps", {}).get("q1_lane", 0) > 0.5 * max(stats["n_steps"], 1):
        # p != 1, q == 1, one walker per lane (§4.8): per step 4 B path + 4 B rev + 8 B return edge (position, weight) + 8 B
        # row sum; + the guide records (16 B) and prefix probes (8 B) the kernel counts, taken at 12 B each
        alg = steps_per_launch * 24 + stats["ent_reads"] * 12 + stats["n_walkers"] * 24
        name = "k_walk_q1"
        formula = "24 B/step (path, rev, return edge, row sum) + counted guide records / prefix probes * 12 B + 24 B/walker"
    else:
        # general kernel (§4.4, SURVEY §8d Mode R): 16 B row + 4 B path per step; + what the step's sampler reads:
        # streamed rows 8*deg(curr) (+ 4*deg(prev) of membership), searches their strategy's bytes, per-edge tables
        # 512 B + 16 B per evaluated candidate, membership masks 8*deg + mask words (all counted by the kernel)
        alg = steps_per_launch * 20 + stats["sum_deg_curr"] * 8 + stats.get("sum_deg_prev", 0) * 4 + stats.get("trials", 0)
        name = "k_walk_tables + k_walk_general" if stats.get("edge_tables", 0) else "k_walk_general"
        formula = "20 B/step + 8*deg(curr) (+4*deg(prev)) for streamed rows + bytes read by the searches / per-edge tables / masks (kernel counters)"
    achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": None, "kernel": name, "kernel_ms_avg": avg_ms, "algorithmic_bytes_per_launch": int(alg),
         "algorithmic_bytes_formula": formula, "record_bytes": stats.get("record_bytes", 0)}
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc) and scale is not None:
        try:
            js = json.load(open(pmc))
            lib_commit = _library_commit()
            for j in (js if isinstance(js, list) else [js]):
                # an entry stands for ONE workload: kernel, scale and (ef, p, q, weighted, directed) must all agree — and for ONE library:
                # counters of another commit are not this run's traffic: such an entry is skipped, the keys stay absent
                # unless the run measures them itself (pmc_child_pass below)
                if j.get("commit") != lib_commit:
                    continue
                if j.get("kernel") == name and j.get("scale") == scale and all(
                        float(j.get(k, -1)) == float(v) for k, v in (wl or {}).items()):
                    r["traffic"] = j.get("hbm_bytes_per_launch")
                    r["traffic_commit"] = j.get("commit", "?")
                    r["traffic_source"] = ("profiles/pmc_latest.json <- %s (commit %s): separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ "
                                           "passes over the same command; the PMC counters cannot be read from inside bench.py" % (j.get("source", "profiles/"), j.get("commit", "?")))
                    if j.get("requests_per_launch"):
                        r["requests_per_launch_pmc"] = j["requests_per_launch"]
                    if r["traffic"]:
                        r["physical_traffic_frac"] = r["traffic"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        except Exception:
            pass
    # The fixed denominator of SURVEY §8(d): what the REFERENCE's algorithm reads for the same walk (RandomSample.sample scans N(curr);
    # computeSecondOrderWeights scans N(prev) when q != 1) — counted from the finished paths by srw_result_scan_sums, whatever kernel
    # and tables produced them, so that the figure is comparable across rounds.  (`achieved` above uses what THIS kernel reads.)
    if scan:
        sdc, sdp, nst = scan
        second = max(nst - stats["n_walkers"], 0)
        b = 20 * nst + 8 * sdc + ((16 * second + 4 * sdp) if q != 1.0 else 0)
        r["algorithmic_bytes_scan"] = int(b)
        r["algorithmic_bytes_scan_per_step"] = b / max(nst, 1)
        r["algorithmic_bytes_scan_formula"] = "SURVEY 8(d) Mode R: 16 + 8*deg(curr) + 4 per step (+ 16 + 4*deg(prev) per second-order step when q != 1), summed over the finished paths (srw_result_scan_sums)"
        r["scan_equivalent_GBs"] = b / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        r["scan_equivalent_frac"] = r["scan_equivalent_GBs"] / HBM_PEAK_GBS
    # The bound of these kernels is the L2-miss REQUEST rate (one request per record / probe whatever its size), not bytes
    # (profiles/r02_translation_and_request_rate.md).  The ceiling is MEASURED on this box before the headline (srw_probe_request_rate:
    # dependent random 16-byte reads over a table of the headline's size); requests of the kernel: counted by the kernel itself for the
    # first-order walk (records read + 1/16 path-store sector per step), from this round's TCC_EA0_RDREQ pass (profiles/pmc_latest.json)
    # for the table kernels.
    table_bytes = (n_entries or 0) * (stats.get("record_bytes") or 0)
    if ceiling:
        r["request_rate_ceiling"] = ceiling["reads_per_s"]
        r["request_rate_source"] = "measured in this run: srw_probe_request_rate, %.1f GiB table, dependent random 16-byte reads (csrc/probe.hip)" % ceiling["table_gib"]
    if kind == 1 and stats["ent_reads"] and table_bytes > (288 << 20):
        # Printed only for a table that the caches cannot hold (32 MiB of L2 + 256 MiB of Infinity Cache): a cache-resident table
        # (config 2) is not subject to that ceiling.
        req = (stats["ent_reads"] + steps_per_launch / 16.0) / (avg_ms * 1e-3)
        r["requests_per_s"] = req
        r["requests_per_step"] = (stats["ent_reads"] + steps_per_launch / 16.0) / max(steps_per_launch, 1)
        if ceiling:
            r["request_rate_frac"] = req / ceiling["reads_per_s"]
    elif r.get("requests_per_launch_pmc"):
        r["requests_per_step"] = r["requests_per_launch_pmc"] / max(steps_per_launch, 1)
        r["requests_per_s"] = r["requests_per_launch_pmc"] / (avg_ms * 1e-3)
        if ceiling:
            r["request_rate_frac"] = r["requests_per_s"] / ceiling["reads_per_s"]
    return r


def measure(eng, walk_kw, K, W, first_walk=0):
    """W + K walk iterations of one walker per vertex; returns (steps, wall seconds of the K, kernel_ms list, last stats,
    setup seconds spent inside the calls)."""
    import torch
    setup_ms = 0.0
    for it in range(W):
        st = eng.walk(fetch=False, first_walk=first_walk + it, **walk_kw)
        setup_ms += st["setup_ms"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps, kernel_ms, stats, inner_setup = 0, [], None, 0.0
    for it in range(W, W + K):
        st = eng.walk(fetch=False, first_walk=first_walk + it, **walk_kw)  # srw_walk: launch + hipEvents on its stream
        steps += st["n_steps"]
        kernel_ms.append(st["kernel_ms"])
        inner_setup += st["setup_ms"]
        stats = st
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0 - inner_setup * 1e-3
    return steps, dt, kernel_ms, stats, (setup_ms + inner_setup) * 1e-3


def run_config(pkg, device, name, scale, ef, weighted, directed, p, q, sampler, K, W, L=80, ceiling=None, plan_walks=0, pmc=False):
    """One BASELINE configuration on one GPU: graph generated on the device, W + K walk iterations.  pmc: afterwards (the engine closed: its
    tables fill most of the GPU) the child rocprofv3 --pmc passes over the same workload — the row's traffic and request count, in this run."""
    out = _run_config(pkg, device, name, scale, ef, weighted, directed, p, q, sampler, K, W, L, ceiling, plan_walks)
    if pmc and "roofline" in out and time.perf_counter() - T_START < 330:
        spec = "%d%s%s" % (scale, "w" if weighted else "", "d" if directed else "")
        pm = pmc_child_pass("k_walk_tables", [spec, p, q, sampler, 2, ef], timeout_s=150)
        if pm is not None:
            apply_pmc(out["roofline"], pm, out["kernel_ms"], out["walk_steps_per_bench_step"], ceiling)
    return out


def _run_config(pkg, device, name, scale, ef, weighted, directed, p, q, sampler, K, W, L=80, ceiling=None, plan_walks=0):
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng = pkg.Engine(device=device)
    try:
        eng.generate_rmat(scale, ef << scale, seed=42, weighted=weighted, directed=directed)
        nv, ne = eng.stats()
        if plan_walks:                             # the job's --numWalks (srw_plan_walks): a long job gets the finer per-edge tables
            eng.plan_walks(plan_walks)
        torch.cuda.synchronize()                   # device-wide: the build runs on the engine's own stream
        t_graph = time.perf_counter() - t0
        kw = dict(p=p, q=q, walk_length=L, num_walks=1, seed=42)
        if sampler == "alias":
            kw["sampler"] = "alias"
        steps, dt, kms, st, t_tables = measure(eng, kw, K, W)
        avg_ms = sum(kms) / max(len(kms), 1)
        try:
            scan = eng.result_scan_sums()
        except Exception:
            scan = None
        out = {"name": name,
               "workload": "RMAT scale-%d ef%d %s %s p=%g q=%g walkLength=%d, %s" % (
                   scale, ef, "directed" if directed else "undirected", "weighted" if weighted else "unweighted", p, q, L,
                   "Mode A (alias + rejection)" if sampler == "alias" else "Mode R (reference-exact)"),
               "vertices": nv, "adjacency_entries": ne, "value": steps / dt, "unit": "walk-steps/s",
               "steps": K, "warmup": W, "ms_per_step": dt / max(K, 1) * 1e3, "kernel_ms": avg_ms,
               "walk_steps_per_bench_step": int(steps / max(K, 1)),
               "setup_s": {"graph_generate_and_csr": t_graph, "sampling_tables": t_tables},
               "roofline": roofline_of(st, steps / max(K, 1), avg_ms, scale=scale, n_entries=ne, ceiling=ceiling, scan=scan, q=q,
                                       wl=dict(ef=ef, p=p, q=q, weighted=int(weighted), directed=int(directed), planned_walks=plan_walks))}
        if st["kernel_kind"] == 2:
            out["strategy_steps"] = {k: v for k, v in st["strategy_steps"].items() if v}
            out["edge_tables"] = {"count": st["edge_tables"], "bytes": st["edge_table_bytes"]}
            # the whole job of the config's numWalks = 10 from a cold start: tables once + 10 iterations
            out["job_numWalks10_steps_per_s"] = 10 * steps / max(K, 1) / (t_tables + 10 * dt / max(K, 1))
            if plan_walks:
                out["planned_walks"] = plan_walks
                out["job_planned_steps_per_s"] = plan_walks * steps / max(K, 1) / (t_tables + plan_walks * dt / max(K, 1))
        if st["kernel_kind"] == 3:
            out["trials_per_step"] = st["trials"] / max(st["n_steps"], 1)
        return out
    finally:
        eng.close()


def run_sharded_world1(pkg, device, scale=24, K=4, L=80):
    """The vertex-sharded protocol (chunks, home-rank paths, row links, fused sample-and-bucket kernel) at world = 1 on this
    GPU, next to the single-launch replicated kernel on the same graph — the only sharded measurement a 1-GPU run allows;
    what it shows is the cost of the super-step machinery, not scaling."""
    import torch
    kw = dict(p=1.0, q=1.0, walk_length=L, seed=42)
    with pkg.Engine(device=device) as eng:
        eng.generate_rmat(scale, 16 << scale, seed=42)
        nv, ne = eng.stats()
        eng.walk(fetch=False, num_walks=1, **kw)
        st = eng.walk(fetch=False, num_walks=K, first_walk=1, **kw)
        rep = st["n_steps"] / (st["kernel_ms"] * 1e-3)
    torch.cuda.synchronize()
    with pkg.Cluster([device]) as cl:
        t0 = time.perf_counter()
        cl.generate_rmat(scale, 16 << scale, seed=42)
        torch.cuda.synchronize()
        t_graph = time.perf_counter() - t0
        t0 = time.perf_counter()
        cl.walk(fetch=False, num_walks=1, batch=1, **kw)                     # tables, row links, buffers
        torch.cuda.synchronize()
        t_tables = time.perf_counter() - t0
        cl.walk(fetch=False, num_walks=K, first_walk=1, batch=K, **kw)
        st = cl.walk(fetch=False, num_walks=K, first_walk=1 + K, batch=K, **kw)
        val = st["n_steps"] / (st["kernel_ms"] * 1e-3)
    return {"name": "sharded w1 p=q=1",
            "workload": "RMAT scale-%d ef16 undirected unweighted p=1 q=1 walkLength=%d, Mode R; %d walk iterations as one walker "
                        "population (srw_cluster_walk: %d super-steps, 2 kernels each)" % (scale, L, K, L + 1),
            "vertices": nv, "adjacency_entries": ne, "value": val, "unit": "walk-steps/s", "steps": K, "warmup": K,
            "ms_per_step": st["kernel_ms"] / K, "replicated_kernel_same_graph": rep, "fraction_of_replicated": val / rep,
            "setup_s": {"graph_generate_and_csr": t_graph, "tables_row_links_first_walk": t_tables},
            "timed": "wall time of srw_cluster_walk (all super-steps of K iterations, no host sync inside)"}


XGMI_LINK_GBS = 153.0           # prompt / MI355X_MICROARCH.md: 7 xGMI links x ~153 GB/s per GPU, point to point
WIRE_BYTES_PER_WALKER_STEP = 24  # 16-byte walker + 8-byte path return (csrc/walk_kernels.hip: WWalker, WRet)


def exchange_model(world, steps_per_s=None):
    """What the walker exchange of the vertex-sharded walk allows: every walker-step moves one 16-byte walker
    to owner(next) and one 8-byte return to owner(source); with a mixing owner function (world - 1) / world of them cross
    GPUs, spread evenly over the world - 1 peers of a fully connected node, each over its own link."""
    if world < 2:
        return None
    cross = (world - 1) / world
    payload = WIRE_BYTES_PER_WALKER_STEP * cross
    egress = XGMI_LINK_GBS * 1e9 * (world - 1)             # all links busy at once: the all-to-all's pattern
    m = {"wire_bytes_per_walker_step": WIRE_BYTES_PER_WALKER_STEP, "crossing_fraction": cross,
         "xgmi_bytes_per_walker_step": payload, "xgmi_GBs_per_link": XGMI_LINK_GBS, "links_used": world - 1,
         "ceiling_walker_steps_per_s_per_gpu": egress / payload, "ceiling_walker_steps_per_s_job": world * egress / payload,
         "note": "payload bound; the RCCL driver ships whole fixed-capacity chunks (slack 1.25: x1.25 / fill), the in-process driver "
                 "stores payload only.  NOT measured: a prediction for the first multi-GPU run to be held against."}
    if steps_per_s:
        m["measured_fraction_of_exchange_ceiling"] = steps_per_s / m["ceiling_walker_steps_per_s_job"]
    return m


def run_sharded_biased_world1(pkg, device, name, scale, ef, weighted, directed, p, q, replicated_value, K=1, L=80, batch2=False):
    """A biased configuration through the vertex-sharded protocol at world = 1.  q != 1: the shard holds the per-edge tables
    of the pairs into its own rows behind the pair hash (edge_tables.hip:prepare_shard_tables), walkers arrive as 16-byte
    records, every super-step is the lean table step + the chain kernels + one fused bucketing pass.  q == 1 (p != 1): one record
    per lane behind the return-edge hash (k_sh_step_q1), pairs with many parallel return edges one wave each.  What one GPU can show
    is the cost of that machinery against the single-launch kernel on the same graph (`fraction_of_replicated`)."""
    import torch
    kw = dict(p=p, q=q, walk_length=L, seed=42)
    torch.cuda.synchronize()
    with pkg.Cluster([device]) as cl:
        t0 = time.perf_counter()
        cl.generate_rmat(scale, ef << scale, seed=42, weighted=weighted, directed=directed)
        nv, ne = cl.stats()
        torch.cuda.synchronize()
        t_graph = time.perf_counter() - t0
        t0 = time.perf_counter()
        cl.walk(fetch=False, num_walks=1, batch=1, **dict(kw, walk_length=1))     # membership, prefix sums, tables, hash: outside the timed walk
        torch.cuda.synchronize()
        t_tables = time.perf_counter() - t0
        st = cl.walk(fetch=False, num_walks=K, first_walk=1, batch=K, **kw)
        val = st["n_steps"] / (st["kernel_ms"] * 1e-3)
        two = None
        if batch2:            # a numWalks >= 2 job walks several iterations per population: the per-super-step launches are shared
            try:
                st2 = cl.walk(fetch=False, num_walks=2, first_walk=1 + K, batch=2, **kw)
                two = {"value": st2["n_steps"] / (st2["kernel_ms"] * 1e-3), "ms_per_iteration": st2["kernel_ms"] / 2.0}
            except Exception as ex:
                two = {"error": str(ex)[:200]}
    out = {"name": name,
           "workload": "RMAT scale-%d ef%d %s %s p=%g q=%g walkLength=%d, Mode R; %d walk iteration(s) as one population through "
                       "srw_cluster_walk at world 1" % (scale, ef, "directed" if directed else "undirected",
                                                         "weighted" if weighted else "unweighted", p, q, L, K),
           "vertices": nv, "adjacency_entries": ne, "value": val, "unit": "walk-steps/s", "steps": K, "warmup": 0,
           "ms_per_step": st["kernel_ms"] / K, "strategy_steps": {k: v for k, v in st["strategy_steps"].items() if v},
           "edge_tables": {"count": st["edge_tables"], "bytes": st["edge_table_bytes"]},
           "setup_s": {"graph_generate_and_csr": t_graph, "sampling_tables_pair_hash_first_walk": t_tables},
           "timed": "wall time of srw_cluster_walk (all super-steps, no host sync inside)"}
    if replicated_value:
        out["replicated_kernel_same_graph"] = replicated_value
        out["fraction_of_replicated"] = val / replicated_value
    if two:
        if replicated_value and "value" in two:
            two["fraction_of_replicated"] = two["value"] / replicated_value
        out["two_iterations_per_population"] = two
    return out


def cluster_leg(pkg, torch, args, world, K):
    """Vertex-sharded walk of the whole job on `world` devices driven by ONE process (srw_cluster_*): peer stores over xGMI,
    super-steps ordered by events, paths on the home GPU.  Returns the `vertex_sharded` object of the bench line."""
    n_edges = args.edge_factor << args.scale
    try:
        if not args.share_device and torch.cuda.device_count() < world:
            raise RuntimeError("the process sees %d devices, needs %d" % (torch.cuda.device_count(), world))
        t0 = time.perf_counter()
        with pkg.Cluster([0] * world if args.share_device else list(range(world)), membership=(args.q != 1.0)) as cl:       # q == 1: memory per shard ~ 1 / world
            cl.generate_rmat(args.scale, n_edges, seed=42, weighted=bool(args.weighted), directed=bool(args.directed))
            cnv, cne = cl.stats()
            torch.cuda.synchronize()
            t_graph = time.perf_counter() - t0
            kw = dict(p=args.p, q=args.q, walk_length=args.walk_length, seed=42)
            B = max(1, min(K, 4))
            cl.walk(fetch=False, num_walks=K, first_walk=0, batch=B, **kw)          # warm-up: tables + buffers
            st = cl.walk(fetch=False, num_walks=K, first_walk=K, batch=B, **kw)
            dt_v = st["kernel_ms"] * 1e-3
            return {"value": st["n_steps"] / dt_v, "unit": "walk-steps/s", "ms_per_step": dt_v / max(K, 1) * 1e3,
                    "scaling": "strong", "steps": K, "warmup": K, "iterations_per_population": B,
                    "workload": "RMAT scale-%d (%d edge lines, %d adjacency entries, %d vertices) p=%g q=%g walkLength=%d"
                                % (args.scale, n_edges, cne, cnv, args.p, args.q, args.walk_length),
                    "parallelism": "graph sharded by source vertex x%d (owner = mix32(id) mod world), one process driving all devices: "
                                   "chunks stored into the peers' buffers over xGMI, super-steps ordered by events, paths on the "
                                   "home GPU, no host sync per super-step" % world,
                    "timed": "wall time of the super-steps of K iterations (srw_cluster_walk), result buffers preallocated",
                    "strategy_steps": {k: v for k, v in st["strategy_steps"].items() if v},
                    "edge_tables": {"count": st["edge_tables"], "bytes": st["edge_table_bytes"]},
                    "exchange_model": exchange_model(world, st["n_steps"] / dt_v),
                    "setup_s": {"graph_generate_and_csr_all_shards": t_graph}}
    except Exception as ex:
        return {"error": str(ex)[:300]}


ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_avg", "algorithmic_bytes_per_launch",
                 "record_bytes", "physical_traffic_frac", "requests_per_step", "requests_per_s", "request_rate_ceiling",
                 "request_rate_frac", "scan_equivalent_frac", "traffic_commit")
SUMMARY_KEYS = ("name", "value", "ms_per_step", "kernel_ms", "job_numWalks10_steps_per_s", "job_planned_steps_per_s", "fraction_of_replicated", "error")
T_START = time.perf_counter()
LINE_LIMIT = 4096               # the driver reads the LAST stdout line; round 4's 24 KB line was not parsed


def _round(x, digits=6):
    """Numbers of the compact line at 6 significant digits (the detail file keeps them whole)."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _round(v, digits) for k, v in x.items()}
    if isinstance(x, list):
        return [_round(v, digits) for v in x]
    return x


def run_embedding_stage(pkg, device, scale):
    """SURVEY §8 (f) rank 4 behind config 2's graph: one walk iteration, then ONE training iteration of skip-gram + hierarchical
    softmax over the paths where they are (srw_w2v_fit_device; dim 128, window 10: Params.scala:7-23's defaults)."""
    eng = pkg.Engine(device=device)
    try:
        eng.generate_rmat(scale, 16 << scale, seed=42)
        st = eng.walk(fetch=False, walk_length=80, num_walks=1, seed=42, p=1.0, q=1.0)
        t0 = time.perf_counter(); ids, _ = eng.w2v_fit_device(iterations=0, seed=7); t_vocab = time.perf_counter() - t0
        t0 = time.perf_counter(); eng.w2v_fit_device(iterations=1, seed=7); t_fit = time.perf_counter() - t0
        words = int(st["n_steps"]) + int(st.get("n_walkers", 0))
        return {"what": "RMAT-%d, 1 walk per vertex (L = 80), Word2Vec dim 128 window 10, one training iteration, paths stay in HBM" % scale,
                "words": words, "vocabulary": int(len(ids)), "vocabulary_and_init_s": t_vocab, "training_iteration_s": max(t_fit - t_vocab, 1e-9),
                "words_per_s": words / max(t_fit - t_vocab, 1e-9)}
    finally:
        eng.close()


def compact_line(out):
    """The ONE stdout line: the driver's keys, `config`, `roofline` (numbers only), `cpu_baseline` without its plan, a
    `configs_summary` of the other configurations.  Everything else lives in the detail file."""
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                             "vs_baseline", "dtype", "data", "config") if k in out}
    if "roofline" in out:
        c["roofline"] = {k: out["roofline"][k] for k in ROOFLINE_KEYS if k in out["roofline"]}
    if "setup_s" in out:
        c["setup_s"] = {k: v for k, v in out["setup_s"].items() if not isinstance(v, str)}
    if "end_to_end" in out:
        c["end_to_end"] = {k: v for k, v in out["end_to_end"].items() if k in ("seconds", "walk_steps_per_s", "text_GB_per_s", "part_files", "skipped", "error")}
        if isinstance(out["end_to_end"].get("parts_200"), dict):
            c["end_to_end"]["parts_200_walk_steps_per_s"] = out["end_to_end"]["parts_200"].get("walk_steps_per_s")
        if isinstance(out["end_to_end"].get("first_call"), dict):
            c["end_to_end"]["first_call_seconds"] = out["end_to_end"]["first_call"].get("seconds")
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        c["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "value_fast_variant", "walk_steps", "seconds") if k in cb}
    if "configs" in out:
        rows = []
        for cf in out["configs"]:
            row = {k: cf[k] for k in SUMMARY_KEYS if k in cf}
            rf = cf.get("roofline") or {}
            for k in ("frac", "requests_per_step", "request_rate_frac"):
                if k in rf:
                    row[k] = rf[k]
            if "setup_s" in cf:
                row["setup_s"] = sum(v for v in cf["setup_s"].values() if isinstance(v, (int, float)))
            rows.append(row)
        c["configs_summary"] = rows
    if "vertex_sharded" in out:
        vs = {}
        for leg, v in out["vertex_sharded"].items():
            vs[leg] = {k: v[k] for k in ("value", "ms_per_step", "scaling", "steps", "leg_wall_s", "error") if k in v}
            m = v.get("exchange_model") or {}
            if "measured_fraction_of_exchange_ceiling" in m:
                vs[leg]["fraction_of_exchange_ceiling"] = m["measured_fraction_of_exchange_ceiling"]
        c["vertex_sharded"] = vs
    if isinstance(out.get("replicated_weak_scaling"), dict):      # N > 1: `value` above is the vertex-sharded walk (promote_vertex_sharded)
        c["replicated_weak_scaling"] = {k: v for k, v in out["replicated_weak_scaling"].items() if k in ("value", "ms_per_step", "scaling", "steps")}
    if isinstance(out.get("per_superstep_unoverlapped"), dict):
        c["per_superstep_unoverlapped"] = {k: v for k, v in out["per_superstep_unoverlapped"].items() if isinstance(v, (int, float))}
    if "embedding_stage" in out:
        c["embedding_stage"] = {k: v for k, v in out["embedding_stage"].items() if k in ("words_per_s", "training_iteration_s", "error")}
    if out.get("switches_set"):
        c["switches_set"] = out["switches_set"]
    if out.get("library"):
        c["library"] = out["library"]
    if "detail_file" in out:
        c["detail_file"] = out["detail_file"]
    c = _round(c)
    line = json.dumps(c, separators=(",", ":"))
    # never hand the driver a line it cannot read: shed the optional objects, largest first
    for k in ("configs_summary", "vertex_sharded", "end_to_end", "embedding_stage", "setup_s"):
        if len(line) <= LINE_LIMIT:
            break
        if k == "configs_summary" and k in c:
            for row in c[k]:
                row["name"] = row["name"][:24]
            line = json.dumps(c, separators=(",", ":"))
            if len(line) <= LINE_LIMIT:
                break
        c.pop(k, None)
        line = json.dumps(c, separators=(",", ":"))
    return line


def emit(out, args):
    """Full detail -> bench_detail.json (next to gpurun_out/ when that exists, else the working directory) and stderr;
    the compact object -> the last (and only) stdout line."""
    # tools/SWITCHES.md: the line describes the defaults unless it says otherwise
    try:
        import _pkg
        out["library"] = _pkg.load().version()
    except Exception:
        pass
    out["switches_set"] = sorted(k for k in os.environ if k.startswith("SRW_") and k not in ("SRW_TIMING",))
    full = json.dumps(out)
    path = args.detail or os.path.join("gpurun_out" if os.path.isdir("gpurun_out") else ".", "bench_detail.json")
    try:
        with open(path, "w") as f:
            f.write(full + "\n")
        out["detail_file"] = path
    except OSError as ex:
        out["detail_file"] = "not written: %s" % ex
    sys.stderr.write("BENCH_DETAIL " + full + "\n")
    sys.stderr.flush()
    print(compact_line(out), flush=True)


def main():
    global T_START
    T_START = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--detail", default="", help="where the full (uncompacted) result goes; default gpurun_out/bench_detail.json or ./bench_detail.json")
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
    ap.add_argument("--shard", choices=["both", "replicate", "vertex"], default="both",
                    help="N > 1: which multi-GPU mode(s) to run; `value` is the vertex-sharded walk when that leg runs (the replicated figure: replicated_weak_scaling)")
    ap.add_argument("--shard-driver", choices=["both", "cluster", "rccl"], default="both",
                    help="vertex-sharded legs of an N > 1 run: one process driving all devices (peer stores), one process per GPU "
                         "over RCCL (all_to_all_single per super-step), or both (default)")
    ap.add_argument("--biased-leg", type=int, default=1,
                    help="N > 1: also run north_star's biased multi-GPU shape (directed RMAT-26 ef27, p=4 q=.5) vertex-sharded")
    ap.add_argument("--rccl-leg", type=int, default=0, help=argparse.SUPPRESS)      # child job of an N > 1 run: see main()
    ap.add_argument("--nt-loads", type=int, default=-1, help="-1 auto, 0 cached, 1 nontemporal record loads")
    ap.add_argument("--compact", type=int, default=1, help="0: do not use the 16-byte lattice records")
    ap.add_argument("--configs", type=int, default=1, help="1 GPU: also run BASELINE configs C2, C3 (Mode R / A), C5 stand-in")
    ap.add_argument("--configs-scale-cap", type=int, default=0, help="tests: run the `configs` plan with every scale capped at this value")
    ap.add_argument("--time-budget", type=float, default=420.0, help="seconds after which no further optional configuration starts")
    ap.add_argument("--end-to-end", type=int, default=1, help="1 GPU: also time one iteration through srw_walk_and_save")
    ap.add_argument("--pmc", type=int, default=1, help="1 GPU: take roofline.traffic in this run (child rocprofv3 --pmc passes over the headline workload)")
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-plan", choices=["sample", "full"], default="sample", help="sample: ~30 s of CPU work; full: BASELINE.md's whole CPU plan (~100 s)")
    ap.add_argument("--cpu-scale", type=int, default=20)
    ap.add_argument("--cpu-sources", type=int, default=0, help="0 = max(64, 3 per host core)")
    ap.add_argument("--cpu-walk-length", type=int, default=80)
    ap.add_argument("--cluster-leg", type=int, default=0, help=argparse.SUPPRESS)   # child of an N > 1 run: see cluster_leg
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help=argparse.SUPPRESS)     # tests: gloo (chunks staged through host memory)
    ap.add_argument("--share-device", type=int, default=0, help=argparse.SUPPRESS)                     # tests: every rank on device 0 (a one-GPU box)
    args = ap.parse_args()

    import torch
    import _pkg
    pkg = _pkg.load()
    if args.cluster_leg:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X")
        torch.zeros(1, device="cuda")
        print(json.dumps(cluster_leg(pkg, torch, args, args.cluster_leg, args.steps)), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.rccl_leg:
        # one process per GPU, walkers exchanged by ONE RCCL all_to_all_single per super-step (stellar-random-walk_amd/distributed.py)
        import datetime
        import torch.distributed as dist
        dev = 0 if args.share_device else local_rank
        torch.cuda.set_device(dev)
        torch.zeros(1, device="cuda")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend, timeout=datetime.timedelta(minutes=10))
        from importlib import import_module
        sharded = import_module("stellar_random_walk_amd.distributed")

        def bs():
            dist.barrier()
            torch.cuda.synchronize()
        kw = dict(p=args.p, q=args.q, walk_length=args.walk_length, num_walks=1, seed=42)
        vs = sharded.bench_vertex_sharded(dist, dev, rank, world, args.scale, args.edge_factor << args.scale, bool(args.weighted),
                                          bool(args.directed), kw, args.steps, args.warmup, bs)
        if rank == 0:
            vs["exchange_model"] = exchange_model(world, vs.get("value"))
            print(json.dumps(vs), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                         % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    torch.zeros(1, device="cuda")          # torch's device context first (it ships its own HIP runtime)
    dist = None
    if "RANK" in os.environ:  # launched by torch.distributed.run (any world size, including 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        import datetime
        tmo = datetime.timedelta(minutes=30)     # the other ranks wait at a barrier while rank 0 runs the cluster leg
        if args.backend != "nccl":
            dist.init_process_group(backend=args.backend, timeout=tmo)
        else:
            try:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
            except TypeError:  # older signature without device_id
                dist.init_process_group(backend="nccl", timeout=tmo)

    def barrier_sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def allreduce(x, op):
        if dist is None:
            return x
        t = torch.tensor([float(x)], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=op)
        return float(t.item())

    n_edges = args.edge_factor << args.scale
    K, W = args.steps, args.warmup
    walk_kw = dict(p=args.p, q=args.q, walk_length=args.walk_length, num_walks=1, seed=42)
    if args.nt_loads >= 0:
        walk_kw["nt_loads"] = bool(args.nt_loads)
    if args.sampler == "alias":
        walk_kw["sampler"] = "alias"
    if not args.compact:
        walk_kw["compact"] = False

    out = None
    nv = ne = 0
    # ---- replicated mode (N = 1: the single-GPU headline) -----------------------------------------------------------
    if args.shard in ("both", "replicate") or dist is None:
        t0 = time.perf_counter()
        eng = pkg.Engine(device=local_rank)
        eng.generate_rmat(args.scale, n_edges, seed=42, weighted=bool(args.weighted), directed=bool(args.directed))
        nv, ne = eng.stats()
        torch.cuda.synchronize()
        t_graph = time.perf_counter() - t0
        base = rank * (W + K)  # disjoint walk-iteration indices per rank: numWalks = world * K in total
        # the request-rate ceiling of THIS box (about a second; outside the timed region): dependent random 16-byte reads over a
        # table of the size of the headline's record table
        ceiling = None
        if rank == 0:
            try:
                rps, gib = eng.probe_request_rate(ne * 16)
                ceiling = {"reads_per_s": rps, "table_gib": gib}
            except Exception as ex:
                ceiling = None
        t_tables = 0.0
        for it in range(W):
            t_tables += eng.walk(fetch=False, first_walk=base + it, **walk_kw)["setup_ms"] * 1e-3
        if W == 0:                                    # tables must not be built inside the timed region
            t_tables += eng.walk(fetch=False, first_walk=base, **dict(walk_kw, walk_length=1))["setup_ms"] * 1e-3
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
        total_steps = int(allreduce(steps, dist.ReduceOp.SUM)) if dist is not None else steps
        max_dt = allreduce(dt, dist.ReduceOp.MAX) if dist is not None else dt
        if rank == 0:
            avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
            try:
                scan = eng.result_scan_sums()
            except Exception:
                scan = None
            out = {
                "metric": "walk-steps/sec", "value": total_steps / max_dt, "unit": "walk-steps/s", "n_gpus": world,
                "steps": K, "warmup": W, "ms_per_step": max_dt / max(K, 1) * 1e3, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": "RMAT scale-%d ef%d (%d edge lines, %d adjacency entries, %d vertices) %s "
                                       "%s p=%g q=%g walkLength=%d, 1 walk iteration per step, %s"
                                       % (args.scale, args.edge_factor, n_edges, ne, nv,
                                          "directed" if args.directed else "undirected",
                                          "weighted" if args.weighted else "unweighted", args.p, args.q, args.walk_length,
                                          "Mode A (alias + rejection)" if args.sampler == "alias" else "Mode R (reference-exact)"),
                           "walk_steps_per_bench_step": int(steps / max(K, 1)),
                           "parallelism": ("graph replicated, walk iterations sharded x%d, no collective" % world) if world > 1 else "1 GPU",
                           "rng": "Philox4x32-10 keyed (iteration, source, step)"},
                "roofline": roofline_of(stats, steps / max(K, 1), avg_ms, scale=args.scale, n_entries=ne, ceiling=ceiling, scan=scan, q=args.q,
                                        wl=dict(ef=args.edge_factor, p=args.p, q=args.q, weighted=int(args.weighted), directed=int(args.directed), planned_walks=0)),
                "setup_s": {"graph_generate_and_csr": t_graph, "sampling_tables": t_tables,
                            "note": "outside the timed region; one-off per graph / per (p, q)"},
            }
        # ---- the headline kernel's HBM-side counters, taken in this run (child processes; the engine above keeps its graph) -----------
        if rank == 0 and world == 1 and args.pmc and out is not None:
            spec = "%d%s%s" % (args.scale, "w" if args.weighted else "", "d" if args.directed else "")
            pm = pmc_child_pass(out["roofline"]["kernel"].split(" ")[0], [spec, args.p, args.q, args.sampler, 3, args.edge_factor])
            if pm is not None:
                apply_pmc(out["roofline"], pm, avg_ms, steps / max(K, 1), ceiling)
            else:
                out["roofline"]["traffic_note"] = "rocprofv3 --pmc child pass unavailable on this box: traffic not measured in this run"
        # ---- end to end: the reference's contract is path FILES --------------------------------------------------------
        if rank == 0 and world == 1 and args.end_to_end:
            e2e = {"what": "srw_walk_and_save, 1 walk iteration: walk kernel + device-side formatter + PCIe + <output>/path/part-00000 on local "
                           "disk — ONE part file, the reference's default (singleOutput = true: Params.scala:21, Main.scala:64-69; bound by one "
                           "inode's buffered writes); `parts_200`: the same with --singleOutput false (rddPartitions = 200 part files, written in "
                           "parallel); `first_call`: the single-file form with the text path's one-off allocations"}
            tmp_root = os.environ.get("TMPDIR", "/tmp")
            need = nv * (args.walk_length + 2) * 8          # generous bound on the text size
            try:
                free = shutil.disk_usage(tmp_root).free
                if free < need * 1.2:
                    e2e["skipped"] = "needs ~%.0f GB under %s, %.0f GB free" % (need / 1e9, tmp_root, free / 1e9)
                else:
                    # first call: with the one-off allocations of the text path (staging + text slots in HBM, pinned ring); second: without
                    for key, parts in (("first_call", 1), ("", 1), ("parts_200", 200)):
                        d = tempfile.mkdtemp(prefix="srw_bench_", dir=tmp_root)
                        try:
                            t0 = time.perf_counter()
                            st, _ = eng.walk_and_save(os.path.join(d, "out"), n_parts=parts, first_walk=base + W + K, device_format=True, **walk_kw)
                            dt_e = time.perf_counter() - t0
                            nbytes = sum(os.path.getsize(os.path.join(d, "out", "path", f)) for f in os.listdir(os.path.join(d, "out", "path")))
                            r = {"seconds": dt_e, "walk_steps_per_s": st["n_steps"] / dt_e, "text_bytes": nbytes,
                                 "text_GB_per_s": nbytes / dt_e / 1e9, "kernel_ms": st["kernel_ms"], "part_files": parts}
                            if key:
                                e2e[key] = r
                            else:
                                e2e.update(r)
                        finally:
                            shutil.rmtree(d, ignore_errors=True)
            except Exception as ex:  # the headline must survive a full disk
                e2e["error"] = str(ex)[:200]
            out["end_to_end"] = e2e
        eng.close()
        del eng

    # ---- vertex-sharded mode (north_star's split) --------------------------------------------------------------------
    # Default driver: rank 0 alone drives all N devices through srw_cluster_* (one process, peer stores over xGMI, events
    # between the shards' streams) while the other ranks wait — no collective inside the leg, so a failure in it cannot
    # hang the job or cost the headline line.  --shard-driver rccl runs the one-process-per-GPU driver instead
    # (stellar-random-walk_amd/distributed.py: one RCCL all_to_all_single per super-step).
    if dist is not None and args.shard in ("both", "vertex"):
        vs = {}
        if rank == 0:
            # Every leg runs in a CHILD job while the ranks of this one wait at a barrier: a fault or a hang on a peer device
            # (neither driver had met more than one real GPU before the driver's run) costs that leg, not the headline line.
            import socket
            env = {k: v for k, v in os.environ.items()
                   if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_PORT",
                                "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "GROUP_WORLD_SIZE",
                                "ROLE_WORLD_SIZE", "ROLE_NAME", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING")}

            def workload(scale, ef, weighted, directed, p, q, steps):
                return ["--steps", str(steps), "--scale", str(scale), "--edge-factor", str(ef), "--walk-length", str(args.walk_length),
                        "--p", repr(p), "--q", repr(q), "--weighted", str(int(weighted)), "--directed", str(int(directed))]

            def child(cmd, timeout):
                # every leg lives inside --time-budget as well (an N > 1 run must not lose its line to a slow leg): its limit is what is left
                left = args.time_budget - (time.perf_counter() - T_START)
                if left < 45:
                    return {"error": "skipped: %.0f s time budget spent" % args.time_budget}
                timeout = min(timeout, left)
                try:
                    t0 = time.perf_counter()
                    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, text=True)
                    lines = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
                    out_ = json.loads(lines[-1]) if lines else {"error": "exit %d: %s" % (r.returncode, r.stderr[-300:])}
                    out_["leg_wall_s"] = time.perf_counter() - t0
                    return out_
                except Exception as ex:
                    return {"error": str(ex)[:300]}

            me = [sys.executable, os.path.abspath(__file__)]
            test_sw = ["--backend", args.backend, "--share-device", str(args.share_device)]
            head = test_sw + workload(args.scale, args.edge_factor, args.weighted, args.directed, args.p, args.q, K)
            if args.shard_driver in ("both", "cluster"):
                vs["cluster"] = child(me + ["--cluster-leg", str(world)] + head, 360)
            if args.shard_driver in ("both", "rccl"):
                sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
                vs["rccl"] = child([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                                    "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), "--rccl-leg", "1",
                                    "--gpus", str(world), "--warmup", str(min(W, 1))] + head, 480)
            if args.biased_leg and args.shard_driver in ("both", "cluster"):
                # north_star's named biased multi-GPU configuration (C5's stand-in): per-edge tables on the shards
                vs["cluster_c5_shape"] = child(me + ["--cluster-leg", str(world)] + test_sw + workload(26, 27, 0, 1, 4.0, 0.5, 1), 540)
        dist.barrier()
        if rank == 0 and world > 1 and out is not None:
            out = promote_vertex_sharded(out, vs, world)
        if rank == 0:
            first = vs.get("cluster") or vs.get("rccl") or {}
            if out is None:
                out = {"metric": "walk-steps/sec", "value": first.get("value"), "unit": "walk-steps/s", "n_gpus": world, "steps": K,
                       "warmup": W, "ms_per_step": first.get("ms_per_step"), "higher_is_better": True, "scaling": "strong",
                       "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                       "config": {"workload": first.get("workload"), "parallelism": first.get("parallelism")}}
            out["vertex_sharded"] = vs

    if rank == 0:
        if world == 1 and args.configs:
            cfgs = []
            plan = [("C2", 20, 16, False, False, 1.0, 1.0, "reference", 10, 1),
                    ("C3 Mode R", 24, 16, True, False, 0.25, 4.0, "reference", 2, 1),
                    ("C3 Mode R numWalks=100", 24, 16, True, False, 0.25, 4.0, "reference", 2, 1),
                    ("C3 Mode A", 24, 16, True, False, 0.25, 4.0, "alias", 3, 1),
                    ("C3 graph q=1 Mode R", 24, 16, True, False, 0.25, 1.0, "reference", 3, 1),
                    ("C5 stand-in Mode R", 26, 27, False, True, 4.0, 0.5, "reference", 1, 1),
                    ("C5 stand-in Mode A", 26, 27, False, True, 4.0, 0.5, "alias", 2, 1)]
            cap = args.configs_scale_cap or 99
            # every optional leg is timed (`leg_wall_s`, detail file) and starts only while the run is inside --time-budget: on a box whose
            # allocations stall (profiles/r05_table_build.md 5) the line loses its last rows instead of the run losing its line
            def leg(name, fn):
                if time.perf_counter() - T_START > args.time_budget:
                    cfgs.append({"name": name, "error": "skipped: %.0f s time budget spent" % args.time_budget})
                    return
                t0 = time.perf_counter()
                try:
                    r = fn()
                except Exception as ex:
                    r = {"name": name, "error": str(ex)[:300]}
                r["leg_wall_s"] = time.perf_counter() - t0
                cfgs.append(r)

            for (name, sc, ef, wt, dr, p, q, smp, k, w) in plan:
                sc = min(sc, cap)
                leg(name, lambda: run_config(pkg, local_rank, name, sc, ef, wt, dr, p, q, smp, k, w, ceiling=ceiling,
                                             plan_walks=100 if "numWalks=100" in name else 0, pmc=bool(args.pmc) and name == "C3 Mode R"))
            if args.shard in ("both", "vertex"):
                leg("sharded w1 p=q=1", lambda: run_sharded_world1(pkg, local_rank, scale=min(24, cap)))
                rep = {c.get("name"): c.get("value") for c in cfgs}
                for (name, sc, ef, wt, dr, p, q, of) in [
                        ("sharded w1 C3", 24, 16, True, False, 0.25, 4.0, "C3 Mode R"),
                        ("sharded w1 C3 graph q=1", 24, 16, True, False, 0.25, 1.0, "C3 graph q=1 Mode R"),
                        ("sharded w1 C5 stand-in", 26, 27, False, True, 4.0, 0.5, "C5 stand-in Mode R")]:
                    sc = min(sc, cap)
                    leg(name, lambda: run_sharded_biased_world1(pkg, local_rank, name, sc, ef, wt, dr, p, q, rep.get(of), batch2=(sc <= 24)))
            out["configs"] = cfgs
            if time.perf_counter() - T_START <= args.time_budget:
                try:
                    out["embedding_stage"] = run_embedding_stage(pkg, local_rank, min(20, cap))
                except Exception as ex:
                    out["embedding_stage"] = {"error": str(ex)[:200]}
            out["wall_s_before_cpu_baseline"] = time.perf_counter() - T_START
        if world == 1 and args.cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        emit(out, args)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
