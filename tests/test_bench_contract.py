"""bench.py's output contract, on a small graph: one JSON line with the keys the driver reads (metric, value, unit, n_gpus,
steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload), the `roofline` and
`cpu_baseline` objects, and — launched through torch.distributed.run with one rank — the `vertex_sharded` legs that an N > 1
run adds (child jobs started by rank 0: the in-process cluster and the one-process-per-GPU RCCL driver)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"]


def last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scale", "16", "--steps", "3", "--warmup", "1", "--configs", "0",
                        "--end-to-end", "1", "--cpu-baseline", "1", "--cpu-scale", "10", "--cpu-sources", "64", "--cpu-walk-length", "10"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    for k in KEYS:
        assert k in d, k
    assert d["metric"] == "walk-steps/sec" and d["unit"] == "walk-steps/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - d["config"]["walk_steps_per_bench_step"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel"] == "k_walk_first_order"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["kernel_ms_avg"] > 0
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms_avg"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert rf["traffic"] is None                      # no PMC file for this graph: never a number that was not measured
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "walk-steps/s" and cb["sample"]
    # BASELINE.md §3's plan: karate and the RMATs, (1, 1) and (.25, 4), faithful and fast
    plan = cb["plan"]
    assert {(e["p"], e["q"]) for e in plan} == {(1.0, 1.0), (0.25, 4.0)} and all(e["value"] > 0 and e["kind"] == "port" for e in plan)
    assert any("karate" in e["workload"] for e in plan) and any(e["variant"].startswith("fast") for e in plan)
    e2e = d["end_to_end"]
    assert e2e.get("text_bytes", 0) > 0 and e2e["walk_steps_per_s"] > 0, e2e


def test_torchrun_one_rank_adds_the_vertex_sharded_leg():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--scale", "16", "--steps", "2", "--warmup", "1",
                        "--configs", "0", "--end-to-end", "0", "--cpu-baseline", "0", "--biased-leg", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    for k in KEYS:
        assert k in d, k
    # both exchange drivers report: one process driving all devices (peer stores) and one process per GPU (RCCL)
    for leg in ("cluster", "rccl"):
        vs = d["vertex_sharded"][leg]
        assert "error" not in vs, (leg, vs)
        assert vs["value"] > 0 and vs["scaling"] == "strong" and "sharded by source vertex" in vs["parallelism"], (leg, vs)
    assert d["scaling"] == "weak" and d["n_gpus"] == 1
