"""bench.py's output contract, on a small graph: ONE stdout line of at most 4 KB (the full record goes to --detail and stderr) with the keys the driver reads (metric, value, unit, n_gpus,
steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload), the `roofline` and
`cpu_baseline` objects, and — launched through torch.distributed.run with one rank — the `vertex_sharded` legs that an N > 1
run adds (child jobs started by rank 0: the in-process cluster and the one-process-per-GPU RCCL driver)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"]


LINE_LIMIT = 4096        # round 4's 24 KB line was not parsed by the driver: the last stdout line stays small, the detail goes to a file


def last_json(out):
    lines = [l for l in out.splitlines() if l.strip()]
    # the LAST stdout line is the JSON object, and the only one (RCCL prints its version banner on stdout before it)
    assert lines and lines[-1].startswith("{") and sum(l.startswith("{") for l in lines) == 1, out[-2000:]
    assert len(lines[-1]) <= LINE_LIMIT, len(lines[-1])
    return json.loads(lines[-1])


def check_driver_keys(d):
    for k in KEYS:
        assert k in d, k
    assert d["metric"] == "walk-steps/sec" and d["unit"] == "walk-steps/s" and d["n_gpus"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["dtype"] == "f64"
    rf = d["roofline"]
    assert all(isinstance(v, (int, float, str)) or v is None for v in rf.values()), rf           # numbers and names, no prose formulas
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel"] == "k_walk_first_order"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5 * rf["frac"] and rf["kernel_ms_avg"] > 0
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms_avg"] * 1e-3) / 1e9) < 1e-4 * rf["achieved"]
    # traffic: measured IN this run (child rocprofv3 --pmc passes over the same workload, stamped with the library's commit) or absent —
    # never a counter replayed from another commit
    if rf["traffic"] is not None:
        assert rf["traffic"] > 0 and rf["traffic_commit"] == d["library"].split()[-1] and rf["physical_traffic_frac"] > 0, rf


@pytest.mark.gpu
def test_single_gpu_line(tmp_path):
    detail = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scale", "16", "--steps", "3", "--warmup", "1", "--configs", "0",
                        "--end-to-end", "1", "--cpu-baseline", "1", "--cpu-scale", "10", "--cpu-sources", "64", "--cpu-walk-length", "10",
                        "--detail", detail],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    check_driver_keys(d)
    assert d["steps"] == 3 and d["warmup"] == 1
    assert d["value"] > 0 and abs(d["value"] - d["config"]["walk_steps_per_bench_step"] / (d["ms_per_step"] * 1e-3)) < 1e-4 * d["value"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "walk-steps/s" and cb["sample"]
    assert "plan" not in cb
    e2e = d["end_to_end"]
    assert e2e["walk_steps_per_s"] > 0, e2e
    # the full record: same headline, plus BASELINE.md §3's CPU plan (karate and the RMATs, (1, 1) and (.25, 4), faithful and fast)
    full = json.load(open(detail))
    assert abs(full["value"] - d["value"]) < 1e-5 * d["value"] and d["detail_file"] == detail
    plan = full["cpu_baseline"]["plan"]
    assert {(e["p"], e["q"]) for e in plan} == {(1.0, 1.0), (0.25, 4.0)} and all(e["value"] > 0 and e["kind"] == "port" for e in plan)
    assert any("karate" in e["workload"] for e in plan) and any(e["variant"].startswith("fast") for e in plan)
    assert full["end_to_end"].get("text_bytes", 0) > 0
    assert "BENCH_DETAIL {" in r.stderr


@pytest.mark.gpu
def test_single_gpu_line_with_every_config_stays_small(tmp_path):
    """The form the driver meets (--configs 1: ten more configurations, the CPU plan) on graphs capped at scale 13: the one
    stdout line must still fit, and every configuration must be in its summary."""
    detail = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scale", "14", "--steps", "2", "--warmup", "1", "--configs", "1",
                        "--configs-scale-cap", "13", "--end-to-end", "1", "--cpu-baseline", "1", "--cpu-scale", "10", "--cpu-sources", "64",
                        "--cpu-walk-length", "10", "--detail", detail],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    check_driver_keys(d)
    rows = d["configs_summary"]
    full = json.load(open(detail))
    assert [c["name"] for c in rows] == [c["name"] for c in full["configs"]] and len(rows) == 11
    for c in rows:
        assert "error" not in c and c["value"] > 0 and c["ms_per_step"] > 0, c
    assert all("fraction_of_replicated" in c for c in rows if c["name"].startswith("sharded"))
    assert all("job_numWalks10_steps_per_s" in c for c in rows if c["name"] in ("C3 Mode R", "C5 stand-in Mode R"))
    assert "cpu_baseline" in d and "plan" not in d["cpu_baseline"]


def test_compact_line_sheds_before_it_overflows():
    """compact_line never hands the driver more than LINE_LIMIT bytes, whatever the detail holds."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    out = {"metric": "walk-steps/sec", "value": 1.0, "unit": "walk-steps/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": "w"},
           "roofline": {"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 1.0 / 8000, "traffic": None,
                        "algorithmic_bytes_formula": "x" * 5000},
           "cpu_baseline": {"value": 1.0, "unit": "walk-steps/s", "cores": 1, "kind": "port", "sample": "s", "plan": [{"x": "y" * 100}] * 100},
           "configs": [{"name": "n" * 200, "value": 1.0, "ms_per_step": 1.0, "roofline": {"frac": 0.1}} for _ in range(60)]}
    line = b.compact_line(out)
    d = json.loads(line)
    assert len(line) <= b.LINE_LIMIT == LINE_LIMIT and "roofline" in d and "cpu_baseline" in d and "algorithmic_bytes_formula" not in d["roofline"]


@pytest.mark.gpu
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
    full = json.loads([l for l in r.stderr.splitlines() if l.startswith("BENCH_DETAIL ")][-1][len("BENCH_DETAIL "):])
    # both exchange drivers report: one process driving all devices (peer stores) and one process per GPU (RCCL)
    for leg in ("cluster", "rccl"):
        vs = full["vertex_sharded"][leg]
        assert "error" not in vs, (leg, vs)
        assert vs["value"] > 0 and vs["scaling"] == "strong" and "sharded by source vertex" in vs["parallelism"], (leg, vs)
        assert d["vertex_sharded"][leg]["value"] > 0
    assert d["scaling"] == "weak" and d["n_gpus"] == 1


def test_promote_vertex_sharded_makes_the_sharded_walk_the_value():
    """N > 1: the line's value is the vertex-sharded RCCL leg; the replicated figure is an extra key; a run whose
    sharded legs all failed keeps the replicated value and says so."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    rep = {"metric": "walk-steps/sec", "value": 8.0e10, "unit": "walk-steps/s", "n_gpus": 8, "steps": 10, "warmup": 2, "ms_per_step": 60.0,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "w", "parallelism": "graph replicated, walk iterations sharded x8, no collective", "rng": "philox"},
           "roofline": {"bound": "hbm", "achieved": 800.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": None, "kernel": "k_walk_first_order"}}
    legs = {"cluster": {"value": 3.0e10, "ms_per_step": 160.0, "scaling": "strong", "steps": 10, "warmup": 10, "workload": "ws", "parallelism": "px"},
            "rccl": {"value": 2.5e10, "ms_per_step": 190.0, "scaling": "strong", "steps": 10, "warmup": 2, "workload": "ws", "parallelism": "py",
                     "per_superstep_unoverlapped": {"kernels_ms": 1.5, "exchange_ms": 0.4, "super_steps": 81}, "exchange_model": {"x": 1}}}
    top = b.promote_vertex_sharded(dict(rep), legs, 8)
    assert top["value"] == 2.5e10 and top["scaling"] == "strong" and top["n_gpus"] == 8 and top["ms_per_step"] == 190.0
    assert "source-vertex shards x8, one all_to_all per super-step" in top["config"]["parallelism"] and top["config"]["driver"] == "rccl"
    assert top["replicated_weak_scaling"]["value"] == 8.0e10 and top["replicated_weak_scaling"]["scaling"] == "weak"
    assert top["per_superstep_unoverlapped"]["exchange_ms"] == 0.4 and top["roofline"]["kernel"] == "k_walk_first_order"
    for k in KEYS:
        assert k in top, k
    line = json.loads(b.compact_line(dict(top, vertex_sharded=legs)))
    assert line["value"] == 2.5e10 and line["replicated_weak_scaling"]["value"] == 8.0e10 and line["per_superstep_unoverlapped"]["super_steps"] == 81
    # the RCCL leg failed: the cluster driver's leg is the value; both failed: the replicated value stays, with a note
    top = b.promote_vertex_sharded(dict(rep), {"rccl": {"error": "x"}, "cluster": legs["cluster"]}, 8)
    assert top["value"] == 3.0e10 and top["config"]["driver"] == "cluster"
    top = b.promote_vertex_sharded(dict(rep), {"rccl": {"error": "x"}}, 8)
    assert top["value"] == 8.0e10 and top["scaling"] == "weak" and "vertex_sharded_note" in top


@pytest.mark.gpu
def test_two_ranks_line_is_the_vertex_sharded_walk():
    """The N > 1 form end to end on a one-GPU box: two ranks over gloo, both on device 0 (--backend gloo --share-device 1: the chunks are staged
    through host memory, everything else is the path an 8-GPU run takes): the line's value is the RCCL driver's vertex-sharded leg."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29549", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "1",
                        "--scale", "15", "--steps", "2", "--warmup", "1", "--configs", "0", "--end-to-end", "0", "--cpu-baseline", "0",
                        "--biased-leg", "0", "--pmc", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "source-vertex shards x2, one all_to_all per super-step" in d["config"]["parallelism"]
    assert d["config"]["driver"] == "rccl" and d["value"] == d["vertex_sharded"]["rccl"]["value"] > 0
    assert d["replicated_weak_scaling"]["value"] > 0 and d["replicated_weak_scaling"]["scaling"] == "weak"
    assert d["per_superstep_unoverlapped"]["super_steps"] > 0
    full = json.loads([l for l in r.stderr.splitlines() if l.startswith("BENCH_DETAIL ")][-1][len("BENCH_DETAIL "):])
    assert full["vertex_sharded"]["cluster"]["value"] > 0 and "exchange_model" in full
