"""The bench line contract (task statement, section 4).  ``check_bench_line`` is applied (a) here, without a GPU, to the
line committed under profiles/ together with the rocprofv3 / PMC summaries committed beside it -- a consistency check of
the committed artefacts, not a measurement -- and (b) in tests/test_bench_gpu.py to the line a live ``bench.py`` run prints
on the MI355X."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check_bench_line(d, n_gpus=None):
    for key, typ in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int),
                     ("warmup", int), ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str),
                     ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None                      # BASELINE.md holds no published number for this metric
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["n_gpus"] * d["config"]["per_gpu_batch"] * 1e3 / d["ms_per_step"]) < 0.01 * d["value"]
    if n_gpus is not None:
        assert d["n_gpus"] == n_gpus and d["config"]["parallelism"] == f"dp{n_gpus}"
    if "roofline" in d:
        r = d["roofline"]
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert r["traffic"] is None or r["traffic"] > 0
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def latest(pattern):
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    assert hits, pattern
    return hits[-1]


def test_committed_bench_line_agrees_with_the_committed_profiles():
    with open(latest("r*_bench_line.json")) as fh:
        d = json.loads(fh.read().strip().splitlines()[-1])
    check_bench_line(d)
    r = d["roofline"]
    # the dominant kernel is named as rocprofv3 lists it, and the committed stats hold it with a matching duration
    import csv
    tag = os.path.basename(latest("r*_bench_line.json"))[:3]
    # (the pass with every kernel alone on the chip is the one bench.py's HIP events correspond to, when a round committed one)
    alone = os.path.join(ROOT, "profiles", tag + "_bench_kernel_stats_alone.csv")
    with open(alone if os.path.exists(alone) else latest(tag + "_bench_kernel_stats*.csv")) as fh:
        rows = {row["Name"]: row for row in csv.DictReader(fh)}
    sym = [n for n in rows if r["kernel"] in n]
    assert sym, r["kernel"]
    avg_us = float(rows[sym[0]]["AverageNs"]) / 1e3
    assert abs(avg_us - r["avg_launch_us"]) / avg_us < 0.10, (avg_us, r["avg_launch_us"])
    with open(latest(tag + "_pmc_traffic.json")) as fh:
        pmc = json.load(fh)
    assert r["kernel"] in pmc["kernels"] and 0.95 < pmc["calibration"]["ratio"] < 1.05


def test_bench_cli_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
