"""The bench line contract (task statement, section 4), checked on the line committed under profiles/ -- the last one
measured on an MI355X -- and on bench.py's command line.  No GPU needed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r01_bench_line.json")) as fh:
        d = json.loads(fh.read().strip().splitlines()[-1])
    for key, typ in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int),
                     ("warmup", int), ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str),
                     ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None                      # BASELINE.md holds no published number for this metric
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["n_gpus"] * d["config"]["per_gpu_batch"] * 1e3 / d["ms_per_step"]) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # the dominant kernel is named as rocprofv3 lists it, and the committed stats hold it with a matching duration
    import csv
    with open(os.path.join(ROOT, "profiles", "r01_bench_kernel_stats_final.csv")) as fh:
        rows = {row["Name"]: row for row in csv.DictReader(fh)}
    sym = [n for n in rows if r["kernel"] in n]
    assert sym, r["kernel"]
    avg_us = float(rows[sym[0]]["AverageNs"]) / 1e3
    assert abs(avg_us - r["avg_launch_us"]) / avg_us < 0.15, (avg_us, r["avg_launch_us"])
    with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
        pmc = json.load(fh)
    assert r["kernel"] in pmc["kernels"] and 0.95 < pmc["calibration"]["ratio"] < 1.05


def test_bench_cli_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
