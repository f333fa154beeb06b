"""-m gpu: bench.py end to end on a tiny configuration -- the JSON contract the driver parses."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_contract(built):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tris", "20000", "--width", "320", "--height", "184",
                          "--spp", "2", "--depth", "8", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.5", "--tex-size", "32"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["unit"] == "Mray/s" and r["n_gpus"] == 1 and r["steps"] == 2 and r["vs_baseline"] is None and r["dtype"] == "f32"
    assert r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is None                       # PMC traffic is only quoted for the configuration it was measured on
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert r["parity"]["culled_equals_reference_traversal"] is True and r["parity"]["oracle_bit_exact_on_sample"] is True
    assert r["value"] > 0 and r["ms_per_step"] > 0
