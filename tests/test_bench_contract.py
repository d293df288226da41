"""bench.py's contract with the driver: ONE JSON line on stdout carrying the keys the driver parses (metric, value, unit, n_gpus, steps, warmup, ms_per_step,
higher_is_better, scaling, vs_baseline, dtype, data, config.workload) plus `roofline` {bound, achieved, peak, unit, frac, traffic} and `cpu_baseline`
{value, unit, cores, kind, sample}; and the provenance helpers that tie a line to the committed profiles (source_hash, workload tags, traffic lookup)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_workload_tags_select_the_committed_profile_sets():
    assert bench.workload_tag("a1c2", 8192, 3, 128) == "" and bench.workload_tag("a1c2", 8192, 8, 128) == "L8"
    assert bench.workload_tag("mck4", 8192, 8, 128) == "mck4" and bench.workload_tag("solo", 65536, 8, 128) == "solo"
    assert bench.workload_tag("synth32", 1024, 6, 512) == "synth32"
    assert bench.workload_tag("a1c2", 4096, 3, 128) is None and bench.workload_tag("mck4", 8192, 3, 128) is None      # other sizes: no committed profile


def test_committed_traffic_names_its_source_and_whether_the_sources_match():
    h = bench.source_hash()
    assert len(h) == 12 and h == bench.source_hash()
    for tag, dtype, kernel in (("", "bf16", "stack_step"), ("", "x3", "stack_step"), ("L8", "bf16", "gradw"), ("mck4", "bf16", "stack_step"), ("synth32", "bf16", "gradw")):
        got = bench.committed_traffic(tag, dtype, kernel)
        assert got is not None, (tag, dtype, kernel)
        nbytes, src = got
        assert nbytes > 1e7 and src.startswith("r0") and "pmc_traffic.json" in src and "source_hash" in src
        assert (("same sources" in src) or ("OTHER sources" in src))
        assert (f"_{tag}_" in src) if tag else not any(f"_{t}_" in src for t in bench.WORKLOAD_TAGS)
    assert bench.committed_traffic(None, "bf16", "stack_step") is None and bench.committed_traffic("", "f32", "stack_step") is None
    assert bench.committed_traffic("", "bf16", "no_such_kernel") is None


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "512", "--no-extras", "--cpu-batch", "32",
                        "--min-time", "0.01"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[:500]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "source_hash", "flat_gradient_bytes"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None      # (no committed profile of a 512-window batch)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "windows/s" and "sample" in c
    assert d["source_hash"] == bench.source_hash()
