"""The bench line's contract (task statement, Measurement section), checked on the committed line of the default run
(profiles/r03_bench_c3.log: `python bench.py` on an MI355X) -- so that a change to bench.py that drops or renames a field
shows up on the CPU -- and on bench.py's argument parser (defaults: N = 1, a run of seconds)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r03_bench_c3.log")) as f:
        last = [l for l in f.read().splitlines() if l.startswith("{")][-1]
    return json.loads(last)


def test_committed_bench_line_has_every_contract_field():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f32" and d["unit"] == "queries/s" and d["scaling"] in ("strong", "weak")
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] < 1.0
    # value = queries of the batch / step time; achieved = algorithmic bytes / the scoring kernels' time (tier 1 + tier 2)
    assert abs(d["value"] - d["config"]["n_queries"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    t_score = (r["kernel_ms"] + r["tier2_kernel_ms"]) * 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / t_score / 1e9) / r["achieved"] < 1e-3
    assert t_score < d["ms_per_step"] * 1e-3  # the kernels fit inside the step the driver's clock sees
    assert r["traffic"] is None or r["traffic"] > 0.9 * r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0


def test_bench_defaults_are_one_gpu_and_a_short_run():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    src = open(spec.origin).read()
    assert 'add_argument("--gpus", type=int, default=1)' in src
    assert 'add_argument("--steps", type=int, default=20)' in src and 'add_argument("--warmup", type=int, default=3)' in src
    assert "oracle" in src and "reference" not in [l.strip() for l in src.splitlines() if l.strip().startswith("import ")]
