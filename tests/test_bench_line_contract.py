"""The committed bench lines (profiles/r06_c{2..5}_bench_line.json: what `python bench.py [--config cN]` printed on the MI355X for the
final build of the round) honour the driver's JSON contract: one object, the metric BASELINE.json names, whole-job value, the
`roofline` and `cpu_baseline` objects with their fields, `task_graph_fallbacks` (round 6) zero.  CPU test: reads files only."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cfg", ["c2", "c3", "c4", "c5"])
def test_committed_bench_line(cfg):
    path = os.path.join(ROOT, "profiles", f"r06_{cfg}_bench_line.json")
    text = open(path).read().strip()
    assert len(text.splitlines()) == 1, "one JSON line"
    d = json.loads(text)
    assert d["metric"] == "cavi_iters_per_sec" and d["unit"] == "iter/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["dtype"] in ("f64", "f32") and d["scaling"] in ("weak", "strong")
    assert set(d["config"]) >= {"workload"} and "model" not in d["config"]
    assert d["value"] == pytest.approx(d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3), rel=2e-3)
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=2e-3)
    assert r["peak"] in (78.6, 157.3) and isinstance(r["traffic"], int) and r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "iter/s" and c["sample"]
    assert d["task_graph_fallbacks"] == 0
    if cfg == "c2":  # the headline configuration carries the second half of the metric and the round's new objects
        assert d["time_to_elbo_tol_reachable"]["seconds"] < 0.72 and d["time_to_elbo_tol_reachable"]["iters"] == 1900
        p = d["predict_roofline"]
        assert p["seconds_per_pass"] < 2.2e-3 and p["valu"]["instructions_per_value"] == 22 and 30 < p["valu"]["instructions_per_value_total"] < 40
        assert 0 < p["hbm"]["frac"] < 0.05 and 0.3 < p["mfma"]["frac"] < 0.6
