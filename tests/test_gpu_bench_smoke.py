"""bench.py with the SHORT step counts a driver may pass (the index stream it pre-generates has steps + warmup entries; every loop
behind the timed region -- hyper-on iteration, hyper_roofline, isolated launches -- has to live within it), in a fresh process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("steps,warmup", [(20, 5), (3, 1)])
def test_bench_line_with_short_runs(built, steps, warmup):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup),
                        "--cpu-seconds", "1", "--cpu-elbo-seconds", "0", "--no-elbo-tol"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["metric"] == "cavi_iters_per_sec" and d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["value"] > 0 and d["higher_is_better"] is True and d["dtype"] == "f64"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"]
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"
    assert d["ms_per_step_with_hyper_update"] > 0 and d["hyper_roofline"]["frac"] > 0
