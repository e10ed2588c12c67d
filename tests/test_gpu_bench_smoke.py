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


@pytest.mark.parametrize("config,extra", [("c2", []), ("c4", ["--m", "256", "--batch", "256", "--N", "20000"])])
def test_bench_starts_its_own_ranks(built, config, extra):
    """`python bench.py --gpus 2` -- the shape of the driver's command, no launcher around it -- starts two ranks itself
    (one process per GPU, rendezvous on 127.0.0.1) and rank 0 prints the ONE line.  AGP_BENCH_SHARE_GPU=1 puts both ranks on GPU 0 (gloo + the
    callback transport): the N > 1 code path of bench.py through agp_svgp_cavi_step_multi, batch-parallel (c2: the packed
    statistic of analyticVI.jl:168,179 all-reduced, with the split-overlap A/B) and latent-parallel (c4)."""
    env = dict(os.environ, AGP_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--config", config,
                        "--no-cpu-baseline", "--no-elbo-tol", "--no-extras"] + extra, capture_output=True, text=True, timeout=1500,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 only
    d = json.loads(lines[0])
    assert d["metric"] == "cavi_iters_per_sec" and d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2
    assert d["value"] > 0 and d["ms_per_step"] > 0
    c = d["collective"]
    assert c["ranks_seen"] == 2 and c["issuer"] and c["us_per_call"] >= 0
    if config == "c2":
        assert d["scaling"] == "weak" and d["config"]["global_batch"] == 2048
        mp = 1024
        assert c["bytes_per_step"] == 8 * (mp + (mp // 64) * (mp // 64 + 1) // 2 * 4096) and c["calls_per_step"] == 1
        assert c["split_overlap_ab"]["AGP_SPLIT_OVERLAP"] == 1 and c["split_overlap_ab"]["ms_per_step"] > 0
    else:
        assert d["scaling"] == "strong" and c["calls_per_step"] == 2  # sum_k gamma_k twice per step (logisticsoftmax.jl:65-72)


def test_bench_under_the_drivers_launcher_command(built):
    """The driver's N > 1 form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port
    P bench.py --gpus N --steps K --warmup W` (both ranks on GPU 0 here: AGP_BENCH_SHARE_GPU=1)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, AGP_BENCH_SHARE_GPU="1", AGP_BENCH_NO_OVERLAP_AB="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["collective"]["ranks_seen"] == 2
