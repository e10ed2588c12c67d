"""HBM traffic per kernel launch from rocprofv3 PMC passes (run on the GPU box).

  python tools/pmc_traffic.py collect [config] [tag]  # two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short bench.py run
  python tools/pmc_traffic.py parse   [config] [tag]  # -> gpurun_out/pmc_<tag>/<tag>_pmc_hbm_bytes.json  (copy it to profiles/)
  config: c2 (default) | c3 | c4 | c5 (bench.py --config) | hyper (tools/prof_hyper.py: the hyper-on iteration); tag: default r06_<config>

Units / corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): the counters are in KB and
FETCH_SIZE reports half of the bytes actually fetched on gfx950, so bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  The
calibration block re-checks both facts on this machine with a torch fill (pure write) and a rocBLAS gemv (pure read) of
a 256 MB array.
"""
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = sys.argv[2] if len(sys.argv) > 2 else "c2"
TAG = sys.argv[3] if len(sys.argv) > 3 else f"r06_{CFG}"
OUT = os.path.join(ROOT, "gpurun_out", f"pmc_{TAG}")
CMD = ["python", os.path.join(ROOT, "bench.py"), "--config", CFG, "--no-cpu-baseline", "--no-elbo-tol", "--steps",
       "10" if CFG == "c5" else "40", "--warmup", "4" if CFG == "c5" else "10"]
if CFG == "hyper":  # the reference's default mode: hyper-parameter step every iteration (tools/prof_hyper.py, C2 shape)
    CMD = ["python", os.path.join(ROOT, "tools", "prof_hyper.py")]
CAL = ["python", "-c", "import torch; x=torch.empty(1000000,32,dtype=torch.float64,device='cuda'); x.fill_(1.0); "
       "v=torch.ones(32,dtype=torch.float64,device='cuda'); y=x@v; torch.cuda.synchronize()"]


def collect():
    env = dict(os.environ, TMPDIR="/tmp")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for tag, cmd in (("bench", CMD), ("cal", CAL)):
            d = os.path.join(OUT, f"{tag}_{ctr}")
            os.makedirs(d, exist_ok=True)
            subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"]
                           + cmd, check=True, env=env, cwd="/tmp", stdout=subprocess.DEVNULL)


def _read(tag, ctr):
    acc = {}  # name -> [launches, sum, max]
    for f in glob.glob(os.path.join(OUT, f"{tag}_{ctr}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            name = re.sub(r"^void agp::", "", r["Kernel_Name"])
            name = re.sub(r"\(.*$", "", name)
            a = acc.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] = max(a[2], float(r["Counter_Value"]))
    return acc


def parse():
    res = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --kernel-trace -- " + " ".join(CMD[:1] + [os.path.relpath(CMD[1], ROOT)] + CMD[2:]),
           "units": "counter values are KB; corrected bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024", "kernels": {}}
    f, w = _read("bench", "FETCH_SIZE"), _read("bench", "WRITE_SIZE")
    for k in f:
        if k not in w or not k.startswith("k_"):
            continue
        fa, wa = f[k][1] / f[k][0], w[k][1] / w[k][0]
        res["kernels"][k] = {"FETCH_SIZE": {"launches": f[k][0], "avg_counter_KB": round(fa, 1)},
                             "WRITE_SIZE": {"launches": w[k][0], "avg_counter_KB": round(wa, 1)},
                             "hbm_bytes_per_launch_corrected": int((2 * fa + wa) * 1024)}
    cf, cw = _read("cal", "FETCH_SIZE"), _read("cal", "WRITE_SIZE")
    res["calibration"] = {"note": "fill of 1e6x32 f64 (256.0 MB written) and gemv over it (256.0 MB read)",
                          "WRITE_SIZE_KB_largest_launch": {k: round(v[2], 1) for k, v in cw.items() if v[2] > 1e5},
                          "FETCH_SIZE_KB_largest_launch": {k: round(v[2], 1) for k, v in cf.items() if v[2] > 5e4}}
    with open(os.path.join(OUT, f"{TAG}_pmc_hbm_bytes.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: v["hbm_bytes_per_launch_corrected"] for k, v in res["kernels"].items()}, indent=1))
    print(json.dumps(res["calibration"], indent=1))


def mfma():
    """MFMA utilisation per kernel: one more --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES; the counter sums busy cycles over all SIMDs,
    so it is put in proportion to the kernel's duration and to the same ratio of k_mfma_peak (back-to-back MFMAs on every SIMD of
    the chip, part of every bench.py run) instead of to an assumed clock: util = (busy / us) / (busy / us of k_mfma_peak)."""
    env = dict(os.environ, TMPDIR="/tmp")
    ctr = "SQ_VALU_MFMA_BUSY_CYCLES"
    d = os.path.join(OUT, f"bench_{ctr}")
    os.makedirs(d, exist_ok=True)
    subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"]
                   + [c for c in CMD if c not in ("--no-extras",)], check=True, env=env, cwd="/tmp", stdout=subprocess.DEVNULL)
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            name = re.sub(r"\(.*$", "", re.sub(r"^void agp::", "", r["Kernel_Name"]))
            a = acc.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    peak = [v for k, v in acc.items() if k.startswith("k_mfma_peak")]
    ref = max(v[1] / v[2] for v in peak) if peak else None
    res = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -- python bench.py --config " + CFG + " ...",
           "reference": "k_mfma_peak (back-to-back v_mfma 16x16x4 on every SIMD): busy cycles per microsecond = " + (f"{ref:.1f}" if ref else "n/a"),
           "kernels": {}}
    for k, (n, busy, us) in sorted(acc.items(), key=lambda kv: -kv[1][2]):
        if not k.startswith("k_") or us <= 0:
            continue
        res["kernels"][k] = {"launches": n, "avg_us": round(us / n, 2), "mfma_busy_cycles_per_launch": round(busy / n, 1),
                             "mfma_util_vs_k_mfma_peak": round(busy / us / ref, 4) if ref else None}
    with open(os.path.join(OUT, f"{TAG}_pmc_mfma_util.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: (v["avg_us"], v["mfma_util_vs_k_mfma_peak"]) for k, v in list(res["kernels"].items())[:12]}, indent=1))


if __name__ == "__main__":
    {"collect": collect, "parse": parse, "mfma": mfma}[sys.argv[1]]()
