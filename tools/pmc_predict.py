"""Counters of the streaming predictor (round 6): instructions per kernel value, MFMA-busy share, fabric bytes.

    python tools/pmc_predict.py          # on the GPU box -> gpurun_out/r06_predict_pmc.json (copy to profiles/)

One rocprofv3 --pmc pass per counter over tools/prof_predict.py (five passes of predict_f means over N = 1e6 at the C2 shape);
SQ_INSTS_* count wave instructions summed over all waves, so instructions per kernel value = counter x 64 lanes / (N x m)."""
import csv
import glob
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "pmc_r06_predict")
CMD = ["python", os.path.join(ROOT, "tools", "prof_predict.py")]
COUNTERS = ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVES",
            "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "FETCH_SIZE", "WRITE_SIZE"]
N, M = 1000000, 1024


def main():
    env = dict(os.environ, TMPDIR="/tmp")
    res = {"command": "rocprofv3 --pmc <counter> --kernel-trace -- python tools/prof_predict.py (one pass per counter)",
           "kernel": "k_kernelmatrix_mma<double, 0, 1>", "N": N, "m": M, "counters": {}, "unavailable": []}
    for ctr in COUNTERS:
        d = os.path.join(OUT, ctr)
        os.makedirs(d, exist_ok=True)
        r = subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + CMD,
                           env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        vals = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == ctr and re.search(r"k_kernelmatrix_mma<double, 0, 1>", row["Kernel_Name"]):
                    vals.append(float(row["Counter_Value"]))
        if r.returncode != 0 or not vals:
            res["unavailable"].append(ctr)
            continue
        res["counters"][ctr] = {"launches": len(vals), "per_launch": sum(vals) / len(vals)}
    c = res["counters"]
    der = {}
    nval = float(N) * M
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"):
        if k in c:
            der[k.lower() + "_per_kernel_value_lanes"] = round(c[k]["per_launch"] * 64.0 / nval, 3)
    if "SQ_INSTS_MFMA" in c:
        der["mfma_flops_per_kernel_value"] = round(c["SQ_INSTS_MFMA"]["per_launch"] * 2048.0 / nval, 2)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        der["fabric_bytes_per_launch_corrected"] = int((2 * c["FETCH_SIZE"]["per_launch"] + c["WRITE_SIZE"]["per_launch"]) * 1024)
    if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c:  # (SQ_INSTS_VALU counts the MFMA instructions too)
        der["non_mfma_valu_per_kernel_value_lanes"] = round((c["SQ_INSTS_VALU"]["per_launch"] - c["SQ_INSTS_MFMA"]["per_launch"]) * 64.0 / nval, 2)
    res["derived"] = der
    with open(os.path.join(ROOT, "gpurun_out", "r06_predict_pmc.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
