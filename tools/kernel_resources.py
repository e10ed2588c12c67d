#!/usr/bin/env python
"""Register / scratch / LDS table of every kernel in libagp_hip.so, from the compiler's own remarks.

    python tools/kernel_resources.py [--out profiles/r06_kernel_resources.txt] [--remarks FILE]

Compiles csrc/agp_capi.hip with build()'s exact flags plus -Rpass-analysis=kernel-resource-usage (cross-compiles for gfx950
without a GPU, ~80 s), demangles the kernel names (llvm-cxxfilt) and writes one row per kernel:
kernel, VGPRs, AGPRs, SGPRs, scratch bytes/lane, occupancy (waves/SIMD), LDS bytes/block.  `--remarks FILE` parses a remark
file produced earlier instead of compiling.  tests/test_kernel_resources.py reads the committed table and fails when a kernel
on the per-step / per-check whitelist reports scratch.
"""
import argparse
import glob
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "augmentedgaussianprocesses.jl_amd", "csrc", "agp_capi.hip")
CXXFILT = "c++filt"  # binutils (the ROCm image ships no llvm-cxxfilt)

FIELDS = [("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("TotalSGPRs", "sgpr"), ("ScratchSize [bytes/lane]", "scratch"),
          ("Occupancy [waves/SIMD]", "occupancy"), ("LDS Size [bytes/block]", "lds"), ("VGPRs Spill", "vgpr_spill"),
          ("SGPRs Spill", "sgpr_spill")]


def source_hash():
    """sha256 over the device sources the table was generated from (csrc/*.h, csrc/*.hip, in name order)"""
    h = hashlib.sha256()
    d = os.path.dirname(SRC)
    for f in sorted(glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.hip"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def compile_remarks():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as td:
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
               "-o", os.path.join(td, "lib.so"), SRC]
        print("[kernel_resources]", " ".join(cmd), file=sys.stderr, flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-4000:])
            raise SystemExit(r.returncode)
        return r.stderr


def parse(text):
    """-> list of dicts (mangled name + the numeric fields), in the order the compiler reports them"""
    rows, cur = [], None
    for line in text.splitlines():
        if "remark:" not in line:
            continue
        body = line.split("remark:", 1)[1].replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
        if body.startswith("Function Name:"):
            cur = {"mangled": body.split(":", 1)[1].strip()}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for label, key in FIELDS:
            if body.startswith(label + ":"):
                v = body[len(label) + 1:].strip()
                cur[key] = int(v) if re.fullmatch(r"-?\d+", v) else v
    return rows


def demangle(names):
    r = subprocess.run([CXXFILT], input="\n".join(names) + "\n", capture_output=True, text=True, check=True)
    out = r.stdout.splitlines()
    assert len(out) == len(names)
    return out


def short(name):
    """`void agp::k_foo<double, 1>(args...)` -> `k_foo<double, 1>`"""
    s = re.sub(r"^void\s+", "", name)
    depth, end = 0, len(s)
    for i, ch in enumerate(s):  # the argument list starts at the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            end = i
            break
    s = s[:end]
    return s.replace("agp::", "").replace("(anonymous namespace)::", "")


def table(rows):
    names = [short(n) for n in demangle([r["mangled"] for r in rows])]
    for r, n in zip(rows, names):
        r["kernel"] = n
    rows = sorted(rows, key=lambda r: r["kernel"])
    w = max(len(r["kernel"]) for r in rows)
    head = f"{'kernel':<{w}}  {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'occ':>4} {'LDS':>7}"
    lines = [head, "-" * len(head)]
    for r in rows:
        lines.append(f"{r['kernel']:<{w}}  {r.get('vgpr', -1):>5} {r.get('agpr', -1):>5} {r.get('sgpr', -1):>5} "
                     f"{r.get('scratch', -1):>8} {r.get('occupancy', -1):>4} {r.get('lds', -1):>7}")
    n_scr = sum(1 for r in rows if r.get("scratch", 0))
    lines.append("")
    lines.append(f"{len(rows)} kernels, {n_scr} with scratch")
    return "\n".join(lines) + "\n", rows


def read_table(path):
    """the committed table -> {kernel: dict(vgpr, agpr, sgpr, scratch, occupancy, lds)} (+ key "_sha256": the sources' hash)"""
    out = {}
    with open(path) as fh:
        for line in fh:
            if line.startswith("# sources sha256:"):
                out["_sha256"] = line.split(":", 1)[1].strip()
            m = re.match(r"^(k_\S.*?)\s{2,}(-?\d+)\s+(-?\d+)\s+(-?\d+)\s+(-?\d+)\s+(-?\d+)\s+(-?\d+)\s*$", line)
            if m:
                out[m.group(1)] = dict(zip(("vgpr", "agpr", "sgpr", "scratch", "occupancy", "lds"), map(int, m.groups()[1:])))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_kernel_resources.txt"))
    ap.add_argument("--remarks", default=None)
    a = ap.parse_args()
    text = open(a.remarks).read() if a.remarks else compile_remarks()
    tab, rows = table(parse(text))
    hdr = ("# every kernel of libagp_hip.so: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Rpass-analysis=kernel-resource-usage\n"
           "# (tools/kernel_resources.py; scratch = bytes per lane, occ = waves per SIMD, LDS = static bytes per workgroup)\n"
           f"# sources sha256: {source_hash()}\n")
    with open(a.out, "w") as fh:
        fh.write(hdr + tab)
    print(tab.splitlines()[-1], "->", a.out)
    for r in rows:
        if r.get("scratch", 0):
            print(f"  scratch {r['scratch']:>5} B/lane  {r['kernel']}")


if __name__ == "__main__":
    main()
