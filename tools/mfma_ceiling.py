#!/usr/bin/env python
"""The chip's MFMA ceiling, varied (round 6) -> profiles/r06_mfma_ceiling.txt

    python tools/mfma_ceiling.py [out.txt]

Register-only MFMA issue loops (tools/ubench/mfma_ceiling.hip) over: instruction shape (f64 16x16x4, f64 4x4x4 4-block, f32 16x16x4,
f32 32x32x2), independent accumulators per wave (4 / 8 / 16; 2 / 4 / 8 for 32x32x2), waves per SIMD (1 / 2 / 4), zero vs non-zero
operands, and a short (~100 us, one launch) vs a long (~3 s of back-to-back launches) measurement, with the engine clock and the socket
power sampled at 20 Hz through librocm_smi64 (tools/clock_sampler.py) during the long ones.  `of peak` is against the datasheet:
78.6 TF fp64, 157.3 TF fp32."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from clock_sampler import ClockSampler  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "tools", "ubench", "libmfma_ceiling.so"))
lib.mfma_ceiling_run.restype = C.c_double
lib.mfma_ceiling_run.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_double)]
SHAPES = {0: ("f64 16x16x4", 78.6, 64), 1: ("f64 4x4x4 (4 blocks)", 78.6, 16), 2: ("f32 16x16x4", 157.3, 32), 3: ("f32 32x32x2", 157.3, 64)}


def run(shape, nacc, wps, iters, zero, launches):
    fl = C.c_double()
    t = lib.mfma_ceiling_run(shape, nacc, wps, iters, zero, launches, C.byref(fl))
    return t, fl.value


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout

    def emit(s):
        print(s, file=out, flush=True)
        if out is not sys.stdout:
            print(s, flush=True)

    idle = ClockSampler(0, 20.0).start()
    time.sleep(1.0)
    i = idle.stop()
    emit(f"# idle: sclk {i['sclk_mhz']} MHz, power {i['power_w']} W, source {i['source']}")
    emit("# shape | accumulators | waves/SIMD | operands | short: TF (of peak) [us] | long 3 s: TF (of peak), sclk MHz median (min), power W median (max)")
    for shape, (name, peak, cyc) in SHAPES.items():
        accs = (2, 4, 8) if shape == 3 else (4, 8, 16)
        for nacc in accs:
            for wps in (1, 2, 4):
                for zero in (0, 1):
                    if zero and not (nacc == accs[1] and wps in (1, 4)):
                        continue  # zero operands: two configurations per shape are enough to see the data dependence of the power
                    # short: one launch of ~100 us (iterations from the nominal cycle count at 2.4 GHz)
                    it_short = max(8, int(100e-6 * 2.4e9 / (cyc * nacc * wps)))
                    t, fl = run(shape, nacc, wps, it_short, zero, 1)
                    short = f"{fl / t / 1e12:6.1f} ({fl / t / 1e12 / peak:.2f}) [{t * 1e6:5.0f} us]"
                    time.sleep(0.3)  # cool-down between configurations
                    # long: launches of ~10 ms back to back for 3 s
                    it_long = it_short * 100
                    t1, _ = run(shape, nacc, wps, it_long, zero, 1)
                    n = max(1, int(3.0 / max(t1, 1e-4)))
                    smp = ClockSampler(0, 20.0).start()
                    t, fl = run(shape, nacc, wps, it_long, zero, n)
                    s = smp.stop()
                    ck, pw = s["sclk_mhz"] or {}, s["power_w"] or {}
                    emit(f"{name:22s} | {nacc:2d} | {wps} | {'zero   ' if zero else 'nonzero'} | {short} | {fl / t / 1e12:6.1f} ({fl / t / 1e12 / peak:.2f}), "
                         f"sclk {ck.get('median')} ({ck.get('min')}), power {pw.get('median')} ({pw.get('max')})")
                    time.sleep(1.0)


if __name__ == "__main__":
    main()
