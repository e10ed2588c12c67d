"""Engine-clock / socket-power sampler for bench.py ("sustained" must be a number with a source).

A background thread polls librocm_smi64 through ctypes (no subprocess: `rocm-smi` / `amd-smi` take 0.3-0.5 s per call, far from the
>= 10 Hz asked for) while the caller keeps the GPU busy:
    rsmi_dev_gpu_clk_freq_get(dev, RSMI_CLK_TYPE_SYS)   -> current engine clock (the frequency table's `current` entry)
    rsmi_dev_current_socket_power_get / rsmi_dev_power_ave_get -> socket power in microwatts
Falls back to sysfs (pp_dpm_sclk's starred line, hwmon power1_average / power1_input) and finally to "unavailable": the caller
reports whatever source worked, never a guessed number.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import statistics
import threading
import time


class _Freqs(C.Structure):  # rsmi_frequencies_t (rocm_smi.h): has_deep_sleep, num_supported, current, frequency[33] in Hz
    _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32),
                ("frequency", C.c_uint64 * 33)]


class ClockSampler:
    def __init__(self, device_index: int = 0, hz: float = 20.0):
        self.dev, self.period = device_index, 1.0 / hz
        self.samples = []  # (t, sclk_mhz or None, power_w or None)
        self.source = {"sclk": None, "power": None}
        self._stop = threading.Event()
        self._thr = None
        self._lib = None
        for name in ("librocm_smi64.so", "/opt/rocm/lib/librocm_smi64.so", "librocm_smi64.so.1"):
            try:
                lib = C.CDLL(name)
                if lib.rsmi_init(C.c_uint64(0)) == 0:
                    self._lib = lib
                    break
            except OSError:
                continue
        self._sysfs = None
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        if cards:
            self._sysfs = os.path.dirname(cards[min(device_index, len(cards) - 1)])

    # -- single readings -----------------------------------------------------------------------------------------
    def _sclk(self):
        if self._lib is not None:
            f = _Freqs()
            try:
                if self._lib.rsmi_dev_gpu_clk_freq_get(C.c_uint32(self.dev), C.c_int(0), C.byref(f)) == 0 and f.num_supported > 0 \
                        and f.current < f.num_supported:
                    self.source["sclk"] = "rsmi_dev_gpu_clk_freq_get(RSMI_CLK_TYPE_SYS)"
                    return f.frequency[f.current] / 1e6
            except Exception:
                pass
        if self._sysfs:
            try:
                with open(os.path.join(self._sysfs, "pp_dpm_sclk")) as fh:
                    for line in fh:
                        if "*" in line:
                            self.source["sclk"] = "sysfs pp_dpm_sclk"
                            return float(line.split(":")[1].strip().split("Mhz")[0].split("MHz")[0])
            except Exception:
                pass
        return None

    def _power(self):
        if self._lib is not None:
            uw = C.c_uint64(0)
            for fn, args in (("rsmi_dev_current_socket_power_get", (C.c_uint32(self.dev), C.byref(uw))),
                             ("rsmi_dev_power_ave_get", (C.c_uint32(self.dev), C.c_uint32(0), C.byref(uw)))):
                try:
                    if getattr(self._lib, fn)(*args) == 0 and uw.value > 0:
                        self.source["power"] = fn
                        return uw.value / 1e6
                except Exception:
                    continue
        if self._sysfs:
            for name in ("power1_average", "power1_input"):
                for p in glob.glob(os.path.join(self._sysfs, "hwmon", "hwmon*", name)):
                    try:
                        with open(p) as fh:
                            self.source["power"] = f"sysfs hwmon {name}"
                            return float(fh.read().strip()) / 1e6
                    except Exception:
                        continue
        return None

    # -- sampling loop ---------------------------------------------------------------------------------------------
    def _run(self):
        t0 = time.perf_counter()
        while not self._stop.is_set():
            t = time.perf_counter()
            self.samples.append((t - t0, self._sclk(), self._power()))
            dt = self.period - (time.perf_counter() - t)
            if dt > 0:
                self._stop.wait(dt)

    def start(self):
        self.samples = []
        self._stop.clear()
        self._thr = threading.Thread(target=self._run, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thr is not None:
            self._thr.join(timeout=2.0)
        return self.summary()

    def summary(self):
        def stat(vals):
            vals = [v for v in vals if v is not None]
            if not vals:
                return None
            return {"min": round(min(vals), 1), "median": round(statistics.median(vals), 1), "max": round(max(vals), 1)}

        n = len(self.samples)
        span = self.samples[-1][0] - self.samples[0][0] if n > 1 else 0.0
        return {"sclk_mhz": stat([s[1] for s in self.samples]), "power_w": stat([s[2] for s in self.samples]), "samples": n,
                "rate_hz": round((n - 1) / span, 1) if span > 0 else None, "source": dict(self.source)}


if __name__ == "__main__":  # python tools/clock_sampler.py [seconds]: idle readings
    import sys

    s = ClockSampler().start()
    time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
    print(s.stop())
