"""Socket power, shader clock and the energy accumulator of GPU 0 while something runs (bench.py's timed loop, tools/energy_probe.py).
Measurement plumbing only: amdsmi (python bindings in the image) first, the hwmon files of the amdgpu driver as a fallback; a box
that offers neither yields ``{"available": False}`` — callers print that instead of a number."""
from __future__ import annotations

import glob
import threading
import time
from typing import Dict, Optional


class _Smi:
    def __init__(self):
        import amdsmi
        self.smi = amdsmi
        amdsmi.amdsmi_init()
        self.h = amdsmi.amdsmi_get_processor_handles()[0]

    def power_w(self) -> Optional[float]:
        p = self.smi.amdsmi_get_power_info(self.h)
        for k in ("current_socket_power", "average_socket_power", "socket_power"):
            v = p.get(k)
            if isinstance(v, (int, float)) and 0 < v < 5000:
                return float(v)
        return None

    def sclk_mhz(self) -> Optional[float]:
        try:
            c = self.smi.amdsmi_get_clock_info(self.h, self.smi.AmdSmiClkType.GFX)
            for k in ("clk", "cur_clk", "current_clk"):
                v = c.get(k)
                if isinstance(v, (int, float)) and 0 < v < 10000:
                    return float(v)
        except Exception:
            pass
        return None

    def energy_j(self) -> Optional[float]:
        try:
            e = self.smi.amdsmi_get_energy_count(self.h)
            acc = e.get("energy_accumulator", e.get("power"))
            res = e.get("counter_resolution", 15.3)                  # micro-joules per count
            return float(acc) * float(res) * 1e-6
        except Exception:
            return None


class _Hwmon:
    def __init__(self):
        cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
                       + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average"))
        if not cands:
            raise RuntimeError("no hwmon power file")
        self.pfile = cands[0]
        f = glob.glob(self.pfile.rsplit("/", 1)[0] + "/freq1_input")
        self.ffile = f[0] if f else None

    def power_w(self):
        try:
            return int(open(self.pfile).read()) * 1e-6
        except Exception:
            return None

    def sclk_mhz(self):
        try:
            return int(open(self.ffile).read()) * 1e-6 if self.ffile else None
        except Exception:
            return None

    def energy_j(self):
        return None


def _source():
    for cls in (_Smi, _Hwmon):
        try:
            s = cls()
            if s.power_w() is not None:
                return s
        except Exception:
            continue
    return None


class PowerSampler:
    """``with PowerSampler() as ps: ...`` then ``ps.summary()``: mean / max socket power, mean shader clock over the samples taken every
    ``interval`` seconds (the first ``skip`` seconds dropped), and the energy accumulator's difference where the device has one."""

    def __init__(self, interval: float = 0.01, skip: float = 0.0):
        self.src = _source()
        self.interval, self.skip = interval, skip
        self.samples = []
        self._stop = threading.Event()
        self._th = None
        self.e0 = self.e1 = None
        self.t0 = self.t1 = 0.0

    def __enter__(self):
        if self.src is not None:
            self.e0 = self.src.energy_j()
            self.t0 = time.perf_counter()
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def _run(self):
        while not self._stop.is_set():
            t = time.perf_counter() - self.t0
            self.samples.append((t, self.src.power_w(), self.src.sclk_mhz()))
            self._stop.wait(self.interval)

    def __exit__(self, *exc):
        if self.src is not None:
            self._stop.set()
            self._th.join()
            self.t1 = time.perf_counter()
            self.e1 = self.src.energy_j()
        return False

    def summary(self) -> Dict:
        if self.src is None:
            return {"available": False}
        rows = [(p, c) for t, p, c in self.samples if t >= self.skip and p is not None]
        if not rows:
            return {"available": False, "samples": 0}
        pw = [p for p, _ in rows]
        ck = [c for _, c in rows if c is not None]
        out = {"available": True, "source": type(self.src).__name__.strip("_").lower(), "samples": len(rows), "seconds": self.t1 - self.t0,
               "mean_w": sum(pw) / len(pw), "max_w": max(pw), "min_w": min(pw)}
        if ck:
            out["mean_sclk_mhz"] = sum(ck) / len(ck)
            out["min_sclk_mhz"] = min(ck)
        if self.e0 is not None and self.e1 is not None and self.e1 > self.e0:
            out["energy_j"] = self.e1 - self.e0
            out["energy_mean_w"] = (self.e1 - self.e0) / max(self.t1 - self.t0, 1e-9)
        return out
