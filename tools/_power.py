"""Board power of the GPU while a measurement runs (read-only: the amdgpu hwmon file `power1_average` / `power1_input`, microwatts).

    with PowerSampler() as ps: ...run...      ->  ps.mean_w, ps.max_w, ps.n   (None when the box exposes no hwmon power file)

A daemon thread polls the file every `period` seconds.  Used by tools/overlap_probe.py and tools/gemm_sustain.py to say whether a loop runs at the board's power cap
(the sustained shader clock under the GEMMs is 1.5 - 1.9 of 2.4 GHz: the part clocks to its power budget).
"""
import glob
import threading
import time


def _power_file():
    for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input",
                "/sys/class/hwmon/hwmon*/power1_average", "/sys/class/hwmon/hwmon*/power1_input"):
        for f in sorted(glob.glob(pat)):
            try:
                if int(open(f).read().strip()) > 0:
                    return f
            except (OSError, ValueError):
                continue
    return None


def power_cap_w():
    for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap", "/sys/class/hwmon/hwmon*/power1_cap"):
        for f in sorted(glob.glob(pat)):
            try:
                v = int(open(f).read().strip())
                if v > 0:
                    return v * 1e-6
            except (OSError, ValueError):
                continue
    return None


class PowerSampler:
    def __init__(self, period=0.01):
        self.file, self.period = _power_file(), period
        self.samples, self._stop, self._t = [], threading.Event(), None

    def __enter__(self):
        if self.file:
            self._t = threading.Thread(target=self._run, daemon=True)
            self._t.start()
        return self

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append(int(open(self.file).read().strip()) * 1e-6)
            except (OSError, ValueError):
                pass
            time.sleep(self.period)

    def __exit__(self, *exc):
        self._stop.set()
        if self._t:
            self._t.join()
        return False

    @property
    def n(self):
        return len(self.samples)

    @property
    def mean_w(self):
        return round(sum(self.samples) / len(self.samples), 1) if self.samples else None

    @property
    def max_w(self):
        return round(max(self.samples), 1) if self.samples else None
