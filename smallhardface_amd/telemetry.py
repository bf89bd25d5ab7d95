"""Host-side clock / power sampler for the benchmark line (bench.py `telemetry`): reads the amdgpu hwmon files of ONE
card -- ``freq1_input`` (sclk, Hz), ``power1_input`` (socket power, microwatts), ``power1_cap`` -- from a thread, no GPU
call of its own, so that "the chip is power-limited under the conv stack" is a sampled clock and a sampled wattage of the
timed run instead of a figure derived from separate counter passes.  Measurement only: nothing in
the product path imports this module.  (The reference has no counterpart: its timers are lib/utils/timer.py.)"""
import glob
import os
import threading
import time


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def find_card(pci_bus_id=None):
    """hwmon directory of the amdgpu card with this PCI bus id ("0000:c5:00.0"; matched against the card's uevent
    PCI_SLOT_NAME, case-insensitively); with no id (or no match) None -- callers then sample every card and keep the
    busiest (``Sampler(card=None)``)."""
    want = (pci_bus_id or "").strip().lower()
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        try:
            if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                continue
            ue = open(os.path.join(dev, "uevent")).read().lower()
        except OSError:
            continue
        slot = [l.split("=", 1)[1].strip() for l in ue.splitlines() if l.startswith("pci_slot_name=")]
        hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
        if want and slot and slot[0] == want and hw:
            return hw[0]
    return None


def all_cards():
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        try:
            if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                continue
        except OSError:
            continue
        out += sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))[:1]
    return out


class Sampler(object):
    """``with Sampler(hwmon_dir) as s: ...`` then ``s.summary()``.  ``hwmon_dir`` None: every amdgpu card is sampled and
    the summary is that of the card with the highest mean power (the one this process loads)."""

    def __init__(self, hwmon_dir=None, period_s=0.02):
        self.dirs = [hwmon_dir] if hwmon_dir else all_cards()
        self.period = float(period_s)
        self.rows = {d: [] for d in self.dirs}      # (t, sclk_hz, power_uw)
        self._stop = threading.Event()
        self._th = None

    def _run(self):
        while not self._stop.is_set():
            t = time.perf_counter()
            for d in self.dirs:
                f, p = _read_int(os.path.join(d, "freq1_input")), _read_int(os.path.join(d, "power1_input"))
                if f is None and p is None:
                    p = _read_int(os.path.join(d, "power1_average"))
                if f is not None or p is not None:
                    self.rows[d].append((t, f, p))
            self._stop.wait(self.period)

    def __enter__(self):
        if self.dirs:
            self._th = threading.Thread(target=self._run, name="shf-telemetry", daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th is not None:
            self._th.join(timeout=2.0)
        return False

    def summary(self):
        """None when nothing could be read (no amdgpu hwmon files visible to this user)."""
        best, best_key = None, None
        for d, rows in self.rows.items():
            if not rows:
                continue
            fs = [r[1] for r in rows if r[1] is not None]
            ps = [r[2] for r in rows if r[2] is not None]
            cap = _read_int(os.path.join(d, "power1_cap"))
            s = {"samples": len(rows), "period_ms": 1000.0 * self.period,
                 "sclk_mhz_mean": (sum(fs) / len(fs) / 1e6) if fs else None,
                 "sclk_mhz_min": (min(fs) / 1e6) if fs else None,
                 "sclk_mhz_max": (max(fs) / 1e6) if fs else None,
                 "power_w_mean": (sum(ps) / len(ps) / 1e6) if ps else None,
                 "power_w_max": (max(ps) / 1e6) if ps else None,
                 "power_cap_w": (cap / 1e6) if cap else None,
                 "hwmon": d, "cards_sampled": len(self.dirs),
                 "source": "amdgpu hwmon freq1_input (sclk) / power1_input (socket power) / power1_cap, read from the host "
                           "side by a sampler thread"}
            key = s["power_w_mean"] if s["power_w_mean"] is not None else -1.0
            if best is None or key > best_key:
                best, best_key = s, key
        return best
