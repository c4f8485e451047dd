"""Shader clock and board power of GPU 0 while a workload runs (is the attention kernel clock / power limited
inside bench.py?).  A host thread polls the driver -- amdsmi's gpu-metrics table when the Python binding is importable, the hwmon
files under /sys/class/drm/card*/device/hwmon otherwise -- every few milliseconds between start() and stop(); the calls release the
GIL while the workload sits in hipStreamSynchronize.  Measurement infrastructure for bench.py only; nothing in the library uses it."""
import glob
import os
import threading
import time


class PowerSampler:
    def __init__(self, device_index=0, period_s=0.003):
        self.period = period_s
        self.samples = []          # (t, sclk_mhz or None, watts or None)
        self._stop = threading.Event()
        self._thread = None
        self.source = None
        self._read = self._pick_reader(device_index)

    # ---- readers -------------------------------------------------------------------------------------------------------
    def _pick_reader(self, idx):
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            handles = amdsmi.amdsmi_get_processor_handles()
            h = handles[idx]

            def read():
                clk = pw = None
                try:
                    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                    clk = m.get("current_gfxclk")
                    if isinstance(m.get("current_gfxclks"), (list, tuple)):   # per-XCD clocks (MI300 family): mean of the valid ones
                        v = [c for c in m["current_gfxclks"] if isinstance(c, (int, float)) and 0 < c < 10000]
                        if v:
                            clk = sum(v) / len(v)
                    pw = m.get("current_socket_power")
                    if not isinstance(pw, (int, float)) or pw <= 0 or pw > 5000:
                        pw = m.get("average_socket_power")
                except Exception:
                    pass
                if not isinstance(clk, (int, float)) or clk <= 0 or clk > 10000:
                    try:
                        clk = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX).get("clk")
                    except Exception:
                        clk = None
                if not isinstance(pw, (int, float)) or pw <= 0 or pw > 5000:
                    try:
                        p = amdsmi.amdsmi_get_power_info(h)
                        pw = p.get("current_socket_power") or p.get("average_socket_power")
                    except Exception:
                        pw = None
                return (clk if isinstance(clk, (int, float)) else None, pw if isinstance(pw, (int, float)) else None)
            if read() != (None, None):
                self.source = "amdsmi gpu_metrics"
                return read
        except Exception:
            pass
        for card in sorted(glob.glob("/sys/class/drm/card*/device")):
            hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
            if not hw:
                continue
            f_clk = os.path.join(hw[0], "freq1_input")
            f_pw = next((p for p in (os.path.join(hw[0], "power1_input"), os.path.join(hw[0], "power1_average")) if os.path.exists(p)), None)
            if not os.path.exists(f_clk) and f_pw is None:
                continue

            def read(f_clk=f_clk, f_pw=f_pw):
                clk = pw = None
                try:
                    with open(f_clk) as fh:
                        clk = int(fh.read()) / 1e6
                except Exception:
                    pass
                try:
                    with open(f_pw) as fh:
                        pw = int(fh.read()) / 1e6
                except Exception:
                    pass
                return clk, pw
            if read() != (None, None):
                self.source = f"hwmon {hw[0]}"
                return read
        self.source = None
        return None

    # ---- sampling ------------------------------------------------------------------------------------------------------
    def start(self):
        self.samples = []
        if self._read is None:
            return self
        self._stop.clear()

        def loop():
            while not self._stop.is_set():
                c, p = self._read()
                self.samples.append((time.perf_counter(), c, p))
                time.sleep(self.period)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None
        return self.summary()

    def summary(self):
        if self._read is None:
            return {"source": None, "note": "neither amdsmi nor hwmon readable on this box"}
        clk = sorted(c for _, c, _ in self.samples if c)
        pw = sorted(p for _, _, p in self.samples if p)
        q = lambda v, f: round(v[min(len(v) - 1, int(f * len(v)))], 1) if v else None  # noqa: E731
        span = self.samples[-1][0] - self.samples[0][0] if len(self.samples) > 1 else 0.0
        return {"source": self.source, "samples": len(self.samples), "seconds": round(span, 3),
                "sclk_mhz": {"mean": round(sum(clk) / len(clk), 1) if clk else None, "p10": q(clk, 0.1), "p50": q(clk, 0.5),
                             "p90": q(clk, 0.9), "min": q(clk, 0.0), "max": q(clk, 1.0)},
                "power_w": {"mean": round(sum(pw) / len(pw), 1) if pw else None, "p50": q(pw, 0.5), "max": q(pw, 1.0)}}
