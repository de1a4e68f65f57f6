"""Power and shader clock of the GPU while the dycore steps on the cloud-free initial state and on the developed storm: one process, a sampler thread reading the amdgpu hwmon files
(power1_average / power1_input, freq1_input) every few milliseconds during each steady loop.  No profiler (a profiled arm runs at another
clock: MI355X_MICROARCH.md, DVFS give-back).  python tools/power_clock_sample.py [--file /tmp/storm.pt] [--steps 300] -> one JSON line."""
import argparse, glob, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miniweatherml_amd import modules

ap = argparse.ArgumentParser()
ap.add_argument("--file", default="/tmp/storm.pt"); ap.add_argument("--steps", type=int, default=300); ap.add_argument("--period-ms", type=float, default=5.0)
a = ap.parse_args()
NAMES = ("density_dry", "uvel", "vvel", "wvel", "temp", "water_vapor", "cloud_liquid", "precip_liquid")


def my_pci_bus_id():
    """PCI address of HIP device 0 (hipDeviceGetPCIBusId), e.g. 0000:05:00.0 -- the box has more cards than this process may use."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            return buf.value.decode().lower()
    except OSError:
        pass
    return None


def hwmon():
    out = {}
    want = my_pci_bus_id()
    cards = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")
    if want:
        mine = [h for h in cards if want in os.path.realpath(os.path.join(h, "..", "..")).lower()]
        cards = mine or cards
        out["pci_bus_id"] = want
        out["matched_card"] = bool(mine)
    for h in cards:
        found = {}
        for key, names in (("power_uW", ("power1_average", "power1_input")), ("sclk_Hz", ("freq1_input",)), ("mclk_Hz", ("freq2_input",)), ("temp_mC", ("temp1_input",))):
            for n in names:
                p = os.path.join(h, n)
                if key not in found and os.path.exists(p):
                    try:
                        float(open(p).read()); found[key] = p
                    except (OSError, ValueError):
                        pass
        if found:
            out.update(found)
            return out
    return out


class Sampler(threading.Thread):
    def __init__(self, files, period):
        super().__init__(daemon=True); self.files, self.period, self.rows, self.on = files, period, [], True
    def run(self):
        while self.on:
            r = {}
            for k, p in self.files.items():
                if not isinstance(p, str) or not p.startswith("/sys"):
                    continue
                try:
                    r[k] = float(open(p).read())
                except (OSError, ValueError):
                    pass
            self.rows.append(r); time.sleep(self.period)
    def stats(self):
        out = {"samples": len(self.rows)}
        for k in self.files:
            v = [r[k] for r in self.rows if k in r]
            if v:
                v.sort(); out[k] = {"mean": sum(v) / len(v), "min": v[0], "p50": v[len(v) // 2], "max": v[-1]}
        return out


nx, ny, nz = 400, 400, 100
c, d, m, n = modules.make_supercell(nx, ny, nz, 1, 500.0 * nx, 500.0 * ny, 20000.0, with_nudger=True)
dt = d.compute_time_step(c)
dm = c.get_data_manager_readwrite()
init = {k: dm.get(k).clone() for k in NAMES}
files = hwmon()
res = {"hwmon_files": files, "steps_per_state": a.steps}


def run(label, state, dycore_only=False):
    for k in NAMES:
        dm.get(k).copy_(state[k])
    step = (lambda: d.time_step(c, dt)) if dycore_only else (lambda: modules.supercell_step(c, d, m, n, dt))
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    s = Sampler(files, a.period_ms * 1e-3); s.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        step()
    e1.record(); torch.cuda.synchronize()
    s.on = False; s.join()
    res[label] = dict(s.stats(), ms_per_step=e0.elapsed_time(e1) / a.steps)


run("initial", init)                                           # (the complete loop -- dycore, Kessler, sponge, nudger -- so that the states stay what they are)
if os.path.exists(a.file):
    st = {k: v.to(dm.get(k).device) for k, v in torch.load(a.file).items()}
    run("storm", st)
    run("initial_again", init)
    run("storm_dycore_only", st, True)
    run("initial_dycore_only", init, True)
print(json.dumps(res))
