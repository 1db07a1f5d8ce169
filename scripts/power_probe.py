"""Direct power / clock reading of the headline kernel: loops the bench step for a few seconds while a thread samples
every GPU's hwmon (power1_input, freq1_input, power1_cap) at >= 20 Hz and amd-smi once a second; reports the card whose power rose.

  python scripts/power_probe.py [--seconds 3] [--max-blocks N] [--precision bf16x3_fused] [--zeros]   (UCNERF_LIB selects a variant library)
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uc_nerf_amd.pipeline import CoarseFineRenderer, flat_params_of  # noqa: E402
from uc_nerf_amd.synthetic import init_ucnerf_state_dict, make_scene, random_pixels, scene_to  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=3.0)
ap.add_argument("--max-blocks", type=int, default=0)
ap.add_argument("--precision", default="bf16x3_fused")
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--tag", default="")
args = ap.parse_args()

def _own_hwmon():
    """hwmon directories of THE card torch's device 0 is (matched by PCI bus id: the host's other cards run other people's jobs); all cards when
    the id cannot be matched."""
    every = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    try:
        pr = torch.cuda.get_device_properties(0)
        bus = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        own = [h for h in every if bus in os.path.realpath(h.split("/hwmon/")[0]).lower()]
        return own or every
    except Exception:      # noqa: BLE001
        return every


HW = _own_hwmon()


def rd(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


samples, smi, stop = [], [], threading.Event()


def sampler():
    while not stop.is_set():
        t = time.perf_counter()
        samples.append((t, [(rd(h + "/power1_input"), rd(h + "/freq1_input")) for h in HW]))
        time.sleep(0.02)


def smi_sampler():
    while not stop.is_set():
        t = time.perf_counter()
        try:
            r = subprocess.run(["amd-smi", "metric", "--power", "--clock", "--json"], capture_output=True, text=True, timeout=10)
            smi.append((t, json.loads(r.stdout)))
        except Exception as e:      # noqa: BLE001
            smi.append((t, {"error": str(e)[:200]}))
        stop.wait(0.7)


dev = torch.device("cuda:0")
scene = scene_to(make_scene(seed=0), dev)
sd = init_ucnerf_state_dict(seed=0, n_src=6, sigma_scale=0.05, sigma_bias=0.05)
r = CoarseFineRenderer(scene, flat_params_of(sd).to(dev), 64, 128, precision=args.precision, max_blocks=args.max_blocks)
xs, ys = random_pixels(args.rays, 256, 320, seed=0)
xs, ys = xs.to(dev), ys.to(dev)
noise = torch.rand(args.rays, 64, generator=torch.Generator().manual_seed(100)).to(dev)
for _ in range(50):
    r.render(xs, ys, perturb=1.0, noise=noise)
torch.cuda.synchronize()
idle = [(rd(h + "/power1_input"), rd(h + "/freq1_input")) for h in HW]
th, th2 = threading.Thread(target=sampler), threading.Thread(target=smi_sampler)
th.start(); th2.start()
t0 = time.perf_counter()
steps = 0
while time.perf_counter() - t0 < args.seconds:
    for _ in range(50):
        r.render(xs, ys, perturb=1.0, noise=noise)
    steps += 50
torch.cuda.synchronize()
t1 = time.perf_counter()
stop.set(); th.join(); th2.join()
load = [s for s in samples if t0 + 0.5 < s[0] < t1]          # skip the ramp
n = len(HW)
mean_p = [sum(s[1][i][0] or 0 for s in load) / max(1, len(load)) for i in range(n)]
card = max(range(n), key=lambda i: mean_p[i] - (idle[i][0] or 0)) if n else -1
out = {"tag": args.tag, "lib": os.environ.get("UCNERF_LIB", "default"), "precision": args.precision, "max_blocks": args.max_blocks, "rays": args.rays,
       "seconds": t1 - t0, "steps": steps, "ms_per_step": (t1 - t0) / steps * 1e3, "hwmon": HW[card] if n else None, "samples": len(load)}
if n:
    ps = [s[1][card][0] for s in load if s[1][card][0] is not None]
    fs = [s[1][card][1] for s in load if s[1][card][1] is not None]
    out.update(power_w_mean=sum(ps) / len(ps) / 1e6, power_w_max=max(ps) / 1e6, power_w_min=min(ps) / 1e6, power_cap_w=(rd(HW[card] + "/power1_cap") or 0) / 1e6,
               sclk_mhz_mean=sum(fs) / len(fs) / 1e6 if fs else None, sclk_mhz_min=min(fs) / 1e6 if fs else None, sclk_mhz_max=max(fs) / 1e6 if fs else None,
               power_w_before_load=(idle[card][0] or 0) / 1e6,
               trace=[(round(s[0] - t0, 3), round((s[1][card][0] or 0) / 1e6, 1), round((s[1][card][1] or 0) / 1e6)) for s in load[::5]])
out["amd_smi"] = [(round(t - t0, 2), m) for t, m in smi if t0 < t < t1][:3]
print(json.dumps(out))
