#!/usr/bin/env python3
"""Engine clock and package power WHILE a bench.py workload runs (box-to-box the Direct step ranges 154.6-166.9 ms with
identical code; the Barnes-Hut and spatial-hash steps do not move): samples sysfs (pp_dpm_sclk, hwmon power / freq) or
rocm-smi every 0.25 s beside a child process.
Usage: python tools/clock_sample.py [bench.py arguments ...]"""
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sysfs_sources():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(os.path.join(card, "pp_dpm_sclk")):
            out["sclk"] = os.path.join(card, "pp_dpm_sclk")
            for h in glob.glob(os.path.join(card, "hwmon/hwmon*")):
                for name in ("power1_average", "power1_input"):
                    if os.path.exists(os.path.join(h, name)):
                        out["power"] = os.path.join(h, name)
                if os.path.exists(os.path.join(h, "freq1_input")):
                    out["freq"] = os.path.join(h, "freq1_input")
            break
    return out


def read_sysfs(src):
    rec = {}
    try:
        if "freq" in src:
            rec["sclk_mhz"] = int(open(src["freq"]).read()) / 1e6
        elif "sclk" in src:
            for line in open(src["sclk"]):
                if "*" in line:
                    rec["sclk_mhz"] = float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
        if "power" in src:
            rec["power_w"] = int(open(src["power"]).read()) / 1e6
    except Exception as e:  # noqa: BLE001
        rec["error"] = str(e)
    return rec


def read_smi():
    try:
        r = subprocess.run(["rocm-smi", "-c", "-P", "--json"], capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout)
        card = next(iter(d.values()))
        rec = {}
        for k, v in card.items():
            if "sclk clock speed" in k.lower():
                rec["sclk_mhz"] = float(str(v).strip("()").lower().replace("mhz", ""))
            if "power" in k.lower() and "w" in k.lower():
                try:
                    rec["power_w"] = float(v)
                except ValueError:
                    pass
        return rec
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


args = sys.argv[1:] or ["--steps", "40", "--warmup", "5", "--no-extra", "--no-cpu-baseline"]
src = {} if os.environ.get("CLOCK_SAMPLE_SMI", "1") == "1" else sysfs_sources()  # sysfs card0 need not be the visible GPU
print("sources:", src or "rocm-smi", flush=True)
child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), *args], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
t0 = time.time()
samples = []
while child.poll() is None:
    rec = read_sysfs(src) if src else read_smi()
    rec["t"] = round(time.time() - t0, 2)
    samples.append(rec)
    time.sleep(0.25)
line = child.stdout.read().strip().splitlines()
busy = [s for s in samples if s.get("power_w", 0) > 600]
for s in samples[:: max(1, len(samples) // 40)]:
    print(s)
if busy:
    clk = [s["sclk_mhz"] for s in busy if "sclk_mhz" in s]
    pw = [s["power_w"] for s in busy]
    print(f"under load ({len(busy)} samples above 600 W): sclk mean {sum(clk) / max(1, len(clk)):.0f} MHz (min {min(clk):.0f}, max {max(clk):.0f}), "
          f"power mean {sum(pw) / len(pw):.0f} W (max {max(pw):.0f})")
if line:
    d = json.loads(line[-1])
    print(f"bench: {d['ms_per_step']:.2f} ms/step, roofline frac {d.get('roofline', {}).get('frac')}")
