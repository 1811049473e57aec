#!/usr/bin/env python3
"""What clock and power does the GPU run at DURING a workload? Samples the amdgpu hwmon files of every card (freq1_input = sclk in Hz,
power1_average / power1_input in microwatts, temp) every `--period` seconds in a thread while tools/ab_probe.py renders the workload
in a child process, and prints the distribution of the samples taken while the child ran. Answers whether a leg whose counters show a
low clock (profiles/r05_pmc_c3.md: 1.85 GHz against C2's 2.37) is power-limited, or clocked down between its many short launches.

    python tools/clock_power_probe.py c3 [--sqrtspp S] [--steps N] [--period 0.02]"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def sensors():
    out = []
    for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        s = {"dir": h}
        for key, names in (("freq", ["freq1_input"]), ("power", ["power1_average", "power1_input"]), ("temp", ["temp1_input"]), ("cap", ["power1_cap"])):
            for n in names:
                p = os.path.join(h, n)
                if os.path.exists(p):
                    s[key] = p
                    break
        if "freq" in s or "power" in s:
            out.append(s)
    return out


def read(p):
    try:
        return int(open(p).read().strip())
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--sqrtspp", type=int, default=None)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--period", type=float, default=0.02)
    a = ap.parse_args()
    sens = sensors()
    samples, stop = [], threading.Event()

    def loop():
        while not stop.is_set():
            t = time.perf_counter()
            samples.append((t, [(read(s.get("freq", "")) if "freq" in s else None, read(s.get("power", "")) if "power" in s else None) for s in sens]))
            time.sleep(a.period)

    th = threading.Thread(target=loop, daemon=True)
    th.start()
    cmd = [sys.executable, os.path.join(ROOT, "tools", "ab_probe.py"), a.workload, "--steps", str(a.steps), "base:"]
    if a.sqrtspp:
        cmd += ["--sqrtspp", str(a.sqrtspp)]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True)
    t1 = time.perf_counter()
    stop.set()
    th.join()
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    res = {"workload": a.workload, "probe": json.loads(line[-1]) if line else r.stderr[-300:], "cards": []}
    for i, s in enumerate(sens):
        f = [v[i][0] for t, v in samples if t0 <= t <= t1 and v[i][0]]
        p = [v[i][1] for t, v in samples if t0 <= t <= t1 and v[i][1]]
        if not f and not p:
            continue
        # the busy card: the one whose power moved
        q = lambda xs, k: sorted(xs)[min(len(xs) - 1, int(k * len(xs)))] if xs else None
        res["cards"].append({"hwmon": s["dir"], "cap_W": (read(s["cap"]) or 0) / 1e6 if "cap" in s else None, "samples": len(f),
                             "sclk_MHz": {k: (q(f, v) or 0) / 1e6 for k, v in (("p05", 0.05), ("p50", 0.5), ("p95", 0.95))},
                             "power_W": {k: (q(p, v) or 0) / 1e6 for k, v in (("p05", 0.05), ("p50", 0.5), ("p95", 0.95), ("max", 0.9999))}})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
