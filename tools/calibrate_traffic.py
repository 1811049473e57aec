#!/usr/bin/env python3
"""Known-byte calibration of the L2 memory-side counters (FETCH_SIZE / WRITE_SIZE and the raw TCC_EA0_* request counters)
for the access patterns of the mcrt kernels. Runs tools/_build/traffic_calib (tools/traffic_calib.hip) under rocprofv3, one
pass per counter set, for a working set that fits the 256 MiB Infinity Cache and one that does not, and prints for every
pattern: bytes asked for, bytes each formula reports, the ratio.

  python tools/calibrate_traffic.py [--out gpurun_out/traffic_calibration.json]

The result decides how bench.py turns counters into `traffic` (bench.py: counters_summary)."""
import argparse
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "_build", "traffic_calib")
PASSES = (
    ("rdreq", ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]),
    ("fetch", ["FETCH_SIZE"]),
    ("write", ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"]),
    ("wrreq", ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_DRAM_sum"]),
)


def build():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    src = os.path.join(ROOT, "tools", "traffic_calib.hip")
    if not os.path.exists(BIN) or os.path.getmtime(src) > os.path.getmtime(BIN):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-o", BIN, src])
    return BIN


def run_pass(names, ws_mib, move_mib):
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    tmp = tempfile.mkdtemp(prefix="mcrt_calib_", dir="/tmp")
    try:
        p = subprocess.run([rocprof, "--kernel-trace", "--pmc"] + names + ["-d", tmp, "--", BIN, str(ws_mib), str(move_mib)],
                           capture_output=True, text=True, timeout=600, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        line = next((l for l in p.stdout.splitlines() if l.startswith("{\"working_set_bytes\"")), None)
        if p.returncode != 0 or line is None:
            return None, {}, "rc %d: %s" % (p.returncode, (p.stderr or p.stdout)[-400:])
        asked = json.loads(line)
        sums = {}
        for db in glob.glob(os.path.join(tmp, "**", "*_results.db"), recursive=True):
            con = sqlite3.connect(db)
            try:
                for kname, cname, val in con.execute("select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name"):
                    key = next((k for k in asked if k.startswith("calib") and k in kname), None)
                    if key:
                        sums.setdefault(key, {})[cname] = sums.get(key, {}).get(cname, 0.0) + float(val)
            finally:
                con.close()
        return asked, sums, None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "traffic_calibration.json"))
    ap.add_argument("--move-mib", type=int, default=8192)
    args = ap.parse_args()
    build()
    report = {"tool": "tools/calibrate_traffic.py", "working_sets": {}}
    for ws in (128, 4096):
        asked_all, counters, errors = None, {}, []
        for tag, names in PASSES:
            asked, sums, err = run_pass(names, ws, args.move_mib)
            if err:
                errors.append("%s: %s" % (tag, err))
                continue
            asked_all = asked
            for k, v in sums.items():
                counters.setdefault(k, {}).update(v)
        rows = {}
        for k, c in counters.items():
            b = asked_all[k]["bytes_per_launch"] * asked_all[k]["launches"]
            r = {"asked_bytes": b, "GBs": asked_all[k]["GBs"], "counters": c}
            if "TCC_EA0_RDREQ_sum" in c:
                n, n32, n64, n128 = (c.get(x, 0.0) for x in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
                r["read_bytes_by_size_classes"] = 32 * n32 + 64 * n64 + 128 * n128
                r["read_bytes_rest_as_64"] = 32 * n32 + 128 * n128 + 64 * (n - n32 - n128)
            if "FETCH_SIZE" in c:
                r["fetch_size_bytes"] = c["FETCH_SIZE"] * 1024.0
            if "WRITE_SIZE" in c:
                r["write_size_bytes"] = c["WRITE_SIZE"] * 1024.0
            if "TCC_HIT_sum" in c:
                r["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0.0), 1.0)
            is_write = k.startswith("calibWrite")
            for f in ("read_bytes_by_size_classes", "read_bytes_rest_as_64", "fetch_size_bytes", "write_size_bytes"):
                if f in r and (f.startswith("write") == is_write):
                    r["ratio_" + f] = r[f] / b
            rows[k] = r
        report["working_sets"]["%d MiB" % ws] = {"patterns": rows, "errors": errors}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    for ws, d in report["working_sets"].items():
        print("working set", ws, d["errors"] or "")
        for k, r in d["patterns"].items():
            print("  %-14s asked %8.2f GB at %6.0f GB/s |" % (k, r["asked_bytes"] / 1e9, r["GBs"]),
                  " ".join("%s=%.3f" % (n[6:], v) for n, v in r.items() if n.startswith("ratio_")), "| L2 hit %.3f" % r.get("l2_hit_rate", -1))


if __name__ == "__main__":
    sys.exit(main())
