#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) results for the mcrt kernels.

  python tools/summarize_rocprof.py <dir with */*.db> > profiles/rNN_<what>.md

Reads every *_results.db under the directory: kernel-trace databases give per-kernel call counts and
average duration (the `--stats` view), PMC databases give per-kernel counter sums."""
import glob
import re
import os
import sqlite3
import sys


def short(name):
    """The kernel's name with its template arguments, without return type, namespaces and parameter list."""
    k = name.replace("(anonymous namespace)::", "").replace("mcrt::", "")
    if k.startswith("void "):
        k = k[5:]
    depth = 0
    for i, ch in enumerate(k):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            k = k[:i]
            break
    return k.strip()[:90]


def main(root):
    print("# rocprofv3 summary for `%s`\n" % root)
    for db in sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)):
        con = sqlite3.connect(db)
        rel = os.path.relpath(db, root)
        try:
            rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        except sqlite3.Error:
            rows = []
        rows = [r for r in rows if "mcrt" in r[0] or "Kernel" in r[0] and "anonymous" in r[0]]
        if rows:
            print("## kernel trace: %s\n" % rel)
            print("| kernel | calls | total ms | avg ms | % |")
            print("|---|---:|---:|---:|---:|")
            for n, c, tot, avg, pct in rows:
                print("| `%s` | %d | %.3f | %.3f | %.2f |" % (short(n), c, tot / 1e3, avg / 1e3, pct))
            try:
                k = list(con.execute("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count "
                                     "from kernels where name like '%renderKernel%' or name like '%wfTraceKernel%' or name like '%wfShadeKernel%'"))
                seen = set()
                for n, g, w, lds, scr, v, a, s in k:
                    if short(n) in seen:
                        continue
                    seen.add(short(n))
                    print("\nlaunch of `%s`: grid %d threads, block %d, LDS %d B/block, scratch %d B/lane, VGPR %d, AGPR %d, SGPR %d\n" % (short(n), g, w, lds, scr, v, a, s))
            except sqlite3.Error:
                pass
        try:
            rows = list(con.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                                    "group by kernel_name, counter_name"))
        except sqlite3.Error:
            rows = []
        rows = [r for r in rows if "Kernel" in r[0] and "anonymous" in r[0]]
        if rows:
            print("## counters: %s\n" % rel)
            print("| kernel | counter | sum over dispatches | dispatches |")
            print("|---|---|---:|---:|")
            for n, c, v, cnt in rows:
                print("| `%s` | %s | %.6g | %d |" % (short(n), c, v, cnt))
            print()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof")
