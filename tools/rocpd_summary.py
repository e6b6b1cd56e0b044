#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd (sqlite) result into the per-kernel summary committed under profiles/.
usage: rocpd_summary.py results.db [out.md]   (kernel trace: top_kernels view; counters: pmc_events view)"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    lines = ["| kernel | calls | total us | avg us | % |", "|---|---|---|---|---|"]
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = name.split("(")[0][:90]
        lines.append(f"| {short} | {calls} | {total:.1f} | {avg:.1f} | {pct:.2f} |")
    try:
        rows = list(cur.execute("select name, counter_name, sum(counter_value), count(distinct dispatch_id) from pmc_events group by name, counter_name"))
    except Exception:
        rows = []
    if rows:
        lines += ["", "| kernel | counter | sum | dispatches |", "|---|---|---|---|"]
        for k, p, v, c in rows:
            lines.append(f"| {k.split('(')[0][:60]} | {p} | {v:.0f} | {c} |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
