#!/usr/bin/env python3
"""Idle gaps of the last timed step in a rocprofv3 kernel trace (rocpd .db) of `bench.py`: from the last `first` kernel (default: the encode's first
suffix-sort histogram) to the end of the trace, every stretch longer than min_us in which no kernel ran, with the kernels either side.
usage: step_gaps.py results.db [first_kernel_substring] [min_us]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "knz_bwt_keys"
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
def short(n):
    m = re.search(r'knz_\w+', n)
    return (m.group(0) if m else n)[:60]
idx = [i for i, r in enumerate(rows) if first in r[0]]
if not idx:
    names = sorted({short(r[0]) for r in rows}); print("no kernel matches", first, "; kernels:", ", ".join(names)); sys.exit(1)
# the last step: the last match that has a gap of > 1 ms of other steps' work in front of it is hard to tell; take the last run of matches
i0 = idx[-1]
while i0 - 1 in idx: i0 -= 1
t0 = rows[i0][1]; busy_end = rows[i0][1]; prev = None; total_gap = 0.0
for name, s, e in rows[i0:]:
    if 'at::native' in name: break
    gap = (s - busy_end) / 1e3
    if gap > min_us:
        total_gap += gap
        print(f"{(busy_end - t0) / 1e3:10.1f} us: nothing runs for {gap:8.1f} us between {short(prev)} and {short(name)}")
    if e > busy_end: busy_end = e; prev = name
print(f"span {(busy_end - t0) / 1e3:.1f} us, idle in gaps over {min_us:g} us: {total_gap:.1f} us")
