#!/usr/bin/env python3
"""Timeline of the LAST decode of a rocprofv3 kernel trace (rocpd .db) of `bench.py`: every kernel from the decode's block walk on, with its start and end
relative to the walk's start, and the gaps in which no kernel ran. usage: decode_timeline.py results.db [min_us] [first kernel of a decode, default knz_dec_walk_blocks_kernel]"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
qcol = "d.queue_id" if "queue_id" in cols else ("d.stream_id" if "stream_id" in cols else "0")
rows = c.execute(f"select s.kernel_name, d.start, d.end, {qcol} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
first = sys.argv[3] if len(sys.argv) > 3 else 'knz_dec_walk_blocks_kernel'
i0 = [i for i, r in enumerate(rows) if first in r[0]][-1]
t0 = rows[i0][1]
busy_end = t0
for name, s, e, q in rows[i0:]:
    m = re.search(r'knz_\w+', name)
    short = m.group(0) if m else name[:40]
    if 'at::native' in name: break
    gap = (s - busy_end) / 1e3
    if gap > 50: print(f"            -- nothing runs for {gap:9.1f} us --")
    if (e - s) / 1e3 >= min_us or gap > 50:
        print(f"{(s - t0) / 1e3:10.1f} .. {(e - t0) / 1e3:10.1f} us  ({(e - s) / 1e3:9.1f})  q{q}  {short}")
    busy_end = max(busy_end, e)
print(f"decode span {(busy_end - t0) / 1e3:.1f} us")
