"""Per-round timeline of the suffix sort from a rocprofv3 kernel trace (rocpd .db): span, busy time and the kernels of every round of the last step."""
import collections, glob, re, sqlite3, sys
db = sys.argv[1] if len(sys.argv) > 1 else glob.glob('gpurun_out/prof_kt/*.db')[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if 'ss_hist' in r[0] and 'ILb1' in r[0]]
i0 = idx[-1]
rounds = []
for r in rows[i0:]:
    n = r[0]
    if 'knz_sbrt' in n:
        break
    if 'sg_keys' in n or 'ss_heads_kernel' in n or not rounds:
        rounds.append([r[1], r[1], collections.OrderedDict()])
    m = re.search(r'knz_\w+?_kernel', n)
    key = m.group(0) if m else n[-30:]
    rounds[-1][2][key] = rounds[-1][2].get(key, 0) + (r[2] - r[1]) / 1e3
    rounds[-1][1] = r[2]
tot = 0
for s, e, d in rounds:
    busy = sum(d.values()); tot += (e - s) / 1e3
    print(f"span {(e-s)/1e3:8.1f} us busy {busy:8.1f} :", ', '.join(f"{k[4:-7]} {v:.0f}" for k, v in d.items() if v > 15))
print("total span", round(tot, 1), "us; first to last", round((rounds[-1][1] - rounds[0][0]) / 1e3, 1))
