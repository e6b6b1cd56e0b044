#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (rocpd sqlite): one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE.
usage: pmc_traffic.py fetch.db write.db out.json
Per-launch averages (counter values summed over the XCDs of a dispatch, averaged over the dispatches of a kernel).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts half of the bytes of wide coalesced reads, so
fetch bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE (KB) is taken as is."""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, sum(counter_value), count(distinct dispatch_id) from pmc_events where counter_name=? group by name", (counter,))
    return {name.split("(")[0]: total / max(n, 1) for name, total, n in rows}


def per_kernel_time(path):
    """{kernel: (average launch duration in us, launches)} from the kernel trace of the same database (view top_kernels: durations in us)."""
    cur = sqlite3.connect(path).cursor()
    return {name.split("(")[0]: (avg, calls) for name, calls, avg in cur.execute("select name,total_calls,average from top_kernels")}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) -- python bench.py --steps 2 "
                   "--warmup 1 --no-cpu-baseline --no-verify ; per-launch averages summed over the XCDs (tools/pmc_traffic.py). gfx950 "
                   "correction per MI355X_MICROARCH.md (HBM section): fetch bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE taken as is.",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("knz_"):
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out["kernels"][k] = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1), "hbm_bytes_corrected": int(2 * f * 1024 + w * 1024)}
    ks = out["kernels"]
    enc = [k for k in ("knz_huf_hist_kernel", "knz_huf_lengths_kernel", "knz_huf_encode_kernel") if k in ks]
    if len(enc) == 3:
        ks["knz_huf_hist+lengths+encode_kernels"] = {"hbm_bytes_corrected": sum(ks[k]["hbm_bytes_corrected"] for k in enc)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
