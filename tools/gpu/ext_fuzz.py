#!/usr/bin/env python3
"""Differential fuzz on the device with seeds other than the suite's (tests/parity_cases.py: check_fuzz, check_bwt_sort_fuzz):
random transform sequences x entropy codecs x block sizes x checksums x data shapes, device stream == oracle stream in both directions.
Arguments: seconds to spend (default 90), first seed (default 1000)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as P


class Env:                                                               # (what the helpers need of pytest's monkeypatch)
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k, raising=False): os.environ.pop(k, None)


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
P.K.build_library()
be = P.GpuBackend()
t0 = time.time(); n = 0
while time.time() - t0 < budget:
    P.check_fuzz(be, cases=100, seed=seed, max_n=600000); n += 100
    P.check_bwt_sort_fuzz(be, Env(), cases=30, seed=seed + 1, max_n=3000000, segs=("",)); n += 30
    print(f"seeds {seed}, {seed + 1}: ok ({n} cases, {time.time() - t0:.0f} s)", flush=True)
    seed += 2
