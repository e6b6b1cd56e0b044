"""Exploratory differential fuzz on the MI355X beyond the suite's fixed seed (tests/parity_cases.py check_fuzz)."""
import sys
sys.path.insert(0, "tests")
import parity_cases as P

be = P.GpuBackend()
for seed in (int(x) for x in (sys.argv[1:] or ["1", "2", "3"])):
    P.check_fuzz(be, cases=1000, seed=seed, max_n=1000000)
    print("seed", seed, "ok", flush=True)
