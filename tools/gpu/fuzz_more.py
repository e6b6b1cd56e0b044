import sys
sys.path.insert(0,'tests')
import parity_cases as P
be=P.GpuBackend()
for seed in (1,2,3):
    P.check_fuzz(be, cases=1000, seed=seed, max_n=2500000)
    print('seed',seed,'ok',flush=True)
