#!/bin/bash
# round 4: (1) which fuzz case aborts, (2) sg_sort phase cycles, (3) T / THREADS variants of the segmented sort
mkdir -p gpurun_out; export TMPDIR=/tmp
cd tests
timeout 600 python - > ../gpurun_out/b_fuzz.log 2>&1 <<'PY'
import sys, numpy as np
import parity_cases as P
be = P.GpuBackend()
orig = P.K.Codec.dev_compress
def traced(self, *a, **k):
    print("dev_compress", self.transform if hasattr(self, "transform") else "", a[1], flush=True)
    return orig(self, *a, **k)
P.K.Codec.dev_compress = traced
origc = P.K.Codec.__init__
def tinit(self, *a, **k):
    print("codec", a, {x: k[x] for x in k if x != "lib"}, flush=True)
    return origc(self, *a, **k)
P.K.Codec.__init__ = tinit
P.check_fuzz(be, cases=300, seed=20260924, max_n=600000)
print("fuzz ok")
PY
echo "fuzz rc=$?"; tail -6 ../gpurun_out/b_fuzz.log
cd ..
for v in prof prof512; do
KNZ_GPU_LIB=$PWD/kanzi-go_amd/variants/libknz_$v.so KNZ_BWT_PROF=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-host-hook --no-verify --steps 1 --warmup 1 > gpurun_out/b_bench_$v.json 2> gpurun_out/b_bench_$v.err; echo "$v rc=$?"
grep -A1 "suffix sort\|sg_sort" gpurun_out/b_bench_$v.err | grep "sg_sort" | tail -8
done
for v in t2048_512 t1024_256 t1024_512; do
KNZ_GPU_LIB=$PWD/kanzi-go_amd/variants/libknz_$v.so timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-host-hook --steps 4 --warmup 1 > gpurun_out/b_bench_$v.json 2> gpurun_out/b_bench_$v.err; echo "$v rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/b_bench_$v.json').read().strip().splitlines()[-1])
print('$v', d['encode_MBps'], d['bit_exact_vs_oracle'], d['roofline'].get('phase_ms_per_step'), d['roofline']['all_stage_ms']['enc_transform'])
PY
done
